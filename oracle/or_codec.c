/*
 * or_codec.c -- ORACLE (test infrastructure): restatement of src/LDPC_Code_LUT.cpp -- codec
 * set-up, LUT design entry point and the quantised message-passing decoder lut_decode().
 * The evaluation keeps the reference's structure (per node: gather the messages, evaluate
 * the tree once per output on a queue with that element removed, write back in place), so
 * that timing it stands in for the single-threaded reference; only the std::deque heap
 * traffic is replaced by stack arrays (which makes this baseline faster, not slower).
 */
#include "or_internal.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* set_code -> decoder_parameterization, LDPC_Code_LUT.cpp:471-541 */
or_codec *or_codec_new(or_code *code, int skip_rank)
{
    or_codec *c = (or_codec *)calloc(1, sizeof(or_codec));
    c->code = code;
    c->nvar = code->nvar; c->nchk = code->nchk; c->nedges = code->nedges;
    /* :493-499 -- the reference also skips the rank for nvar >= 1e5 */
    if (skip_rank || code->nvar >= 1e5) c->nchk_lin_indep = code->nchk;
    else c->nchk_lin_indep = or_code_gf2_rank(code);
    c->cn_msg_idx = (int *)malloc(sizeof(int) * (size_t)c->nedges);
    or_code_cn_msg_idx(code, c->cn_msg_idx);
    c->msgs = (int *)malloc(sizeof(int) * (size_t)c->nedges);
    c->max_iters = 50; c->psc = 1; c->pisc = 0;       /* constructor defaults, LDPC_Code_LUT.cpp:42-60 */
    c->faithful = 1;
    return c;
}

static void free_trees(or_codec *c)
{
    or_tree_array_free(c->var_trees); or_tree_array_free(c->chk_trees);
    c->var_trees = c->chk_trees = NULL;
    free(c->var_tree_idx_iter); free(c->chk_tree_idx_iter); free(c->var_tree_idx_degree); free(c->chk_tree_idx_degree);
    c->var_tree_idx_iter = c->chk_tree_idx_iter = c->var_tree_idx_degree = c->chk_tree_idx_degree = NULL;
}

void or_codec_free(or_codec *c)
{
    if (!c) return;
    free_trees(c);
    free(c->cn_msg_idx); free(c->msgs); free(c->reuse_vec);
    or_ivec_free(&c->Nq_Msg); or_dvec_free(&c->qb_Cha); or_dvec_free(&c->qb_Msg); or_ivec_free(&c->Nq_Cha_2_Nq_Msg_map);
    free(c);
}

/* set_trees, LDPC_Code_LUT.cpp:120-169 (takes ownership of the arrays) */
static int set_trees(or_codec *c, or_tree_array *var, or_tree_array *chk)
{
    int I = c->max_iters;
    if (c->reuse_vec[0] || c->reuse_vec[I - 1]) return -1;
    free_trees(c);
    c->var_tree_idx_iter = (int *)malloc(sizeof(int) * (size_t)I);
    int cum = 0;
    for (int i = 0; i < I; i++) { cum += !c->reuse_vec[i]; c->var_tree_idx_iter[i] = cum - 1; }
    c->var_tree_idx_degree = (int *)calloc((size_t)c->nvar, sizeof(int));
    for (int v = 0; v < c->nvar; v++) {
        int idx = -1;
        for (int d = 0; d < var->n_classes[0]; d++) if (var->t[0][d]->num_leaves == c->code->dv[v]) { idx = d; break; }
        if (idx < 0) return -1;
        c->var_tree_idx_degree[v] = idx;
    }
    c->var_trees = var;
    if (chk && chk->n_sets > 0) {
        c->chk_tree_idx_iter = (int *)malloc(sizeof(int) * (size_t)I);
        memcpy(c->chk_tree_idx_iter, c->var_tree_idx_iter, sizeof(int) * (size_t)I);
        c->chk_tree_idx_degree = (int *)calloc((size_t)c->nchk, sizeof(int));
        for (int r = 0; r < c->nchk; r++) {
            int idx = -1;
            for (int d = 0; d < chk->n_classes[0]; d++) if (chk->t[0][d]->num_leaves + 1 == c->code->dc[r]) { idx = d; break; }
            if (idx < 0) return -1;
            c->chk_tree_idx_degree[r] = idx;
        }
        c->chk_trees = chk;
    } else {
        or_tree_array_free(chk);
        c->chk_trees = NULL;
    }
    return 0;
}

static void set_params(or_codec *c, int min_lut, int max_iters, const unsigned char *reuse_vec, int Nq_Cha, const int *Nq_Msg)
{
    c->minLUT = min_lut; c->max_iters = max_iters; c->Nq_Cha = Nq_Cha;
    free(c->reuse_vec);
    c->reuse_vec = (unsigned char *)calloc((size_t)max_iters, 1);
    if (reuse_vec) memcpy(c->reuse_vec, reuse_vec, (size_t)max_iters);
    or_ivec_free(&c->Nq_Msg);
    c->Nq_Msg = or_ivec_new(max_iters);
    memcpy(c->Nq_Msg.v, Nq_Msg, sizeof(int) * (size_t)max_iters);
}

/* design_luts, LDPC_Code_LUT.cpp:699-746 */
double or_codec_design_luts(or_codec *c, const char *tree_method, int min_lut, double sigma2, int max_iters,
                            const unsigned char *reuse_vec, int Nq_Cha, const int *Nq_Msg, int allow_deg1)
{
    /* the reference stops here or shortly after: LDPC_DE.cpp:199 (no reuse in the first iteration), LDPC_Code_LUT.cpp:122
     * (first and last iteration are exempt from tree reuse); a reused stage in the last iteration would also meet the
     * two-label decision stage with the wrong alphabet */
    if (max_iters < 1 || (reuse_vec && (reuse_vec[0] || reuse_vec[max_iters - 1]))) return -1;
    /* a reused stage reads and writes the alphabets of the stage it repeats (LDPC_DE.cpp:434-487,505-557 add probability
     * vectors of different lengths otherwise) */
    for (int i = 1; reuse_vec && i < max_iters; i++)
        if (reuse_vec[i] && (Nq_Msg[i] != Nq_Msg[i - 1] || (i + 1 < max_iters && Nq_Msg[i + 1] != Nq_Msg[i]))) return -1;
    set_params(c, min_lut, max_iters, reuse_vec, Nq_Cha, Nq_Msg);
    or_ensemble *ens = or_empirical_ensemble(c->code);
    or_tree_array *var_t = NULL, *chk_t = NULL;
    if (or_get_lut_tree_templates(tree_method, ens, Nq_Msg, max_iters, Nq_Cha, min_lut, allow_deg1, &var_t, &chk_t) != 0) {
        or_ensemble_free(ens);
        return -1;
    }
    or_de_lut *de = or_de_lut_new(ens, Nq_Cha, Nq_Msg, max_iters, var_t, chk_t, c->reuse_vec, "joint_root");
    double sig = sqrt(sigma2);
    or_dvec_free(&c->qb_Cha); or_dvec_free(&c->qb_Msg);
    or_de_lut_get_quant_bound(de, sig, &c->qb_Cha, &c->qb_Msg);
    or_tree_array *var_l = NULL, *chk_l = NULL;
    or_de_lut_evolve(de, sig, 1, &var_l, &chk_l);
    int rc = set_trees(c, var_l, chk_l);
    /* Nq_Cha_2_Nq_Msg_map, :735-741 */
    const double LLR_max_mag = 25.0;
    double delta = 2 * LLR_max_mag / Nq_Cha;
    or_dvec pmf_channel = or_gaussian_pmf(2 / (sig * sig), 2 / sig, Nq_Cha, delta);
    or_dvec p_msg;
    or_ivec_free(&c->Nq_Cha_2_Nq_Msg_map);
    (void)or_quant_mi_sym(&p_msg, &c->Nq_Cha_2_Nq_Msg_map, pmf_channel, Nq_Msg[0], 1);
    or_dvec_free(&p_msg); or_dvec_free(&pmf_channel);
    or_de_lut_free(de); or_tree_array_free(var_t); or_tree_array_free(chk_t); or_ensemble_free(ens);
    return rc == 0 ? sig : -1;
}

int or_codec_set_trees_txt(or_codec *c, const char *var_txt, const char *chk_txt, int max_iters,
                           const unsigned char *reuse_vec, int Nq_Cha, const int *Nq_Msg, int minLUT)
{
    set_params(c, minLUT, max_iters, reuse_vec, Nq_Cha, Nq_Msg);
    or_tree_array *var = or_tree_array_deserialize(var_txt);
    if (!var) return -1;
    or_tree_array *chk = (chk_txt && chk_txt[0]) ? or_tree_array_deserialize(chk_txt) : NULL;
    return set_trees(c, var, chk);
}

void or_codec_set_exit_conditions(or_codec *c, int max_iters, int psc, int pisc)
{
    c->max_iters = max_iters; c->psc = psc; c->pisc = pisc;
}

/* syndrome_check(const bvec&), LDPC_Code_LUT.cpp:455-469 */
int or_codec_syndrome_ok(const or_codec *c, const unsigned char *b)
{
    const or_code *H = c->code;
    for (int cc = 0; cc < c->nchk; cc++) {
        int synd = 0;
        for (int k = H->row_ptr[cc]; k < H->row_ptr[cc + 1]; k++) if (b[H->row_idx[k]]) synd++;
        if (synd & 1) return 0;
    }
    return 1;
}

/* syndrome_check(int Nq_Msg, bvec&), LDPC_Code_LUT.cpp:437-452 */
static int syndrome_msgs(const or_codec *c, int Nq, unsigned char *b)
{
    int e = 0, nz = Nq / 2;
    for (int v = 0; v < c->nvar; v++) {
        int bit = c->msgs[e] < nz;
        for (int k = 1; k < c->code->dv[v]; k++) if (bit != (c->msgs[e + k] < nz)) return 0;
        e += c->code->dv[v];
        b[v] = (unsigned char)bit;
    }
    return or_codec_syndrome_ok(c, b);
}

/* chk_update_minsum, LDPC_Code_LUT.cpp:355-402 */
static void chk_update_minsum(or_codec *c, int node, int e0, int iter)
{
    int *m = c->msgs; const int *ix = c->cn_msg_idx + e0;
    int dc = c->code->dc[node], nz = c->Nq_Msg.v[iter] / 2;
    int min1 = nz, min2 = nz, min_idx = 0, sign_prod = 0, tmp;
    for (int k = 0; k < dc; k++) {
        if (m[ix[k]] < nz) { sign_prod ^= 1; tmp = nz - 1 - m[ix[k]]; }
        else tmp = m[ix[k]] - nz;
        if (tmp < min1) { min2 = min1; min1 = tmp; min_idx = k; }
        else if (tmp < min2) min2 = tmp;
    }
    for (int k = 0; k < dc; k++) {
        tmp = (k == min_idx) ? min2 : min1;
        int sign_msg = (m[ix[k]] < nz) ? (sign_prod ^ 1) : sign_prod;
        m[ix[k]] = sign_msg ? nz - 1 - tmp : nz + tmp;
    }
}

/* lut_decode, LDPC_Code_LUT.cpp:259-353 */
int or_codec_lut_decode(or_codec *c, const int *cha, const int *msg0, unsigned char *out)
{
    const or_code *H = c->code;
    int in[512], res[512];
    for (int v = 0; v < c->nvar; v++) out[v] = cha[v] < c->Nq_Cha / 2;
    if (c->pisc && or_codec_syndrome_ok(c, out)) return 0;
    int e = 0;
    for (int v = 0; v < c->nvar; v++) for (int k = 0; k < H->dv[v]; k++) c->msgs[e++] = msg0[v];

    for (int ii = 0; ii < c->max_iters; ii++) {
        e = 0;
        for (int cc = 0; cc < c->nchk; cc++) {
            int dc = H->dc[cc];
            if (c->minLUT) chk_update_minsum(c, cc, e, ii);
            else {  /* chk_update_lut, :416-426 */
                for (int k = 0; k < dc; k++) in[k] = c->msgs[c->cn_msg_idx[e + k]];
                or_tree_chk_msg_update(c->chk_trees->t[c->chk_tree_idx_iter[ii]][c->chk_tree_idx_degree[cc]], in, dc, res);
                for (int k = 0; k < dc; k++) c->msgs[c->cn_msg_idx[e + k]] = res[k];
            }
            e += dc;
        }
        if (ii != c->max_iters - 1) {
            e = 0;
            for (int v = 0; v < c->nvar; v++) {  /* var_update_lut, :404-414 */
                int dv = H->dv[v];
                or_tree_var_msg_update(c->var_trees->t[c->var_tree_idx_iter[ii]][c->var_tree_idx_degree[v]], c->msgs + e, dv, cha[v], res);
                for (int k = 0; k < dv; k++) c->msgs[e + k] = res[k];
                e += dv;
            }
            if (c->psc && syndrome_msgs(c, c->Nq_Msg.v[ii + 1], out)) return ii + 1;
        }
    }
    e = 0;
    for (int v = 0; v < c->nvar; v++) {          /* dec_update_lut, :428-434 */
        int dv = H->dv[v];
        out[v] = or_tree_dec_update(c->var_trees->t[c->var_tree_idx_iter[c->max_iters - 1]][c->var_tree_idx_degree[v]], c->msgs + e, dv, cha[v]) < 1;
        e += dv;
    }
    return or_codec_syndrome_ok(c, out) ? c->max_iters : -c->max_iters;
}

/* lut_decode with output_verbosity = level > 1: the message dumps of LDPC_Code_LUT.cpp:292-298 (initial), :311-317 (after the
 * check update, level > 2) and :331-337 (after the variable update), appended to the text file `path` exactly as the reference
 * streams them to std::cout: every message as setfill('0') << setw(8) << uppercase << hex, two blanks after each; std::hex and
 * std::uppercase are sticky, so the iteration numbers of the later headlines come out in upper-case hex as well.  A frame that
 * leaves through the exit test returns BEFORE the dump of its last variable update (:327-329 precede :331). */
static void dump_msgs(FILE *f, const or_codec *c)
{
    for (int e = 0; e < c->code->nedges; e++) fprintf(f, "%08X  ", (unsigned)c->msgs[e]);
    fprintf(f, "\n");
}
int or_codec_lut_decode_dump(or_codec *c, const int *cha, const int *msg0, unsigned char *out, int level, const char *path)
{
    const or_code *H = c->code;
    int in[512], res[512];
    FILE *f = fopen(path, "a");
    if (!f) return -999999;
    for (int v = 0; v < c->nvar; v++) out[v] = cha[v] < c->Nq_Cha / 2;
    if (c->pisc && or_codec_syndrome_ok(c, out)) { fclose(f); return 0; }
    int e = 0;
    for (int v = 0; v < c->nvar; v++) for (int k = 0; k < H->dv[v]; k++) c->msgs[e++] = msg0[v];
    if (level > 1) { fprintf(f, "Initial VN-to-CN messages: \n"); dump_msgs(f, c); }
    for (int ii = 0; ii < c->max_iters; ii++) {
        e = 0;
        for (int cc = 0; cc < c->nchk; cc++) {
            int dc = H->dc[cc];
            if (c->minLUT) chk_update_minsum(c, cc, e, ii);
            else {
                for (int k = 0; k < dc; k++) in[k] = c->msgs[c->cn_msg_idx[e + k]];
                or_tree_chk_msg_update(c->chk_trees->t[c->chk_tree_idx_iter[ii]][c->chk_tree_idx_degree[cc]], in, dc, res);
                for (int k = 0; k < dc; k++) c->msgs[c->cn_msg_idx[e + k]] = res[k];
            }
            e += dc;
        }
        if (level > 2) { fprintf(f, "CN-to-VN messages after CN update at iteration %X:\n", (unsigned)ii); dump_msgs(f, c); }
        if (ii != c->max_iters - 1) {
            e = 0;
            for (int v = 0; v < c->nvar; v++) {
                int dv = H->dv[v];
                or_tree_var_msg_update(c->var_trees->t[c->var_tree_idx_iter[ii]][c->var_tree_idx_degree[v]], c->msgs + e, dv, cha[v], res);
                for (int k = 0; k < dv; k++) c->msgs[e + k] = res[k];
                e += dv;
            }
            if (c->psc && syndrome_msgs(c, c->Nq_Msg.v[ii + 1], out)) { fclose(f); return ii + 1; }
        }
        if (level > 1) { fprintf(f, "VN-to-CN messages after VN update at iteration %X:\n", (unsigned)ii); dump_msgs(f, c); }
    }
    e = 0;
    for (int v = 0; v < c->nvar; v++) {
        int dv = H->dv[v];
        out[v] = or_tree_dec_update(c->var_trees->t[c->var_tree_idx_iter[c->max_iters - 1]][c->var_tree_idx_degree[v]], c->msgs + e, dv, cha[v]) < 1;
        e += dv;
    }
    fclose(f);
    return or_codec_syndrome_ok(c, out) ? c->max_iters : -c->max_iters;
}

/* decode(const vec&, bvec&), LDPC_Code_LUT.cpp:204-226 (the caller keeps the first
 * nvar - nchk_lin_indep bits as systematic bits) */
int or_codec_decode_llr(or_codec *c, const double *llr, unsigned char *out, int *cha_labels, int *msg_labels)
{
    int *cha = cha_labels ? cha_labels : (int *)malloc(sizeof(int) * (size_t)c->nvar);
    int *msg = msg_labels ? msg_labels : (int *)malloc(sizeof(int) * (size_t)c->nvar);
    for (int v = 0; v < c->nvar; v++) cha[v] = or_quant_nonlin(llr[v], c->qb_Cha.v, c->qb_Cha.n);
    if (c->initial_message_mode == 0) for (int v = 0; v < c->nvar; v++) msg[v] = or_quant_nonlin(llr[v], c->qb_Msg.v, c->qb_Msg.n);
    else for (int v = 0; v < c->nvar; v++) msg[v] = c->Nq_Cha_2_Nq_Msg_map.v[cha[v]];
    int it = or_codec_lut_decode(c, cha, msg, out);
    if (!cha_labels) free(cha);
    if (!msg_labels) free(msg);
    return it;
}

void or_codec_lut_decode_batch_u8(or_codec *c, const uint8_t *cha, const uint8_t *msg0, int B, uint8_t *out_bits, int32_t *out_iters)
{
    int *a = (int *)malloc(sizeof(int) * (size_t)c->nvar), *b = (int *)malloc(sizeof(int) * (size_t)c->nvar);
    for (int f = 0; f < B; f++) {
        for (int v = 0; v < c->nvar; v++) { a[v] = cha[(size_t)f * c->nvar + v]; b[v] = msg0[(size_t)f * c->nvar + v]; }
        out_iters[f] = or_codec_lut_decode(c, a, b, out_bits + (size_t)f * c->nvar);
    }
    free(a); free(b);
}
