/*
 * or_api.c -- ORACLE (test infrastructure): flat entry points for the ctypes loader
 * oracle/oracle.py (scalars and plain pointers only).
 */
#include "or_internal.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* The program printed in the reference's trees/README.md:24-43, parameterised */
char *or_api_readme_tree(const char *tmpl, double m1, double m2, int Nq_in, int Nq_out, int Nq_cha)
{
    or_tree *t = or_tree_parse(tmpl, OR_VARTREE);
    if (!t) return NULL;
    or_dvec p_msg = or_gaussian_pmf(m1, sqrt(2 * m1), Nq_in, sqrt(2 * m1) / 20);
    or_dvec p_cha = or_gaussian_pmf(m2, sqrt(2 * m2), Nq_in, sqrt(2 * m2) / 20);
    or_tree_set_resolution(t, Nq_in, Nq_out, Nq_cha);
    or_tree_set_leaves(t, p_msg, p_cha);
    or_dvec p = or_tree_update(t, 0);
    char *s = or_tree_serialize(t);
    or_dvec_free(&p); or_dvec_free(&p_msg); or_dvec_free(&p_cha); or_tree_free(t);
    return s;
}

void or_api_code_dims(const or_code *c, int *nvar, int *nchk, int *nedges) { *nvar = c->nvar; *nchk = c->nchk; *nedges = c->nedges; }
void or_api_code_graph(const or_code *c, int *dv, int *dc, int *cn_msg_idx)
{
    memcpy(dv, c->dv, sizeof(int) * (size_t)c->nvar);
    memcpy(dc, c->dc, sizeof(int) * (size_t)c->nchk);
    or_code_cn_msg_idx(c, cn_msg_idx);
}
/* chk_equ_idx flattened row by row (LDPC_Code_LUT.cpp:531-535) */
void or_api_code_rows(const or_code *c, int *row_idx) { memcpy(row_idx, c->row_idx, sizeof(int) * (size_t)c->nedges); }

int or_api_codec_ninfo(const or_codec *c) { return c->nvar - c->nchk_lin_indep; }
int or_api_codec_rank(const or_codec *c) { return c->nchk_lin_indep; }
void or_api_codec_set_rank(or_codec *c, int r) { c->nchk_lin_indep = r; }
void or_api_codec_set_initial_message_mode(or_codec *c, int m) { c->initial_message_mode = m; }
char *or_api_codec_var_tree_txt(const or_codec *c) { return or_tree_array_serialize(c->var_trees); }
char *or_api_codec_chk_tree_txt(const or_codec *c) { return or_tree_array_serialize(c->chk_trees); }
int or_api_codec_qb(const or_codec *c, int which, double *out)
{
    const or_dvec *q = which == 0 ? &c->qb_Cha : &c->qb_Msg;
    if (out) memcpy(out, q->v, sizeof(double) * (size_t)q->n);
    return q->n;
}
int or_api_codec_cha2msg_map(const or_codec *c, int *out)
{
    if (out) memcpy(out, c->Nq_Cha_2_Nq_Msg_map.v, sizeof(int) * (size_t)c->Nq_Cha_2_Nq_Msg_map.n);
    return c->Nq_Cha_2_Nq_Msg_map.n;
}
void or_api_quant_nonlin_vec(const double *x, int n, const double *bounds, int nb, uint8_t *out)
{
    for (int i = 0; i < n; i++) out[i] = (uint8_t)or_quant_nonlin(x[i], bounds, nb);
}

/* per-iteration message dump for debugging kernels: runs lut_decode on one frame and copies
 * the message buffer after `stop_after_half_iters` half-iterations (CN pass = 1, VN pass = 2,
 * ...); implemented by clamping max_iters, so only meaningful with psc=pisc=0. */
void or_api_codec_msgs(const or_codec *c, int *out) { memcpy(out, c->msgs, sizeof(int) * (size_t)c->nedges); }

/* DE threshold of an ensemble given by explicit edge distributions (README.md:138-178 set-up,
 * prog/de_sim.cpp:137-260): auto trees, no reuse, uniform resolution. */
int or_api_de_threshold(const int *dl, const double *l, int nl, const int *dr, const double *r, int nr,
                        int qbits_cha, int qbits_msg, int maxiter_de, int min_lut, const char *tree_mode,
                        const char *strategy, double thr_min, double thr_prec, double Pe_max,
                        int maxiter_bisec, int max_ni_de_iters, double LLR_max, int Nq_fine,
                        double *thr_out)
{
    or_ensemble *ens = or_ensemble_from_edge_dist(dl, l, nl, dr, r, nr);
    int *Nq = (int *)malloc(sizeof(int) * (size_t)maxiter_de);
    for (int i = 0; i < maxiter_de; i++) Nq[i] = 1 << qbits_msg;
    or_tree_array *var_t = NULL, *chk_t = NULL;
    if (or_get_lut_tree_templates(tree_mode, ens, Nq, maxiter_de, 1 << qbits_cha, min_lut, 0, &var_t, &chk_t) != 0) {
        free(Nq); or_ensemble_free(ens); return -2;
    }
    or_de_lut *de = or_de_lut_new(ens, 1 << qbits_cha, Nq, maxiter_de, var_t, chk_t, NULL, strategy);
    de->thr_prec = thr_prec; de->Pe_max = Pe_max; de->maxiter_bisec = maxiter_bisec;
    de->LLR_max = LLR_max; de->Nq_fine = Nq_fine; de->max_ni_de_iters = max_ni_de_iters;
    de->thr_min = thr_min;   /* set_bisec_window(thr_min, shannon threshold) */
    int it = or_de_lut_bisec_search(de, thr_out);
    or_de_lut_free(de); or_tree_array_free(var_t); or_tree_array_free(chk_t); or_ensemble_free(ens); free(Nq);
    return it;
}

/* evaluate tree [set][cls] of the codec for one node (tests of the product's tree compiler) */
int or_api_codec_tree_eval(const or_codec *c, int kind, int set, int cls, const int *in, int n_in, int *out)
{
    const or_tree_array *a = (kind == OR_CHKTREE) ? c->chk_trees : c->var_trees;
    if (!a || set < 0 || set >= a->n_sets || cls < 0 || cls >= a->n_classes[set]) return -1;
    const or_tree *t = a->t[set][cls];
    if (kind == OR_VARTREE) or_tree_var_msg_update(t, in, n_in - 1, in[n_in - 1], out);
    else if (kind == OR_CHKTREE) or_tree_chk_msg_update(t, in, n_in, out);
    else out[0] = or_tree_dec_update(t, in, n_in - 1, in[n_in - 1]);
    return 0;
}
int or_api_codec_tree_info(const or_codec *c, int kind, int set, int cls, int *type, int *num_leaves)
{
    const or_tree_array *a = (kind == OR_CHKTREE) ? c->chk_trees : c->var_trees;
    if (!a || set < 0 || set >= a->n_sets || cls < 0 || cls >= a->n_classes[set]) return -1;
    *type = a->t[set][cls]->type; *num_leaves = a->t[set][cls]->num_leaves;
    return 0;
}
int or_api_codec_n_sets(const or_codec *c, int chk) { const or_tree_array *a = chk ? c->chk_trees : c->var_trees; return a ? a->n_sets : 0; }
