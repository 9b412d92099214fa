/*
 * or_flat.c -- ORACLE (test infrastructure), flat-table mode: the CPU side-by-side leg (ii) of SURVEY.md 8(d).
 *
 * The same decode as or_codec_lut_decode (src/LDPC_Code_LUT.cpp:259-353) and the same node semantics
 * (src/LUT_Tree.cpp:402-445,774-820), but the way a CPU implementation that cares about speed would run them:
 *   - every tree is flattened ONCE into arrays (nodes in post-order, children as node / queue-position references,
 *     the half tables of the reference expanded to full tables), no recursion and no per-output queue copies
 *     beyond a small stack array;
 *   - frames are independent (one reference process decodes one frame at a time, src/LDPC_BER_Sim.cpp:260-291):
 *     one frame per thread, all cores, each thread with its own message buffer.
 * It is checked bit for bit against the faithful mode (tests/test_oracle_flat.py); bench.py times both.
 */
#include "or_internal.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n_nodes;          /* LUT nodes, post-order: the root is the last one */
    int *nchild, *first;  /* children of node j: refs[first[j] .. first[j] + nchild[j]) */
    int *ref;             /* >= 0: value of node ref;  < 0: queue element -(ref + 1) (leaves pop the queue in DFS order) */
    int *childK;          /* alphabet of that child */
    int *K;               /* alphabet of the node */
    int *tab, *tab_off, *tab_half;   /* full table of node j at tab + tab_off[j]; tab_half[j] = |Q| of the reference */
} flat_tree;

struct or_flat {
    const or_codec *c;
    flat_tree **var, **chk;          /* [set][class] in the order of the codec's tree arrays */
};

/* ---- flattening ---------------------------------------------------------------------------- */
static int is_leaf(const or_node *n) { return n->type == OR_MSG || n->type == OR_CHA; }
static void count_rec(const or_node *n, int *nodes, int *refs, int *tab, int chk)
{
    if (is_leaf(n)) return;
    long total = 1;
    for (int i = 0; i < n->nchild; i++) { count_rec(n->child[i], nodes, refs, tab, chk); total *= chk ? n->child[i]->K / 2 : n->child[i]->K; }
    (*nodes)++; *refs += n->nchild; *tab += (int)total;
}
/* returns the reference of the value of n: node index or -(queue position + 1) */
static int flat_rec(const or_node *n, flat_tree *f, int *nodes, int *refs, int *tab, int *leaf_pos, int chk)
{
    if (is_leaf(n)) return -((*leaf_pos)++ + 1);
    int r[64];
    for (int i = 0; i < n->nchild; i++) r[i] = flat_rec(n->child[i], f, nodes, refs, tab, leaf_pos, chk);
    const int j = (*nodes)++;
    f->nchild[j] = n->nchild; f->first[j] = *refs; f->K[j] = n->K; f->tab_off[j] = *tab; f->tab_half[j] = n->Q.n;
    long total = 1;
    for (int i = 0; i < n->nchild; i++) {
        f->ref[*refs + i] = r[i]; f->childK[*refs + i] = n->child[i]->K;
        total *= chk ? n->child[i]->K / 2 : n->child[i]->K;
    }
    *refs += n->nchild;
    for (long l = 0; l < total; l++) {
        int v;
        if (chk) v = n->Q.v[l];                                               /* sign handled at evaluation, :420-445 */
        else v = l < n->Q.n ? n->Q.v[l] : n->K - 1 - n->Q.v[2 * n->Q.n - 1 - l];   /* :402-418 */
        f->tab[*tab + l] = v;
    }
    *tab += (int)total;
    return j;
}
static flat_tree *flatten(const or_tree *t)
{
    const int chk = t->type == OR_CHKTREE;
    int nodes = 0, refs = 0, tab = 0;
    count_rec(t->root, &nodes, &refs, &tab, chk);
    flat_tree *f = (flat_tree *)calloc(1, sizeof(flat_tree));
    f->n_nodes = nodes;
    f->nchild = (int *)malloc(sizeof(int) * (size_t)(nodes + 1)); f->first = (int *)malloc(sizeof(int) * (size_t)(nodes + 1));
    f->K = (int *)malloc(sizeof(int) * (size_t)(nodes + 1)); f->tab_off = (int *)malloc(sizeof(int) * (size_t)(nodes + 1));
    f->tab_half = (int *)malloc(sizeof(int) * (size_t)(nodes + 1));
    f->ref = (int *)malloc(sizeof(int) * (size_t)(refs + 1)); f->childK = (int *)malloc(sizeof(int) * (size_t)(refs + 1));
    f->tab = (int *)malloc(sizeof(int) * (size_t)(tab + 1));
    int n2 = 0, r2 = 0, t2 = 0, lp = 0;
    flat_rec(t->root, f, &n2, &r2, &t2, &lp, chk);
    return f;
}
static void flat_tree_free(flat_tree *f)
{
    if (!f) return;
    free(f->nchild); free(f->first); free(f->K); free(f->tab_off); free(f->tab_half); free(f->ref); free(f->childK); free(f->tab); free(f);
}

or_flat *or_flat_new(const or_codec *c)
{
    or_flat *F = (or_flat *)calloc(1, sizeof(or_flat));
    F->c = c;
    F->var = (flat_tree **)calloc((size_t)c->var_trees->n_sets * 64, sizeof(flat_tree *));
    for (int s = 0; s < c->var_trees->n_sets; s++)
        for (int k = 0; k < c->var_trees->n_classes[s] && k < 64; k++) F->var[s * 64 + k] = flatten(c->var_trees->t[s][k]);
    if (c->chk_trees) {
        F->chk = (flat_tree **)calloc((size_t)c->chk_trees->n_sets * 64, sizeof(flat_tree *));
        for (int s = 0; s < c->chk_trees->n_sets; s++)
            for (int k = 0; k < c->chk_trees->n_classes[s] && k < 64; k++) F->chk[s * 64 + k] = flatten(c->chk_trees->t[s][k]);
    }
    return F;
}
void or_flat_free(or_flat *F)
{
    if (!F) return;
    for (int s = 0; s < F->c->var_trees->n_sets * 64; s++) flat_tree_free(F->var[s]);
    free(F->var);
    if (F->chk) { for (int s = 0; s < F->c->chk_trees->n_sets * 64; s++) flat_tree_free(F->chk[s]); free(F->chk); }
    free(F);
}

/* ---- evaluation ------------------------------------------------------------------------------ */
/* variable / decision tree on the queue q (src/LUT_Tree.cpp:402-418) */
static inline int eval_var(const flat_tree *f, const int *q, int *val)
{
    if (f->n_nodes == 0) return q[0];
    for (int j = 0; j < f->n_nodes; j++) {
        int label = 0, base = 1;
        const int *r = f->ref + f->first[j], *ck = f->childK + f->first[j];
        for (int i = 0; i < f->nchild[j]; i++) { label += base * (r[i] >= 0 ? val[r[i]] : q[-r[i] - 1]); base *= ck[i]; }
        val[j] = f->tab[f->tab_off[j] + label];
    }
    return val[f->n_nodes - 1];
}
/* check tree (src/LUT_Tree.cpp:420-445) */
static inline int eval_chk(const flat_tree *f, const int *q, int *val)
{
    if (f->n_nodes == 0) return q[0];
    for (int j = 0; j < f->n_nodes; j++) {
        int label = 0, base = 1, parity = 0;
        const int *r = f->ref + f->first[j], *ck = f->childK + f->first[j];
        for (int i = 0; i < f->nchild[j]; i++) {
            const int s = r[i] >= 0 ? val[r[i]] : q[-r[i] - 1], h = ck[i] / 2;
            if (s < h) { parity ^= 1; label += base * (h - 1 - s); } else label += base * (s - h);
            base *= h;
        }
        const int v = f->tab[f->tab_off[j] + label];
        val[j] = parity == 1 ? v : f->K[j] - 1 - v;
    }
    return val[f->n_nodes - 1];
}

static int syndrome_ok(const or_codec *c, const unsigned char *b)       /* LDPC_Code_LUT.cpp:455-469 */
{
    const or_code *H = c->code;
    for (int cc = 0; cc < c->nchk; cc++) {
        int synd = 0;
        for (int k = H->row_ptr[cc]; k < H->row_ptr[cc + 1]; k++) synd ^= b[H->row_idx[k]];
        if (synd & 1) return 0;
    }
    return 1;
}

/* one frame; m: nedges ints of scratch (src/LDPC_Code_LUT.cpp:259-353) */
static int decode_one(const or_flat *F, const uint8_t *cha, const uint8_t *msg0, unsigned char *out, int *m)
{
    const or_codec *c = F->c;
    const or_code *H = c->code;
    int one[512], res[512], val[1024];
    for (int v = 0; v < c->nvar; v++) out[v] = cha[v] < c->Nq_Cha / 2;
    if (c->pisc && syndrome_ok(c, out)) return 0;
    int e = 0;
    for (int v = 0; v < c->nvar; v++) for (int k = 0; k < H->dv[v]; k++) m[e++] = msg0[v];
    for (int ii = 0; ii < c->max_iters; ii++) {
        e = 0;
        const int nz = c->Nq_Msg.v[ii] / 2;
        for (int cc = 0; cc < c->nchk; cc++) {
            const int dc = H->dc[cc];
            const int *ix = c->cn_msg_idx + e;
            if (c->minLUT) {                                               /* chk_update_minsum, :355-402 */
                int min1 = nz, min2 = nz, min_idx = 0, sp = 0;
                for (int k = 0; k < dc; k++) {
                    const int x = m[ix[k]];
                    int t;
                    if (x < nz) { sp ^= 1; t = nz - 1 - x; } else t = x - nz;
                    if (t < min1) { min2 = min1; min1 = t; min_idx = k; } else if (t < min2) min2 = t;
                }
                for (int k = 0; k < dc; k++) {
                    const int t = (k == min_idx) ? min2 : min1;
                    const int sg = (m[ix[k]] < nz) ? (sp ^ 1) : sp;
                    m[ix[k]] = sg ? nz - 1 - t : nz + t;
                }
            } else {                                                       /* chk_update_lut, :416-426 */
                const flat_tree *f = F->chk[c->chk_tree_idx_iter[ii] * 64 + c->chk_tree_idx_degree[cc]];
                for (int i = 0; i < dc; i++) {
                    int k = 0;
                    for (int j = 0; j < dc; j++) if (j != i) one[k++] = m[ix[j]];
                    res[i] = eval_chk(f, one, val);
                }
                for (int k = 0; k < dc; k++) m[ix[k]] = res[k];
            }
            e += dc;
        }
        if (ii != c->max_iters - 1) {
            e = 0;
            for (int v = 0; v < c->nvar; v++) {                            /* var_update_lut, :404-414 */
                const int dv = H->dv[v];
                const flat_tree *f = F->var[c->var_tree_idx_iter[ii] * 64 + c->var_tree_idx_degree[v]];
                for (int i = 0; i < dv; i++) {
                    int k = 0;
                    for (int j = 0; j < dv; j++) if (j != i) one[k++] = m[e + j];
                    one[k] = cha[v];
                    res[i] = eval_var(f, one, val);
                }
                for (int k = 0; k < dv; k++) m[e + k] = res[k];
                e += dv;
            }
            if (c->psc) {                                                  /* syndrome_check(Nq, b), :437-452 */
                const int nz2 = c->Nq_Msg.v[ii + 1] / 2;
                int ok = 1;
                e = 0;
                for (int v = 0; v < c->nvar && ok; v++) {
                    const int bit = m[e] < nz2;
                    for (int k = 1; k < H->dv[v]; k++) if (bit != (m[e + k] < nz2)) { ok = 0; break; }
                    e += H->dv[v];
                    out[v] = (unsigned char)bit;
                }
                if (ok && syndrome_ok(c, out)) return ii + 1;
            }
        }
    }
    e = 0;
    for (int v = 0; v < c->nvar; v++) {                                    /* dec_update_lut, :428-434 */
        const int dv = H->dv[v];
        const flat_tree *f = F->var[c->var_tree_idx_iter[c->max_iters - 1] * 64 + c->var_tree_idx_degree[v]];
        for (int j = 0; j < dv; j++) one[j] = m[e + j];
        one[dv] = cha[v];
        out[v] = eval_var(f, one, val) < 1;
        e += dv;
    }
    return syndrome_ok(c, out) ? c->max_iters : -c->max_iters;
}

/* ---- batch over threads ------------------------------------------------------------------------ */
typedef struct { const or_flat *F; const uint8_t *cha, *msg0; uint8_t *out; int32_t *iters; int B, next; pthread_mutex_t mu; } job_t;
static void *worker(void *arg)
{
    job_t *J = (job_t *)arg;
    const int N = J->F->c->nvar;
    int *m = (int *)malloc(sizeof(int) * (size_t)J->F->c->nedges);
    for (;;) {
        pthread_mutex_lock(&J->mu);
        const int f = J->next++;
        pthread_mutex_unlock(&J->mu);
        if (f >= J->B) break;
        J->iters[f] = decode_one(J->F, J->cha + (size_t)f * N, J->msg0 + (size_t)f * N, J->out + (size_t)f * N, m);
    }
    free(m);
    return NULL;
}
void or_flat_decode_batch_u8(const or_flat *F, const uint8_t *cha, const uint8_t *msg0, int B, uint8_t *out_bits, int32_t *out_iters, int n_threads)
{
    job_t J = { F, cha, msg0, out_bits, out_iters, B, 0, PTHREAD_MUTEX_INITIALIZER };
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    for (int t = 1; t < n_threads; t++) pthread_create(&th[t], NULL, worker, &J);
    worker(&J);
    for (int t = 1; t < n_threads; t++) pthread_join(th[t], NULL);
}
