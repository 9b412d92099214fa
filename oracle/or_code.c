/*
 * or_code.c -- ORACLE (test infrastructure): alist loading, Tanner-graph indexing, GF(2)
 * rank and empirical degree distributions.  Replaces the IT++ LDPC_Parity / GF2mat calls the
 * reference makes at src/LDPC_BER_Sim.cpp:443, src/LDPC_Code_LUT.cpp:488-541 and
 * src/LDPC_Ensemble.cpp:391-423 (the IT++ fork itself is absent from the reference tree).
 */
#define _GNU_SOURCE
#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int cmp_int(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }

/* read one text line of integers; returns count, fills *out (realloc'd) */
static int read_int_line(FILE *f, int **out, int *cap)
{
    char *line = NULL; size_t lcap = 0;
    ssize_t len;
    int n = 0;
    /* skip blank lines */
    while ((len = getline(&line, &lcap, f)) >= 0) {
        char *p = line; n = 0;
        for (;;) {
            char *e; long v = strtol(p, &e, 10);
            if (e == p) break;
            if (n >= *cap) { *cap = *cap ? *cap * 2 : 64; *out = (int *)realloc(*out, sizeof(int) * (size_t)*cap); }
            (*out)[n++] = (int)v; p = e;
        }
        if (n > 0) break;
    }
    free(line);
    return len < 0 && n == 0 ? -1 : n;
}

/* alist: SURVEY Appendix B; codes/ files are 1-based and unpadded, zero padding tolerated */
or_code *or_code_load_alist(const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) return NULL;
    int *buf = NULL, cap = 0, n;
    or_code *c = (or_code *)calloc(1, sizeof(or_code));
    n = read_int_line(f, &buf, &cap);
    if (n < 2) goto fail;
    c->nvar = buf[0]; c->nchk = buf[1];
    n = read_int_line(f, &buf, &cap);
    if (n < 2) goto fail;
    n = read_int_line(f, &buf, &cap);
    if (n != c->nvar) goto fail;
    c->dv = (int *)malloc(sizeof(int) * (size_t)c->nvar);
    memcpy(c->dv, buf, sizeof(int) * (size_t)c->nvar);
    n = read_int_line(f, &buf, &cap);
    if (n != c->nchk) goto fail;
    c->dc = (int *)malloc(sizeof(int) * (size_t)c->nchk);
    memcpy(c->dc, buf, sizeof(int) * (size_t)c->nchk);
    c->col_ptr = (int *)malloc(sizeof(int) * ((size_t)c->nvar + 1));
    c->row_ptr = (int *)malloc(sizeof(int) * ((size_t)c->nchk + 1));
    c->col_ptr[0] = 0;
    for (int v = 0; v < c->nvar; v++) c->col_ptr[v + 1] = c->col_ptr[v] + c->dv[v];
    c->row_ptr[0] = 0;
    for (int r = 0; r < c->nchk; r++) c->row_ptr[r + 1] = c->row_ptr[r] + c->dc[r];
    c->nedges = c->col_ptr[c->nvar];
    if (c->row_ptr[c->nchk] != c->nedges) goto fail;
    c->col_idx = (int *)malloc(sizeof(int) * (size_t)c->nedges);
    c->row_idx = (int *)malloc(sizeof(int) * (size_t)c->nedges);
    for (int v = 0; v < c->nvar; v++) {
        n = read_int_line(f, &buf, &cap);
        int k = 0;
        for (int i = 0; i < n; i++) if (buf[i] != 0) {
            if (k >= c->dv[v] || buf[i] < 1 || buf[i] > c->nchk) goto fail;
            c->col_idx[c->col_ptr[v] + k++] = buf[i] - 1;
        }
        if (k != c->dv[v]) goto fail;
        qsort(c->col_idx + c->col_ptr[v], (size_t)k, sizeof(int), cmp_int);
    }
    for (int r = 0; r < c->nchk; r++) {
        n = read_int_line(f, &buf, &cap);
        int k = 0;
        for (int i = 0; i < n; i++) if (buf[i] != 0) {
            if (k >= c->dc[r] || buf[i] < 1 || buf[i] > c->nvar) goto fail;
            c->row_idx[c->row_ptr[r] + k++] = buf[i] - 1;
        }
        if (k != c->dc[r]) goto fail;
        qsort(c->row_idx + c->row_ptr[r], (size_t)k, sizeof(int), cmp_int);
    }
    free(buf); fclose(f);
    return c;
fail:
    free(buf); fclose(f); or_code_free(c);
    return NULL;
}

void or_code_free(or_code *c)
{
    if (!c) return;
    free(c->dv); free(c->dc); free(c->col_ptr); free(c->col_idx); free(c->row_ptr); free(c->row_idx);
    free(c);
}

/* LDPC_Code_LUT.cpp:513-527: edges are numbered VN-major (rows ascending inside a VN);
 * cn_msg_idx lists, check by check, the ids of its edges in the order they were numbered,
 * i.e. ascending VN index. */
void or_code_cn_msg_idx(const or_code *c, int *cn_msg_idx)
{
    int *fill = (int *)calloc((size_t)c->nchk, sizeof(int));
    int e = 0;
    for (int v = 0; v < c->nvar; v++)
        for (int k = c->col_ptr[v]; k < c->col_ptr[v + 1]; k++) {
            int r = c->col_idx[k];
            cn_msg_idx[c->row_ptr[r] + fill[r]++] = e++;
        }
    free(fill);
}

/* GF(2) rank.  Stage 1 peels pivots that create no fill (a column with a single entry among
 * the live rows); stage 2 runs dense bit-packed elimination on what is left. */
int or_code_gf2_rank(const or_code *c)
{
    int N = c->nvar, M = c->nchk, rank = 0;
    int *colw = (int *)malloc(sizeof(int) * (size_t)N);
    unsigned char *row_dead = (unsigned char *)calloc((size_t)M, 1), *col_dead = (unsigned char *)calloc((size_t)N, 1);
    int *stack = (int *)malloc(sizeof(int) * (size_t)N), sp = 0;
    for (int v = 0; v < N; v++) { colw[v] = c->dv[v]; if (colw[v] == 1) stack[sp++] = v; }
    while (sp > 0) {
        int v = stack[--sp];
        if (col_dead[v] || colw[v] != 1) continue;
        int r = -1;
        for (int k = c->col_ptr[v]; k < c->col_ptr[v + 1]; k++) if (!row_dead[c->col_idx[k]]) { r = c->col_idx[k]; break; }
        if (r < 0) continue;
        rank++; row_dead[r] = 1; col_dead[v] = 1;
        for (int k = c->row_ptr[r]; k < c->row_ptr[r + 1]; k++) {
            int u = c->row_idx[k];
            if (col_dead[u]) continue;
            if (--colw[u] == 1) stack[sp++] = u;
        }
    }
    /* residual */
    int *rmap = (int *)malloc(sizeof(int) * (size_t)M), *cmap = (int *)malloc(sizeof(int) * (size_t)N);
    int Mr = 0, Nr = 0;
    for (int r = 0; r < M; r++) rmap[r] = row_dead[r] ? -1 : Mr++;
    for (int v = 0; v < N; v++) cmap[v] = (col_dead[v] || colw[v] == 0) ? -1 : Nr++;
    if (Mr > 0 && Nr > 0) {
        size_t W = ((size_t)Nr + 63) / 64;
        uint64_t *A = (uint64_t *)calloc((size_t)Mr * W, sizeof(uint64_t));
        for (int r = 0; r < M; r++) {
            if (rmap[r] < 0) continue;
            for (int k = c->row_ptr[r]; k < c->row_ptr[r + 1]; k++) {
                int u = cmap[c->row_idx[k]];
                if (u >= 0) A[(size_t)rmap[r] * W + (size_t)(u >> 6)] ^= 1ull << (u & 63);
            }
        }
        int prow = 0;
        for (int col = 0; col < Nr && prow < Mr; col++) {
            size_t w = (size_t)(col >> 6); uint64_t bit = 1ull << (col & 63);
            int p = -1;
            for (int r = prow; r < Mr; r++) if (A[(size_t)r * W + w] & bit) { p = r; break; }
            if (p < 0) continue;
            if (p != prow)
                for (size_t j = w; j < W; j++) { uint64_t t = A[(size_t)p * W + j]; A[(size_t)p * W + j] = A[(size_t)prow * W + j]; A[(size_t)prow * W + j] = t; }
            for (int r = prow + 1; r < Mr; r++)
                if (A[(size_t)r * W + w] & bit)
                    for (size_t j = w; j < W; j++) A[(size_t)r * W + j] ^= A[(size_t)prow * W + j];
            prow++;
        }
        rank += prow;
        free(A);
    }
    free(colw); free(row_dead); free(col_dead); free(stack); free(rmap); free(cmap);
    return rank;
}

/* LDPC_Ensemble.cpp:53-132 (set_*_degree_dist + check_consistency normalisation) */
static void dist_from_edge_pmf(const double *pmf, int len, int *n_act, int **deg, double **w)
{
    int act = 0;
    for (int i = 0; i < len; i++) if (pmf[i] > 0) act++;
    *deg = (int *)malloc(sizeof(int) * (size_t)(act ? act : 1));
    *w = (double *)malloc(sizeof(double) * (size_t)(act ? act : 1));
    int k = 0;
    for (int i = 0; i < len; i++) if (pmf[i] > 0) { (*deg)[k] = i + 1; (*w)[k] = pmf[i]; k++; }
    double s = 0;
    for (int i = 0; i < act; i++) s += (*w)[i];
    for (int i = 0; i < act; i++) (*w)[i] = (*w)[i] / s;
    *n_act = act;
}

/* LDPC_Ensemble.cpp:391-423 */
or_ensemble *or_empirical_ensemble(const or_code *c)
{
    enum { max_degree = 200 };
    double var_edge[max_degree] = {0}, chk_edge[max_degree] = {0};
    for (int v = 0; v < c->nvar; v++) var_edge[c->dv[v] - 1] += c->dv[v];
    for (int r = 0; r < c->nchk; r++) chk_edge[c->dc[r] - 1] += c->dc[r];
    double sv = 0, sc = 0;
    for (int i = 0; i < max_degree; i++) sv += var_edge[i];
    for (int i = 0; i < max_degree; i++) sc += chk_edge[i];
    for (int i = 0; i < max_degree; i++) { var_edge[i] = var_edge[i] / sv; chk_edge[i] = chk_edge[i] / sc; }
    or_ensemble *e = (or_ensemble *)calloc(1, sizeof(or_ensemble));
    dist_from_edge_pmf(chk_edge, max_degree, &e->dc_act, &e->degree_rho, &e->rho);
    dist_from_edge_pmf(var_edge, max_degree, &e->dv_act, &e->degree_lam, &e->lam);
    return e;
}

/* LDPC_Ensemble.cpp:134-148 (explicit degrees, as read from an .ens file) */
or_ensemble *or_ensemble_from_edge_dist(const int *dl, const double *l, int nl, const int *dr, const double *r, int nr)
{
    or_ensemble *e = (or_ensemble *)calloc(1, sizeof(or_ensemble));
    e->dv_act = nl; e->dc_act = nr;
    e->degree_lam = (int *)malloc(sizeof(int) * (size_t)nl); e->lam = (double *)malloc(sizeof(double) * (size_t)nl);
    e->degree_rho = (int *)malloc(sizeof(int) * (size_t)nr); e->rho = (double *)malloc(sizeof(double) * (size_t)nr);
    double sl = 0, sr = 0;
    for (int i = 0; i < nl; i++) { e->degree_lam[i] = dl[i]; sl += l[i]; }
    for (int i = 0; i < nr; i++) { e->degree_rho[i] = dr[i]; sr += r[i]; }
    for (int i = 0; i < nl; i++) e->lam[i] = l[i] / sl;
    for (int i = 0; i < nr; i++) e->rho[i] = r[i] / sr;
    return e;
}

double or_ensemble_rate(const or_ensemble *e)
{
    double a = 0, b = 0;
    for (int i = 0; i < e->dc_act; i++) a += e->rho[i] / e->degree_rho[i];
    for (int i = 0; i < e->dv_act; i++) b += e->lam[i] / e->degree_lam[i];
    return 1 - a / b;
}

void or_ensemble_free(or_ensemble *e)
{
    if (!e) return;
    free(e->degree_lam); free(e->degree_rho); free(e->lam); free(e->rho); free(e);
}
