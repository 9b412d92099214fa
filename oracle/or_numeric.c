/*
 * or_numeric.c -- ORACLE (test infrastructure): restatement of src/common.cpp and the
 * pmf-domain min-sum of src/LDPC_DE.cpp.  Floating-point operation order follows the
 * reference statement by statement, because the designed LUTs depend on it.
 */
#include "oracle.h"
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

or_dvec or_dvec_new(int n)
{
    or_dvec a; a.n = n; a.v = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); return a;
}
or_dvec or_dvec_copy(or_dvec a)
{
    or_dvec b = or_dvec_new(a.n); if (a.n > 0) memcpy(b.v, a.v, sizeof(double) * (size_t)a.n); return b;
}
void or_dvec_free(or_dvec *a) { free(a->v); a->v = NULL; a->n = 0; }
or_ivec or_ivec_new(int n)
{
    or_ivec a; a.n = n; a.v = (int *)calloc((size_t)(n > 0 ? n : 1), sizeof(int)); return a;
}
or_ivec or_ivec_copy(or_ivec a)
{
    or_ivec b = or_ivec_new(a.n); if (a.n > 0) memcpy(b.v, a.v, sizeof(int) * (size_t)a.n); return b;
}
void or_ivec_free(or_ivec *a) { free(a->v); a->v = NULL; a->n = 0; }
void or_free(void *p) { free(p); }

/* IT++ sum(vec): plain left-to-right accumulation */
static double dsum(const double *v, int n)
{
    double s = 0; for (int i = 0; i < n; i++) s += v[i]; return s;
}

/* IT++ Qfunc (itpp/base/math/error.h): 0.5*erfc(x/1.41421356237310) -- the truncated sqrt(2)
 * literal is IT++'s, kept on purpose. */
double or_qfunc(double x) { return 0.5 * erfc(x / 1.41421356237310); }

/* common.cpp:140-149 */
or_dvec or_gaussian_pmf(double mu, double sig, int N, double delta)
{
    or_dvec pmf = or_dvec_new(N);
    pmf.v[0] = 1 - or_qfunc(((-N / 2.0 + 1) * delta - mu) / sig);
    for (int nn = 1; nn < N - 1; nn++)
        pmf.v[nn] = or_qfunc(((nn - N / 2.0) * delta - mu) / sig) -
                    or_qfunc(((nn + 1 - N / 2.0) * delta - mu) / sig);
    pmf.v[N - 1] = or_qfunc(((N / 2.0 - 1) * delta - mu) / sig);
    double s = dsum(pmf.v, N);
    for (int i = 0; i < N; i++) pmf.v[i] = pmf.v[i] / s;
    return pmf;
}

/* common.cpp:180-191 */
static or_dvec kron(or_dvec x, or_dvec y)
{
    or_dvec z = or_dvec_new(x.n * y.n);
    for (int xx = 0; xx < x.n; xx++)
        for (int yy = 0; yy < y.n; yy++) z.v[xx * y.n + yy] = x.v[xx] * y.v[yy];
    return z;
}
static or_dvec fliplr(or_dvec x)
{
    or_dvec y = or_dvec_new(x.n);
    for (int i = 0; i < x.n; i++) y.v[i] = x.v[x.n - 1 - i];
    return y;
}

/* common.cpp:30-39: child 0 ends up least significant */
or_dvec or_var_product_pmf(const or_dvec *p_in, int num_inputs)
{
    or_dvec prod = or_dvec_copy(p_in[num_inputs - 1]);
    for (int ii = num_inputs - 2; ii >= 0; ii--) {
        or_dvec t = kron(prod, p_in[ii]);
        or_dvec_free(&prod);
        prod = t;
    }
    return prod;
}

/* common.cpp:193-228 */
int or_signed_to_unsigned_idx(int idx, const int *inres, int num_in)
{
    int out_max = 2;
    for (int j = 0; j < num_in; j++) out_max *= inres[j] / 2;
    int parity = 0, idx_out = 0, base = 1, tmp = idx;
    for (int j = 0; j < num_in; j++) {
        int lab = tmp % inres[j];
        tmp /= inres[j];
        if (lab < inres[j] / 2) { parity ^= 1; idx_out += base * (inres[j] / 2 - 1 - lab); }
        else idx_out += base * (lab - inres[j] / 2);
        base *= inres[j] / 2;
    }
    return parity == 0 ? out_max - 1 - idx_out : idx_out;
}

/* common.cpp:41-70 */
or_dvec or_chk_product_pmf(const or_dvec *p_in, int num_inputs)
{
    int *res = (int *)malloc(sizeof(int) * (size_t)num_inputs);
    for (int j = 0; j < num_inputs; j++) res[j] = p_in[j].n;
    or_dvec p0 = or_dvec_copy(p_in[num_inputs - 1]);
    or_dvec p1 = fliplr(p_in[num_inputs - 1]);
    for (int ii = num_inputs - 2; ii >= 0; ii--) {
        or_dvec f = fliplr(p_in[ii]);
        or_dvec a = kron(p0, p_in[ii]), b = kron(p1, f);
        or_dvec c = kron(p1, p_in[ii]), d = kron(p0, f);
        or_dvec n0 = or_dvec_new(a.n), n1 = or_dvec_new(a.n);
        for (int i = 0; i < a.n; i++) {
            n0.v[i] = .5 * (a.v[i] + b.v[i]);
            n1.v[i] = .5 * (c.v[i] + d.v[i]);
        }
        or_dvec_free(&a); or_dvec_free(&b); or_dvec_free(&c); or_dvec_free(&d); or_dvec_free(&f);
        or_dvec_free(&p0); or_dvec_free(&p1);
        p0 = n0; p1 = n1;
    }
    int out_len = 2;
    for (int j = 0; j < num_inputs; j++) out_len *= res[j] / 2;
    or_dvec comb = or_dvec_new(out_len);
    for (int mm = 0; mm < p0.n; mm++)
        comb.v[or_signed_to_unsigned_idx(mm, res, num_inputs)] += p0.v[mm];
    or_dvec_free(&p0); or_dvec_free(&p1); free(res);
    return comb;
}

/* common.cpp:120-129 */
int or_quant_nonlin(double x, const double *bounds, int nb)
{
    int idx = 0;
    for (int i = 0; i < nb; i++) { if (x > bounds[i]) idx++; else break; }
    return idx;
}

/* common.cpp:162-167 */
static double x_log2_y(double x, double y)
{
    if (x == 0) return 0;
    if (x > 0 && y > 0) return x * log2(y);
    fprintf(stderr, "oracle: x_log2_y(): input invalid (%g,%g)\n", x, y);
    abort();
}

/* common.cpp:371-380 */
static double mi_bcpmf_sym(or_dvec p)
{
    int K = p.n; double mi = 0;
    for (int i = 0; i < K / 2; i++)
        mi += p.v[i] * log2(2 * p.v[i] / (p.v[i] + p.v[K - 1 - i])) +
              p.v[K - 1 - i] * log2(2 * p.v[K - 1 - i] / (p.v[K - 1 - i] + p.v[i]));
    return mi;
}

/* index sort, ascending by key then by index (IT++ sort_index followed by the tie fix-up of
 * common.cpp:338-343 is equivalent to this total order) */
static const double *g_sort_key;
static int cmp_key_idx(const void *a, const void *b)
{
    int ia = *(const int *)a, ib = *(const int *)b;
    double ka = g_sort_key[ia], kb = g_sort_key[ib];
    if (ka < kb) return -1;
    if (ka > kb) return 1;
    return (ia > ib) - (ia < ib);
}

/* common.cpp:333-369 */
or_dvec or_sym_llr_sort_unique(or_dvec p_in, or_ivec *idx_in, or_ivec *idx_sorted, double llr_delta)
{
    int M_in = p_in.n;
    double *llr = (double *)malloc(sizeof(double) * (size_t)M_in);
    for (int i = 0; i < M_in; i++) {
        llr[i] = log(p_in.v[i]) - log(p_in.v[M_in - 1 - i]);
        /* a label pair without any mass gives -inf - -inf = NaN; the reference then sorts NaNs with
         * a comparison sort (undefined order).  Pinned here: such a pair carries LLR 0, which keeps
         * the sorted order symmetric.  Only check-node LUT design can reach this (LUT_Tree.cpp:761
         * does not strip zero-mass labels, unlike :726-728). */
        if (isnan(llr[i])) llr[i] = 0.0;
    }
    *idx_in = or_ivec_new(M_in);
    for (int i = 0; i < M_in; i++) idx_in->v[i] = i;
    g_sort_key = llr;
    qsort(idx_in->v, (size_t)M_in, sizeof(int), cmp_key_idx);

    or_ivec half = or_ivec_new(M_in / 2);
    half.v[0] = 0;
    double dupl = llr[idx_in->v[0]];
    int dupl_idx = 0, num_dupl = 0;
    for (int mm = 1; mm < M_in / 2; mm++) {
        if (fabs(llr[idx_in->v[mm]] - dupl) <= llr_delta) num_dupl++;
        else dupl_idx++;
        half.v[mm] = dupl_idx;
        dupl = llr[idx_in->v[mm]];
    }
    int mx = 0;
    for (int i = 0; i < half.n; i++) if (half.v[i] > mx) mx = half.v[i];
    *idx_sorted = or_ivec_new(M_in);
    for (int i = 0; i < M_in / 2; i++) {
        idx_sorted->v[i] = half.v[i];
        idx_sorted->v[M_in / 2 + i] = 2 * mx + 1 - half.v[M_in / 2 - 1 - i];
    }
    int M = M_in - 2 * num_dupl;
    or_dvec p_sorted = or_dvec_new(M);
    for (int mm = 0; mm < M_in; mm++) p_sorted.v[idx_sorted->v[mm]] += p_in.v[idx_in->v[mm]];
    or_ivec_free(&half);
    free(llr);
    return p_sorted;
}

/* common.cpp:230-331 */
double or_quant_mi_sym(or_dvec *p_out, or_ivec *Q_out, or_dvec p_in, int Nq, int sorted)
{
    int K = Nq, M_in = p_in.n, M;
    if (M_in % 2 != 0 || K % 2 != 0) { fprintf(stderr, "oracle: quant_mi_sym(): odd sizes\n"); abort(); }
    or_dvec p_sorted; or_ivec idx_in, idx_sorted;
    if (!sorted) {
        p_sorted = or_sym_llr_sort_unique(p_in, &idx_in, &idx_sorted, 0.0);
        M = p_sorted.n;
    } else {
        idx_in = or_ivec_new(M_in); idx_sorted = or_ivec_new(M_in);
        for (int i = 0; i < M_in; i++) { idx_in.v[i] = i; idx_sorted.v[i] = i; }
        p_sorted = or_dvec_copy(p_in);
        M = M_in;
    }
    *Q_out = or_ivec_new(M_in);
    *p_out = or_dvec_new(K);
    double ret;

    if (K >= M) { /* trivial case, :257-272 */
        int outlabel = 0;
        for (int mm = 0; mm < M_in / 2; mm++) {
            if (idx_sorted.v[mm] > outlabel) outlabel++;
            Q_out->v[idx_in.v[M_in - 1 - mm]] = K - 1 - outlabel;
            Q_out->v[idx_in.v[mm]] = outlabel;
        }
        for (int mm = 0; mm < M_in; mm++) p_out->v[Q_out->v[mm]] += p_in.v[mm];
        ret = mi_bcpmf_sym(p_in);
        goto done;
    }
    {
        int H = M / 2, Kh = K / 2;
        double *g = (double *)calloc((size_t)H * (size_t)H, sizeof(double));
        for (int ap = 0; ap < H; ap++) {
            double p_plus = 0, p_minus = 0;
            for (int a = ap; a < H; a++) {
                p_plus += p_sorted.v[H + a];
                p_minus += p_sorted.v[H - 1 - a];
                g[(size_t)ap * H + a] = x_log2_y(p_plus, 2 * p_plus / (p_plus + p_minus)) +
                                        x_log2_y(p_minus, 2 * p_minus / (p_plus + p_minus));
            }
        }
        double *S = (double *)calloc((size_t)H * (size_t)Kh, sizeof(double));
        int *h = (int *)calloc((size_t)H * (size_t)Kh, sizeof(int));
        for (int a = 0; a < (M - K) / 2 + 1; a++) S[(size_t)a * Kh + 0] = g[a];
        for (int zz = 1; zz < Kh; zz++) {
            for (int a = zz; a < zz + (M - K) / 2 + 1; a++) {
                S[(size_t)a * Kh + zz] = -DBL_MAX;
                for (int ap = zz; ap <= a; ap++) {
                    double t = S[(size_t)(ap - 1) * Kh + (zz - 1)] + g[(size_t)ap * H + a];
                    if (t > S[(size_t)a * Kh + zz]) { S[(size_t)a * Kh + zz] = t; h[(size_t)a * Kh + zz] = ap; }
                }
            }
        }
        int *astar = (int *)calloc((size_t)Kh + 1, sizeof(int));
        astar[Kh] = H;
        for (int kk = Kh - 1; kk > 0; kk--) astar[kk] = h[(size_t)(astar[kk + 1] - 1) * Kh + kk];
        int outlabel = 0;
        for (int mm = 0; mm < M_in / 2; mm++) {
            if (idx_sorted.v[mm + M_in / 2] - H >= astar[outlabel + 1]) outlabel++;
            Q_out->v[idx_in.v[M_in / 2 + mm]] = Kh + outlabel;
            Q_out->v[idx_in.v[M_in / 2 - 1 - mm]] = Kh - 1 - outlabel;
        }
        for (int mm = 0; mm < M_in; mm++) p_out->v[Q_out->v[mm]] += p_in.v[mm];
        ret = S[(size_t)(H - 1) * Kh + (Kh - 1)];
        free(g); free(S); free(h); free(astar);
    }
done:
    or_dvec_free(&p_sorted); or_ivec_free(&idx_in); or_ivec_free(&idx_sorted);
    return ret;
}

/* LDPC_DE.cpp:1061-1121 (pmf_plus / pmf_minus / pmf_join inlined) */
or_dvec or_chk_update_minsum_pmf(or_dvec p_in, int dc)
{
    int N = p_in.n, H = N / 2;
    double *a_p = (double *)malloc(sizeof(double) * (size_t)H), *a_m = (double *)malloc(sizeof(double) * (size_t)H);
    double *b_p = (double *)malloc(sizeof(double) * (size_t)H), *b_m = (double *)malloc(sizeof(double) * (size_t)H);
    double *c_p = (double *)calloc((size_t)H, sizeof(double)), *c_m = (double *)calloc((size_t)H, sizeof(double));
    for (int nn = 0; nn < H; nn++) {
        a_p[nn] = p_in.v[H + nn] + p_in.v[H - 1 - nn];
        a_m[nn] = p_in.v[H + nn] - p_in.v[H - 1 - nn];
        b_p[nn] = a_p[nn]; b_m[nn] = a_m[nn];
    }
    for (int dd = 1; dd < dc - 1; dd++) {
        for (int i = 0; i < H; i++) { c_p[i] = 0; c_m[i] = 0; }
        for (int ii = 0; ii < H; ii++)
            for (int jj = 0; jj < H; jj++) {
                int kk = ii < jj ? ii : jj;
                c_p[kk] += a_p[ii] * b_p[jj];
                c_m[kk] += a_m[ii] * b_m[jj];
            }
        for (int i = 0; i < H; i++) { a_p[i] = c_p[i]; a_m[i] = c_m[i]; }
    }
    or_dvec out = or_dvec_new(N);
    for (int nn = 0; nn < H; nn++) {
        out.v[H + nn] = .5 * (c_p[nn] + c_m[nn]);
        out.v[H - 1 - nn] = .5 * (c_p[nn] - c_m[nn]);
    }
    free(a_p); free(a_m); free(b_p); free(b_m); free(c_p); free(c_m);
    return out;
}
