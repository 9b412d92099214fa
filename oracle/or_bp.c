/*
 * or_bp.c -- ORACLE (test infrastructure): CPU statement of the [BP] comparison decoder.
 *
 * The reference's BP path (src/LDPC_BER_Sim.cpp:157-244) runs itpp::LDPC_Code::bp_decode with a four-parameter
 * LLR_calc_unit of the forked IT++ (mmeidlinger/itpp, branch lut_ldpc, fork of IT++ 4.3.1), which is absent from the
 * reference tree.  PARITY UNPINNED: this file restates the published IT++ 4.3.1 algorithm as specified in
 * include/lut_ldpc_bp.h (the specification both this oracle and the HIP kernels implement); nothing ties it to the fork.
 */
#include "or_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct or_bp {
    const or_code *code;
    int *cn_msg_idx;           /* check by check, the VN-major edge ids */
    int d1, d2, d3, d4, qmax;
    int *table;                /* d2 entries */
    int max_iters, psc, pisc;
};

static int clip(long long x, int qmax) { return x > qmax ? qmax : (x < -qmax ? -qmax : (int)x); }
int or_bp_to_qllr(const or_bp *b, double l)
{
    const double v = floor(0.5 + ldexp(1.0, b->d1) * l);
    if (v >= (double)b->qmax) return b->qmax;
    if (v <= -(double)b->qmax) return -b->qmax;
    return (int)v;
}
static int logexp(const or_bp *b, int x) { const int ind = x >> b->d3; return ind >= b->d2 ? 0 : b->table[ind]; }
static int boxplus(const or_bp *b, int a, int c)
{
    const int aa = a > 0 ? a : -a, ca = c > 0 ? c : -c, mn = aa > ca ? ca : aa;
    const int t1 = a > 0 ? (c > 0 ? mn : -mn) : (c > 0 ? -mn : mn);
    if (b->d2 == 0) return t1;
    const int apb = a + c, amb = a - c;
    return t1 + logexp(b, apb > 0 ? apb : -apb) - logexp(b, amb > 0 ? amb : -amb);
}

or_bp *or_bp_new(const or_code *code, int d1, int d2, int d3, int d4)
{
    or_bp *b = (or_bp *)calloc(1, sizeof(or_bp));
    b->code = code; b->d1 = d1; b->d2 = d2; b->d3 = d3; b->d4 = d4; b->qmax = (int)((1ll << (d4 - 1)) - 1);
    b->cn_msg_idx = (int *)malloc(sizeof(int) * (size_t)code->nedges);
    or_code_cn_msg_idx(code, b->cn_msg_idx);
    b->table = (int *)malloc(sizeof(int) * (size_t)(d2 > 0 ? d2 : 1));
    for (int i = 0; i < d2; i++) b->table[i] = or_bp_to_qllr(b, log(1.0 + exp(-ldexp(1.0, d3 - d1) * i)));
    b->max_iters = 50; b->psc = 1; b->pisc = 0;
    return b;
}
void or_bp_free(or_bp *b) { if (!b) return; free(b->cn_msg_idx); free(b->table); free(b); }
void or_bp_set_exit_conditions(or_bp *b, int max_iters, int psc, int pisc) { b->max_iters = max_iters; b->psc = psc; b->pisc = pisc; }
int or_bp_table(const or_bp *b, int *out) { if (out) memcpy(out, b->table, sizeof(int) * (size_t)b->d2); return b->d2; }

static int syndrome_ok(const or_bp *b, const int *llr)
{
    const or_code *H = b->code;
    for (int c = 0; c < H->nchk; c++) {
        int s = 0;
        for (int k = H->row_ptr[c]; k < H->row_ptr[c + 1]; k++) s ^= llr[H->row_idx[k]] < 0;
        if (s) return 0;
    }
    return 1;
}

/* LDPC_Code::bp_decode on one frame; mvc / mcv: nedges ints of scratch each */
static int decode_one(const or_bp *b, const int *in, int *out, int *mvc, int *mcv)
{
    const or_code *H = b->code;
    int m[256], ml[256], mr[256];
    if (b->pisc && syndrome_ok(b, in)) { memcpy(out, in, sizeof(int) * (size_t)H->nvar); return 0; }
    memcpy(out, in, sizeof(int) * (size_t)H->nvar);
    int e = 0;
    for (int v = 0; v < H->nvar; v++) for (int k = 0; k < H->dv[v]; k++) mvc[e++] = in[v];
    for (int iter = 1; iter <= b->max_iters; iter++) {
        e = 0;
        for (int c = 0; c < H->nchk; c++) {
            const int n = H->dc[c];
            const int *ix = b->cn_msg_idx + e;
            if (n == 2) { mcv[ix[0]] = mvc[ix[1]]; mcv[ix[1]] = mvc[ix[0]]; }
            else if (n <= 6) {
                /* IT++ 4.3.1 spells the check update out for degrees 3..6 with these association orders (boxplus with the table
                 * correction is not associative, so the order is part of the result) */
                int q[6];
                for (int i = 0; i < n; i++) q[i] = mvc[ix[i]];
                if (n == 3) {
                    mcv[ix[0]] = boxplus(b, q[1], q[2]); mcv[ix[1]] = boxplus(b, q[0], q[2]); mcv[ix[2]] = boxplus(b, q[0], q[1]);
                } else if (n == 4) {
                    const int m01 = boxplus(b, q[0], q[1]), m23 = boxplus(b, q[2], q[3]);
                    mcv[ix[0]] = boxplus(b, q[1], m23); mcv[ix[1]] = boxplus(b, q[0], m23);
                    mcv[ix[2]] = boxplus(b, m01, q[3]); mcv[ix[3]] = boxplus(b, m01, q[2]);
                } else if (n == 5) {
                    const int m01 = boxplus(b, q[0], q[1]), m02 = boxplus(b, m01, q[2]), m34 = boxplus(b, q[3], q[4]), m24 = boxplus(b, q[2], m34);
                    mcv[ix[0]] = boxplus(b, q[1], m24); mcv[ix[1]] = boxplus(b, q[0], m24); mcv[ix[2]] = boxplus(b, m01, m34);
                    mcv[ix[3]] = boxplus(b, m02, q[4]); mcv[ix[4]] = boxplus(b, m02, q[3]);
                } else {
                    const int m01 = boxplus(b, q[0], q[1]), m23 = boxplus(b, q[2], q[3]), m45 = boxplus(b, q[4], q[5]);
                    const int m03 = boxplus(b, m01, m23), m25 = boxplus(b, m23, m45), m0145 = boxplus(b, m01, m45);
                    mcv[ix[0]] = boxplus(b, q[1], m25); mcv[ix[1]] = boxplus(b, q[0], m25);
                    mcv[ix[2]] = boxplus(b, m0145, q[3]); mcv[ix[3]] = boxplus(b, m0145, q[2]);
                    mcv[ix[4]] = boxplus(b, m03, q[5]); mcv[ix[5]] = boxplus(b, m03, q[4]);
                }
            }
            else {
                for (int i = 0; i < n; i++) m[i] = mvc[ix[i]];
                ml[0] = m[0]; mr[0] = m[n - 1];
                for (int i = 1; i < n - 1; i++) { ml[i] = boxplus(b, ml[i - 1], m[i]); mr[i] = boxplus(b, mr[i - 1], m[n - 1 - i]); }
                mcv[ix[0]] = mr[n - 2]; mcv[ix[n - 1]] = ml[n - 2];
                for (int i = 1; i < n - 1; i++) mcv[ix[i]] = boxplus(b, ml[i - 1], mr[n - 2 - i]);
            }
            e += n;
        }
        e = 0;
        for (int v = 0; v < H->nvar; v++) {
            long long s = in[v];
            for (int k = 0; k < H->dv[v]; k++) s += mcv[e + k];
            out[v] = clip(s, b->qmax);
            for (int k = 0; k < H->dv[v]; k++) mvc[e + k] = clip(s - mcv[e + k], b->qmax);
            e += H->dv[v];
        }
        if (b->psc && syndrome_ok(b, out)) return iter;
    }
    return -b->max_iters;
}

/* qllr: [B][nvar] ints; out_bits [B][nvar] = LLRout < 0; out_qllr optional */
void or_bp_decode_qllr_batch(const or_bp *b, const int *qllr, int B, uint8_t *out_bits, int32_t *out_iters, int *out_qllr)
{
    const int N = b->code->nvar, E = b->code->nedges;
    int *mvc = (int *)malloc(sizeof(int) * (size_t)E), *mcv = (int *)malloc(sizeof(int) * (size_t)E), *out = (int *)malloc(sizeof(int) * (size_t)N);
    for (int f = 0; f < B; f++) {
        out_iters[f] = decode_one(b, qllr + (size_t)f * N, out, mvc, mcv);
        for (int v = 0; v < N; v++) out_bits[(size_t)f * N + v] = out[v] < 0;
        if (out_qllr) memcpy(out_qllr + (size_t)f * N, out, sizeof(int) * (size_t)N);
    }
    free(mvc); free(mcv); free(out);
}
void or_bp_decode_llr_batch(const or_bp *b, const double *llr, int B, uint8_t *out_bits, int32_t *out_iters, int *out_qllr)
{
    const size_t n = (size_t)B * b->code->nvar;
    int *q = (int *)malloc(sizeof(int) * n);
    for (size_t i = 0; i < n; i++) q[i] = or_bp_to_qllr(b, llr[i]);
    or_bp_decode_qllr_batch(b, q, B, out_bits, out_iters, out_qllr);
    free(q);
}
