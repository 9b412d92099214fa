#define _GNU_SOURCE
/*
 * or_sim.c -- ORACLE (test infrastructure): the Monte-Carlo frame loop of
 * LDPC_BER_Sim::sim_snr_point / run (src/LDPC_BER_Sim.cpp:121-155,246-311), frame by frame like
 * the reference, on top of the oracle decoder.
 *
 * The reference's random stream (IT++ RNG, randb, AWGN_Channel) is not reproducible without the
 * IT++ fork (PARITY UNPINNED for seed-identical counters against the real ber_sim).
 * What is checked here is the BUILD's own front-end specification, restated independently:
 *   - Philox4x32-10, key = seed, counter = (frame lo, frame hi, code-bit pair, stream);
 *     words (0,1) -> 64-bit uniform of code bit 2p, words (2,3) -> code bit 2p+1;
 *   - the received value is represented by its cell in the partition of the real line by
 *     {qb_Cha*N0/4} u {qb_Msg*N0/4 (CONT mode)} u {0}; cell j is chosen as the number of
 *     cumulative thresholds thr[k] = floor(2^64 * P(x <= t_k | +1 sent)) not exceeding the uniform;
 *   - a sent 1 mirrors the cell (labels K-1-l, slicer sign flipped);
 *   - data bits of frame f: Philox with stream | 0x80000000, counter word 2 = k / 128, bit k % 128.
 */
#include "or_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t w[4]; } px_out;

static px_out philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; r++) {
        uint64_t a = (uint64_t)0xD2511F53u * c0, b = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(b >> 32) ^ c1 ^ k0, n1 = (uint32_t)b, n2 = (uint32_t)(a >> 32) ^ c3 ^ k1, n3 = (uint32_t)a;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    px_out o = {{c0, c1, c2, c3}};
    return o;
}

typedef struct {
    int n_cells;
    uint64_t thr[80];
    uint8_t cha[80], msg[80], neg[80], cha_m[80], msg_m[80];
} or_cells;

static int cmp_dbl(const void *a, const void *b) { double x = *(const double *)a, y = *(const double *)b; return (x > y) - (x < y); }

static void make_cells(or_cells *C, double N0, const double *qb_cha, int n_cha, const double *qb_msg, int n_msg, int mode, const int *map)
{
    double t[80]; int nt = 0;
    for (int i = 0; i < n_cha; i++) t[nt++] = qb_cha[i] * N0 / 4;
    if (mode == 0) for (int i = 0; i < n_msg; i++) t[nt++] = qb_msg[i] * N0 / 4;
    t[nt++] = 0.0;
    qsort(t, (size_t)nt, sizeof(double), cmp_dbl);
    int nu = 0;
    for (int i = 0; i < nt; i++) if (nu == 0 || t[i] != t[nu - 1]) t[nu++] = t[i];
    const double sigma = sqrt(N0 / 2), two64 = 18446744073709551616.0;
    int Kc = n_cha + 1, Km = n_msg + 1;
    C->n_cells = nu + 1;
    for (int j = 0; j <= nu; j++) {
        /* labels of the cell (t[j-1], t[j]]: boundaries strictly below it */
        int lc = 0, lm = 0;
        for (int i = 0; i < n_cha; i++) if (j > 0 && qb_cha[i] * N0 / 4 <= t[j - 1]) lc++;
        if (mode == 0) for (int i = 0; i < n_msg; i++) if (j > 0 && qb_msg[i] * N0 / 4 <= t[j - 1]) lm++;
        C->cha[j] = (uint8_t)lc;
        C->cha_m[j] = (uint8_t)(Kc - 1 - lc);
        C->msg[j] = (uint8_t)(mode == 0 ? lm : map[lc]);
        C->msg_m[j] = (uint8_t)(mode == 0 ? Km - 1 - lm : map[Kc - 1 - lc]);
        C->neg[j] = (uint8_t)(j < nu && t[j] <= 0.0);
        if (j == nu) break;
        double z = (t[j] - 1.0) / sigma;
        uint64_t thr;
        if (z <= 0) { double w = 0.5 * erfc(-z * 0.70710678118654752440) * two64; thr = w >= two64 ? UINT64_MAX : (uint64_t)w; }
        else { double w = 0.5 * erfc(z * 0.70710678118654752440) * two64; thr = UINT64_MAX - (w >= two64 ? UINT64_MAX : (uint64_t)w); }
        if (j > 0 && thr < C->thr[j - 1]) thr = C->thr[j - 1];
        C->thr[j] = thr;
    }
}

static void sample_frame(const or_cells *C, uint64_t seed, uint32_t stream, uint64_t frame, int N, const uint8_t *cw,
                         int *cha, int *msg, int *uncoded_errors)
{
    int unc = 0;
    for (int p = 0; 2 * p < N; p++) {
        px_out o = philox4x32_10((uint32_t)frame, (uint32_t)(frame >> 32), (uint32_t)p, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
        uint64_t u[2] = {((uint64_t)o.w[1] << 32) | o.w[0], ((uint64_t)o.w[3] << 32) | o.w[2]};
        for (int h = 0; h < 2; h++) {
            int v = 2 * p + h;
            if (v >= N) break;
            int cell = 0;
            while (cell < C->n_cells - 1 && u[h] >= C->thr[cell]) cell++;
            int bit = cw ? cw[v] : 0;
            cha[v] = bit ? C->cha_m[cell] : C->cha[cell];
            msg[v] = bit ? C->msg_m[cell] : C->msg[cell];
            int slicer = bit ? !C->neg[cell] : C->neg[cell];
            unc += slicer != bit;
        }
    }
    *uncoded_errors = unc;
}

void or_sim_info_bits(uint64_t seed, uint32_t stream, uint64_t frame, int K, uint8_t *out)
{
    for (int k = 0; k < K; k++) {
        px_out o = philox4x32_10((uint32_t)frame, (uint32_t)(frame >> 32), (uint32_t)(k / 128), stream | 0x80000000u, (uint32_t)seed, (uint32_t)(seed >> 32));
        out[k] = (uint8_t)((o.w[(k % 128) / 32] >> (k % 32)) & 1u);
    }
}

/* labels of frames frame0..frame0+B-1 (frame-major), for comparing the device sampler */
void or_sim_sample_labels(const or_codec *c, double snr_db, double rate, uint64_t seed, uint32_t stream, uint64_t frame0, int B,
                          const uint8_t *codewords, uint8_t *cha_out, uint8_t *msg_out, int *uncoded)
{
    double N0 = pow(10.0, -snr_db / 10.0) / rate;
    or_cells C;
    make_cells(&C, N0, c->qb_Cha.v, c->qb_Cha.n, c->qb_Msg.v, c->qb_Msg.n, c->initial_message_mode, c->Nq_Cha_2_Nq_Msg_map.v);
    int *cha = (int *)malloc(sizeof(int) * (size_t)c->nvar), *msg = (int *)malloc(sizeof(int) * (size_t)c->nvar);
    for (int f = 0; f < B; f++) {
        int unc;
        sample_frame(&C, seed, stream, frame0 + (uint64_t)f, c->nvar, codewords ? codewords + (size_t)f * c->nvar : NULL, cha, msg, &unc);
        for (int v = 0; v < c->nvar; v++) { cha_out[(size_t)f * c->nvar + v] = (uint8_t)cha[v]; msg_out[(size_t)f * c->nvar + v] = (uint8_t)msg[v]; }
        if (uncoded) uncoded[f] = unc;
    }
    free(cha); free(msg);
}

/* sim_snr_point, src/LDPC_BER_Sim.cpp:246-311.  codewords: [Nframes][nvar] sent bits or NULL (zero
 * codeword).  counters = {frames, data bits, frame errors, data bit errors, uncoded bit errors}.
 * per_frame (optional, [Nframes][4]) receives {iters, frame error, bit errors, uncoded errors}.
 * Returns the sweep's exit flag (BER < ber_min || FER < fer_min). */
int or_sim_snr_point(or_codec *c, double snr_db, double rate, int K, uint64_t seed, uint32_t stream, int64_t Nframes, int Nfers,
                     double ber_min, double fer_min, const uint8_t *codewords, int64_t *counters, int32_t *per_frame)
{
    double N0 = pow(10.0, -snr_db / 10.0) / rate;
    or_cells C;
    make_cells(&C, N0, c->qb_Cha.v, c->qb_Cha.n, c->qb_Msg.v, c->qb_Msg.n, c->initial_message_mode, c->Nq_Cha_2_Nq_Msg_map.v);
    int N = c->nvar;
    int *cha = (int *)malloc(sizeof(int) * (size_t)N), *msg = (int *)malloc(sizeof(int) * (size_t)N);
    uint8_t *out = (uint8_t *)malloc((size_t)N);
    int64_t frames = 0, databits = 0, ferr = 0, berr = 0, uerr = 0;
    for (int64_t ff = 0; ff < Nframes; ff++) {
        const uint8_t *cw = codewords ? codewords + (size_t)ff * N : NULL;
        int unc;
        sample_frame(&C, seed, stream, (uint64_t)ff, N, cw, cha, msg, &unc);
        int it = or_codec_lut_decode(c, cha, msg, out);
        int be = 0;
        for (int k = 0; k < K; k++) be += out[k] != (cw ? cw[k] : 0);
        frames++; databits += K; berr += be; uerr += unc; ferr += be > 0;
        if (per_frame) { per_frame[ff * 4] = it; per_frame[ff * 4 + 1] = be > 0; per_frame[ff * 4 + 2] = be; per_frame[ff * 4 + 3] = unc; }
        if (ferr > Nfers) break;                 /* :289 */
    }
    counters[0] = frames; counters[1] = databits; counters[2] = ferr; counters[3] = berr; counters[4] = uerr;
    free(cha); free(msg); free(out);
    double ber = databits ? (double)berr / (double)databits : 0, fer = frames ? (double)ferr / (double)frames : 0;
    return ber < ber_min || fer < fer_min;
}

/* a code from explicit graph arrays (the product permutes H's columns when it builds a generator) */
or_code *or_code_from_graph(int nvar, int nchk, const int *dv, const int *dc, const int *cn_msg_idx)
{
    or_code *c = (or_code *)calloc(1, sizeof(or_code));
    c->nvar = nvar; c->nchk = nchk;
    c->dv = (int *)malloc(sizeof(int) * (size_t)nvar); memcpy(c->dv, dv, sizeof(int) * (size_t)nvar);
    c->dc = (int *)malloc(sizeof(int) * (size_t)nchk); memcpy(c->dc, dc, sizeof(int) * (size_t)nchk);
    c->col_ptr = (int *)malloc(sizeof(int) * ((size_t)nvar + 1)); c->row_ptr = (int *)malloc(sizeof(int) * ((size_t)nchk + 1));
    c->col_ptr[0] = c->row_ptr[0] = 0;
    for (int v = 0; v < nvar; v++) c->col_ptr[v + 1] = c->col_ptr[v] + dv[v];
    for (int r = 0; r < nchk; r++) c->row_ptr[r + 1] = c->row_ptr[r] + dc[r];
    c->nedges = c->col_ptr[nvar];
    c->col_idx = (int *)malloc(sizeof(int) * (size_t)c->nedges); c->row_idx = (int *)malloc(sizeof(int) * (size_t)c->nedges);
    int *edge_vn = (int *)malloc(sizeof(int) * (size_t)c->nedges);
    for (int v = 0; v < nvar; v++) for (int e = c->col_ptr[v]; e < c->col_ptr[v + 1]; e++) edge_vn[e] = v;
    int *fill = (int *)calloc((size_t)nvar, sizeof(int));
    for (int r = 0; r < nchk; r++)
        for (int k = c->row_ptr[r]; k < c->row_ptr[r + 1]; k++) {
            int e = cn_msg_idx[k], v = edge_vn[e];
            c->row_idx[k] = v;
            c->col_idx[e] = r;        /* edge e is the (e - col_ptr[v])-th entry of column v */
            fill[v]++;
        }
    free(fill); free(edge_vn);
    return c;
}

/* the cell table for tests: returns n_cells */
int or_sim_channel_cells(const or_codec *c, double snr_db, double rate, uint64_t *thr, uint8_t *cha, uint8_t *msg, uint8_t *neg, uint8_t *cha_m, uint8_t *msg_m)
{
    or_cells C;
    make_cells(&C, pow(10.0, -snr_db / 10.0) / rate, c->qb_Cha.v, c->qb_Cha.n, c->qb_Msg.v, c->qb_Msg.n, c->initial_message_mode, c->Nq_Cha_2_Nq_Msg_map.v);
    for (int j = 0; j < C.n_cells; j++) {
        if (j < C.n_cells - 1) thr[j] = C.thr[j];
        cha[j] = C.cha[j]; msg[j] = C.msg[j]; neg[j] = C.neg[j]; cha_m[j] = C.cha_m[j]; msg_m[j] = C.msg_m[j];
    }
    return C.n_cells;
}

/* [BP] front end: BPSK over AWGN in double precision, Philox-addressed per (seed, stream = SNR index, frame, bit pair), one
 * Box-Muller pair per block (the specification of lut_ldpc_amd/csrc/host/ber_sim_driver.hpp: awgn_llr_frames; restates
 * src/LDPC_BER_Sim.cpp:270-283 with this build's RNG -- the IT++ RNG is absent, PARITY UNPINNED).  llr[B][N], uncoded[B]. */
void or_sim_awgn_llr(uint64_t seed, uint32_t stream, uint64_t frame0, int B, int N, double N0, const uint8_t *codewords, double *llr, int *uncoded)
{
    const double sigma = sqrt(N0 / 2), two_pi = 6.283185307179586476925286766559;
    for (int i = 0; i < B; i++) {
        const uint64_t frame = frame0 + (uint64_t)i;
        int unc = 0;
        for (int p = 0; p < (N + 1) / 2; p++) {
            px_out o = philox4x32_10((uint32_t)frame, (uint32_t)(frame >> 32), (uint32_t)p, stream | 0x40000000u, (uint32_t)seed, (uint32_t)(seed >> 32));
            const double u1 = ((double)((((uint64_t)o.w[0] << 32) | o.w[1]) >> 11) + 1.0) * (1.0 / 9007199254740992.0);
            const double u2 = (double)((((uint64_t)o.w[2] << 32) | o.w[3]) >> 11) * (1.0 / 9007199254740992.0);
            const double rad = sqrt(-2.0 * log(u1));
            double sn, cs;
            sincos(two_pi * u2, &sn, &cs);         /* one libm entry point for both, as in the product */
            const double z[2] = { rad * cs, rad * sn };
            for (int k = 0; k < 2 && 2 * p + k < N; k++) {
                const int v = 2 * p + k, bit = codewords ? codewords[(size_t)i * N + v] : 0;
                const double x = (bit ? -1.0 : 1.0) + sigma * z[k];
                llr[(size_t)i * N + v] = 4.0 * x / N0;
                unc += ((x < 0) ? 1 : 0) != bit;
            }
        }
        uncoded[i] = unc;
    }
}
