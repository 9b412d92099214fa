// bp_decoder.hip -- the [BP] comparison decoder on the MI355X (C-ABI: include/lut_ldpc_bp.h, which holds the specification).
//
// Replaces itpp::LDPC_Code::bp_decode as called by LDPC_BER_Sim (src/LDPC_BER_Sim.cpp:157-244,281) for a BATCH of frames.
// PARITY UNPINNED (the IT++ fork is absent); bit-identical to oracle/or_bp.c.  This is the comparison back-end, not the hot
// path: plain coalesced kernels, one thread per (node, frame), no tuning beyond that.
//
// HBM layout: int32 rows [e][Bpad] for both message directions (mvc, mcv: IT++ keeps two buffers as well), [v][Bpad] for the
// input and output LLRs; a thread block works on 256 consecutive frames, so every access of a wave is 256 contiguous bytes.
// Check pass without per-thread arrays: the left partial sums are parked in the output rows on the way up and combined with
// the running right sum on the way down (5 row accesses per edge instead of 2, no scratch memory for degree-32 checks).
// Algorithmic bytes per iteration and frame: check pass 2*E*4, variable pass (2*E + 2*N)*4 -- about 16x the nibble-row LUT
// decoder, which is the point of the comparison.
#include "../../../include/lut_ldpc_hip.h"
#include "../../../include/lut_ldpc_bp.h"
#include "kernels_common.hpp"      // launch_k: every kernel argument segment <= 128 bytes

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

extern "C" void lutldpc_set_last_error(const char *msg);

namespace {

int bp_fail(int code, const std::string &msg) { lutldpc_set_last_error(msg.c_str()); return code; }
#define BP_TRY(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return bp_fail(LUTLDPC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

constexpr int kMaxTable = 2048;

struct BpParams {
    int32_t d2, d3, qmax, Bpad;
};

__device__ __forceinline__ int bp_logexp(const int32_t *__restrict__ T, const BpParams &P, int x) {
    const int ind = x >> P.d3;
    return ind >= P.d2 ? 0 : T[ind];
}
__device__ __forceinline__ int bp_boxplus(const int32_t *__restrict__ T, const BpParams &P, int a, int c) {
    const int aa = a > 0 ? a : -a, ca = c > 0 ? c : -c, mn = aa > ca ? ca : aa;
    const int t1 = a > 0 ? (c > 0 ? mn : -mn) : (c > 0 ? -mn : mn);
    if (P.d2 == 0) return t1;
    const int apb = a + c, amb = a - c;
    return t1 + bp_logexp(T, P, apb > 0 ? apb : -apb) - bp_logexp(T, P, amb > 0 ? amb : -amb);
}
__device__ __forceinline__ int bp_clip(long long x, int qmax) { return x > qmax ? qmax : (x < -qmax ? -qmax : (int)x); }

// frame-major doubles -> rows of QLLRs (to_qllr), or frame-major ints -> rows
__global__ __launch_bounds__(256) void bp_load_kernel(const double *__restrict__ llr, const int32_t *__restrict__ q, int32_t *__restrict__ rows,
                                                      int B, int N, int Bpad, double scale, int qmax)
{
    const int f = blockIdx.y * 256 + threadIdx.x;
    if (f >= Bpad) return;
    for (int v = blockIdx.x; v < N; v += gridDim.x) {
        int x = 0;
        if (f < B) {
            if (llr) {
                const double t = floor(0.5 + scale * llr[(size_t)f * N + v]);
                x = t >= (double)qmax ? qmax : (t <= -(double)qmax ? -qmax : (int)t);
            } else x = q[(size_t)f * N + v];
        }
        rows[(size_t)v * Bpad + f] = x;
    }
}
// state: 1 = active; start of a decode
__global__ __launch_bounds__(256) void bp_start_kernel(uint8_t *__restrict__ active, int32_t *__restrict__ iters, uint32_t *__restrict__ fail, int B, int Bpad, int max_iters)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= Bpad) return;
    active[f] = f < B ? 1 : 0; iters[f] = -max_iters; fail[f] = 0;
}
// every edge of v carries LLRin[v]; LLRout = LLRin
__global__ __launch_bounds__(256) void bp_init_kernel(const int32_t *__restrict__ in, int32_t *__restrict__ out, int32_t *__restrict__ mvc,
                                                      const int32_t *__restrict__ vn_ptr, int N, int Bpad)
{
    const int f = blockIdx.y * 256 + threadIdx.x;
    for (int v = blockIdx.x; v < N; v += gridDim.x) {
        const int x = in[(size_t)v * Bpad + f];
        out[(size_t)v * Bpad + f] = x;
        for (int e = vn_ptr[v]; e < vn_ptr[v + 1]; e++) mvc[(size_t)e * Bpad + f] = x;
    }
}
// fail[f] |= some check of this block has odd parity over the signs of llr
__global__ __launch_bounds__(256) void bp_syndrome_kernel(const int32_t *__restrict__ llr, const uint8_t *__restrict__ active, uint32_t *__restrict__ fail,
                                                          const int32_t *__restrict__ cn_ptr, const int32_t *__restrict__ cn_vn, int M, int Bpad)
{
    const int f = blockIdx.y * 256 + threadIdx.x;
    if (!active[f]) return;
    uint32_t bad = 0;
    for (int c = blockIdx.x; c < M; c += gridDim.x) {
        uint32_t s = 0;
        for (int k = cn_ptr[c]; k < cn_ptr[c + 1]; k++) s ^= (uint32_t)(llr[(size_t)cn_vn[k] * Bpad + f] < 0);
        bad |= s;
    }
    if (bad) atomicOr(&fail[f], 1u);
}
// after a syndrome pass: frames without a failing check leave with `value`; fail is cleared
__global__ __launch_bounds__(256) void bp_state_kernel(uint8_t *__restrict__ active, int32_t *__restrict__ iters, uint32_t *__restrict__ fail, int Bpad, int value)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= Bpad) return;
    if (active[f] && !fail[f]) { active[f] = 0; iters[f] = value; }
    fail[f] = 0;
}
// check pass (see the header): edges of check c = cn_idx[cn_ptr[c] ..)
__global__ __launch_bounds__(256) void bp_cn_kernel(const int32_t *__restrict__ mvc, int32_t *__restrict__ mcv, const uint8_t *__restrict__ active,
                                                    const int32_t *__restrict__ cn_ptr, const int32_t *__restrict__ cn_idx, const int32_t *__restrict__ table,
                                                    BpParams P, int M)
{
    __shared__ int32_t T[kMaxTable];
    for (int i = threadIdx.x; i < P.d2; i += 256) T[i] = table[i];
    __syncthreads();
    const int f = blockIdx.y * 256 + threadIdx.x;
    if (!active[f]) return;
    const size_t Bp = (size_t)P.Bpad;
    for (int c = blockIdx.x; c < M; c += gridDim.x) {
        const int k0 = cn_ptr[c], n = cn_ptr[c + 1] - k0;
        const int32_t *ix = cn_idx + k0;
        if (n == 2) {
            const int a = mvc[(size_t)ix[0] * Bp + f], b = mvc[(size_t)ix[1] * Bp + f];
            mcv[(size_t)ix[0] * Bp + f] = b; mcv[(size_t)ix[1] * Bp + f] = a;
            continue;
        }
        if (n <= 6) {
            // IT++ 4.3.1 spells the check update out for degrees 3..6 with these association orders (include/lut_ldpc_bp.h)
            int q[6];
            for (int i = 0; i < 6; i++) q[i] = i < n ? mvc[(size_t)ix[i] * Bp + f] : 0;
            int o[6];
            if (n == 3) {
                o[0] = bp_boxplus(T, P, q[1], q[2]); o[1] = bp_boxplus(T, P, q[0], q[2]); o[2] = bp_boxplus(T, P, q[0], q[1]);
            } else if (n == 4) {
                const int m01 = bp_boxplus(T, P, q[0], q[1]), m23 = bp_boxplus(T, P, q[2], q[3]);
                o[0] = bp_boxplus(T, P, q[1], m23); o[1] = bp_boxplus(T, P, q[0], m23); o[2] = bp_boxplus(T, P, m01, q[3]); o[3] = bp_boxplus(T, P, m01, q[2]);
            } else if (n == 5) {
                const int m01 = bp_boxplus(T, P, q[0], q[1]), m02 = bp_boxplus(T, P, m01, q[2]), m34 = bp_boxplus(T, P, q[3], q[4]), m24 = bp_boxplus(T, P, q[2], m34);
                o[0] = bp_boxplus(T, P, q[1], m24); o[1] = bp_boxplus(T, P, q[0], m24); o[2] = bp_boxplus(T, P, m01, m34);
                o[3] = bp_boxplus(T, P, m02, q[4]); o[4] = bp_boxplus(T, P, m02, q[3]);
            } else {
                const int m01 = bp_boxplus(T, P, q[0], q[1]), m23 = bp_boxplus(T, P, q[2], q[3]), m45 = bp_boxplus(T, P, q[4], q[5]);
                const int m03 = bp_boxplus(T, P, m01, m23), m25 = bp_boxplus(T, P, m23, m45), m0145 = bp_boxplus(T, P, m01, m45);
                o[0] = bp_boxplus(T, P, q[1], m25); o[1] = bp_boxplus(T, P, q[0], m25); o[2] = bp_boxplus(T, P, m0145, q[3]);
                o[3] = bp_boxplus(T, P, m0145, q[2]); o[4] = bp_boxplus(T, P, m03, q[5]); o[5] = bp_boxplus(T, P, m03, q[4]);
            }
            for (int i = 0; i < 6; i++) if (i < n) mcv[(size_t)ix[i] * Bp + f] = o[i];
            continue;
        }
        // up: mcv[e_i] <- ml[i-1] = boxplus of m[0..i-1] (left-associated), i = 1..n-1
        int acc = mvc[(size_t)ix[0] * Bp + f];
        for (int i = 1; i < n; i++) {
            mcv[(size_t)ix[i] * Bp + f] = acc;
            if (i < n - 1) acc = bp_boxplus(T, P, acc, mvc[(size_t)ix[i] * Bp + f]);
        }
        // down: running right sum mr, out[i] = boxplus(ml[i-1], mr[n-2-i]); out[n-1] = ml[n-2] is already in place
        acc = mvc[(size_t)ix[n - 1] * Bp + f];
        for (int i = n - 2; i >= 1; i--) {
            const int left = mcv[(size_t)ix[i] * Bp + f];
            mcv[(size_t)ix[i] * Bp + f] = bp_boxplus(T, P, left, acc);
            acc = bp_boxplus(T, P, acc, mvc[(size_t)ix[i] * Bp + f]);
        }
        mcv[(size_t)ix[0] * Bp + f] = acc;
    }
}
// variable pass: s = LLRin + sum(mcv); LLRout = clip(s); mvc = clip(s - mcv)
__global__ __launch_bounds__(256) void bp_vn_kernel(const int32_t *__restrict__ in, int32_t *__restrict__ out, int32_t *__restrict__ mvc, const int32_t *__restrict__ mcv,
                                                    const uint8_t *__restrict__ active, const int32_t *__restrict__ vn_ptr, int N, int Bpad, int qmax)
{
    const int f = blockIdx.y * 256 + threadIdx.x;
    if (!active[f]) return;
    const size_t Bp = (size_t)Bpad;
    for (int v = blockIdx.x; v < N; v += gridDim.x) {
        const int e0 = vn_ptr[v], e1 = vn_ptr[v + 1];
        long long s = in[(size_t)v * Bp + f];
        for (int e = e0; e < e1; e++) s += mcv[(size_t)e * Bp + f];
        out[(size_t)v * Bp + f] = bp_clip(s, qmax);
        for (int e = e0; e < e1; e++) mvc[(size_t)e * Bp + f] = bp_clip(s - mcv[(size_t)e * Bp + f], qmax);
    }
}
// rows -> frame-major bits (LLRout < 0) and, optionally, LLRout itself
__global__ __launch_bounds__(256) void bp_store_kernel(const int32_t *__restrict__ rows, uint8_t *__restrict__ bits, int32_t *__restrict__ q, int B, int N, int Bpad)
{
    const int f = blockIdx.y * 256 + threadIdx.x;
    if (f >= B) return;
    for (int v = blockIdx.x; v < N; v += gridDim.x) {
        const int x = rows[(size_t)v * Bpad + f];
        bits[(size_t)f * N + v] = x < 0 ? 1 : 0;
        if (q) q[(size_t)f * N + v] = x;
    }
}

template <class T>
struct Buf {
    T *p = nullptr; size_t n = 0;
    hipError_t alloc(size_t c) { if (c <= n) return hipSuccess; release(); hipError_t e = hipMalloc((void **)&p, c * sizeof(T)); if (e == hipSuccess) n = c; else p = nullptr; return e; }
    hipError_t upload(const std::vector<T> &h) { hipError_t e = alloc(h.size() ? h.size() : 1); if (e != hipSuccess || h.empty()) return e; return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

struct lutldpc_bp_decoder {
    int nvar = 0, nchk = 0, E = 0, d1 = 12, d2 = 300, d3 = 7, d4 = 28, qmax = 0;
    int max_iters = 50, psc = 1, pisc = 0, device = -1;
    std::vector<int32_t> vn_ptr, cn_ptr, cn_idx, cn_vn, table;
    hipStream_t stream = nullptr;
    Buf<int32_t> d_vn_ptr, d_cn_ptr, d_cn_idx, d_cn_vn, d_table, d_mvc, d_mcv, d_in, d_out, d_iters, d_q_in, d_q_out;
    Buf<double> d_llr;
    Buf<uint8_t> d_active, d_bits;
    Buf<uint32_t> d_fail;
};

namespace {

int to_qllr_host(const lutldpc_bp_decoder *d, double l) {
    const double v = std::floor(0.5 + std::ldexp(1.0, d->d1) * l);
    if (v >= (double)d->qmax) return d->qmax;
    if (v <= -(double)d->qmax) return -d->qmax;
    return (int)v;
}

int decode_rows(lutldpc_bp_decoder *d, int B, int Bpad) {
    const unsigned gy = (unsigned)(Bpad / 256);
    const unsigned gxN = (unsigned)std::min(d->nvar, 4096), gxM = (unsigned)std::min(d->nchk, 4096);
    BpParams P{d->d2, d->d3, d->qmax, Bpad};
    lutldpc::launch_k(bp_start_kernel, dim3(gy), dim3(256), 0, d->stream, d->d_active.p, d->d_iters.p, d->d_fail.p, B, Bpad, d->max_iters);
    lutldpc::launch_k(bp_init_kernel, dim3(gxN, gy), dim3(256), 0, d->stream, d->d_in.p, d->d_out.p, d->d_mvc.p, d->d_vn_ptr.p, d->nvar, Bpad);
    if (d->pisc) {
        lutldpc::launch_k(bp_syndrome_kernel, dim3(gxM, gy), dim3(256), 0, d->stream, d->d_in.p, d->d_active.p, d->d_fail.p, d->d_cn_ptr.p, d->d_cn_vn.p, d->nchk, Bpad);
        lutldpc::launch_k(bp_state_kernel, dim3(gy), dim3(256), 0, d->stream, d->d_active.p, d->d_iters.p, d->d_fail.p, Bpad, 0);
    }
    for (int it = 1; it <= d->max_iters; it++) {
        lutldpc::launch_k(bp_cn_kernel, dim3(gxM, gy), dim3(256), 0, d->stream, d->d_mvc.p, d->d_mcv.p, d->d_active.p, d->d_cn_ptr.p, d->d_cn_idx.p, d->d_table.p, P, d->nchk);
        lutldpc::launch_k(bp_vn_kernel, dim3(gxN, gy), dim3(256), 0, d->stream, d->d_in.p, d->d_out.p, d->d_mvc.p, d->d_mcv.p, d->d_active.p, d->d_vn_ptr.p, d->nvar, Bpad, d->qmax);
        if (d->psc) {
            lutldpc::launch_k(bp_syndrome_kernel, dim3(gxM, gy), dim3(256), 0, d->stream, d->d_out.p, d->d_active.p, d->d_fail.p, d->d_cn_ptr.p, d->d_cn_vn.p, d->nchk, Bpad);
            lutldpc::launch_k(bp_state_kernel, dim3(gy), dim3(256), 0, d->stream, d->d_active.p, d->d_iters.p, d->d_fail.p, Bpad, it);
        }
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bp_fail(LUTLDPC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return LUTLDPC_OK;
}

int decode_common(lutldpc_bp_decoder *d, const double *llr, const int32_t *q, int B, uint8_t *out_bits, int32_t *out_iters, int32_t *out_qllr) {
    if (!d || (!llr && !q) || !out_bits || !out_iters) return bp_fail(LUTLDPC_ERR_ARG, "NULL argument");
    if (d->device < 0) return bp_fail(LUTLDPC_ERR_STATE, "BP decoder was created without a device (host-only handle)");
    if (B <= 0) return bp_fail(LUTLDPC_ERR_ARG, "B must be positive");
    BP_TRY(hipSetDevice(d->device));
    const int Bpad = (B + 255) / 256 * 256;
    const size_t n = (size_t)B * d->nvar;
    BP_TRY(d->d_mvc.alloc((size_t)d->E * Bpad)); BP_TRY(d->d_mcv.alloc((size_t)d->E * Bpad));
    BP_TRY(d->d_in.alloc((size_t)d->nvar * Bpad)); BP_TRY(d->d_out.alloc((size_t)d->nvar * Bpad));
    BP_TRY(d->d_iters.alloc((size_t)Bpad)); BP_TRY(d->d_active.alloc((size_t)Bpad)); BP_TRY(d->d_fail.alloc((size_t)Bpad));
    BP_TRY(d->d_bits.alloc(n));
    if (out_qllr) BP_TRY(d->d_q_out.alloc(n));
    const unsigned gy = (unsigned)(Bpad / 256), gxN = (unsigned)std::min(d->nvar, 4096);
    if (llr) {
        BP_TRY(d->d_llr.alloc(n));
        BP_TRY(hipMemcpyAsync(d->d_llr.p, llr, n * sizeof(double), hipMemcpyHostToDevice, d->stream));
        lutldpc::launch_k(bp_load_kernel, dim3(gxN, gy), dim3(256), 0, d->stream, d->d_llr.p, (const int32_t *)nullptr, d->d_in.p, B, d->nvar, Bpad, std::ldexp(1.0, d->d1), d->qmax);
    } else {
        BP_TRY(d->d_q_in.alloc(n));
        BP_TRY(hipMemcpyAsync(d->d_q_in.p, q, n * sizeof(int32_t), hipMemcpyHostToDevice, d->stream));
        lutldpc::launch_k(bp_load_kernel, dim3(gxN, gy), dim3(256), 0, d->stream, (const double *)nullptr, d->d_q_in.p, d->d_in.p, B, d->nvar, Bpad, 1.0, d->qmax);
    }
    if (int rc = decode_rows(d, B, Bpad)) return rc;
    lutldpc::launch_k(bp_store_kernel, dim3(gxN, gy), dim3(256), 0, d->stream, d->d_out.p, d->d_bits.p, out_qllr ? d->d_q_out.p : nullptr, B, d->nvar, Bpad);
    BP_TRY(hipMemcpyAsync(out_bits, d->d_bits.p, n, hipMemcpyDeviceToHost, d->stream));
    BP_TRY(hipMemcpyAsync(out_iters, d->d_iters.p, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, d->stream));
    if (out_qllr) BP_TRY(hipMemcpyAsync(out_qllr, d->d_q_out.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, d->stream));
    BP_TRY(hipStreamSynchronize(d->stream));
    return LUTLDPC_OK;
}

}  // namespace

extern "C" {

int lutldpc_bp_create(int nvar, int nchk, const int32_t *dv, const int32_t *dc, const int32_t *cn_msg_idx, int d1, int d2, int d3, int d4, int device,
                      lutldpc_bp_decoder **out) {
    if (!out) return bp_fail(LUTLDPC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (nvar <= 0 || nchk <= 0 || !dv || !dc || !cn_msg_idx) return bp_fail(LUTLDPC_ERR_ARG, "missing or non-positive argument");
    if (d1 < 0 || d1 > 24 || d2 < 0 || d2 > kMaxTable || d3 < 0 || d3 > 24 || d4 < 8 || d4 > 29)
        return bp_fail(LUTLDPC_ERR_ARG, "LLR_calc_unit parameters outside the supported range (d1 <= 24, d2 <= 2048, d3 <= 24, 8 <= d4 <= 29)");
    std::unique_ptr<lutldpc_bp_decoder> d(new lutldpc_bp_decoder);
    d->nvar = nvar; d->nchk = nchk; d->d1 = d1; d->d2 = d2; d->d3 = d3; d->d4 = d4; d->qmax = (int)((1ll << (d4 - 1)) - 1);
    d->vn_ptr.assign((size_t)nvar + 1, 0); d->cn_ptr.assign((size_t)nchk + 1, 0);
    long long ev = 0, ec = 0;
    for (int v = 0; v < nvar; v++) { if (dv[v] < 1 || dv[v] > 255) return bp_fail(LUTLDPC_ERR_ARG, "variable degree outside [1,255]"); ev += dv[v]; d->vn_ptr[(size_t)v + 1] = (int32_t)ev; }
    // itpp::LDPC_Code::bp_decode stops with it_error on a check of degree 0 or 1
    for (int c = 0; c < nchk; c++) { if (dc[c] < 2 || dc[c] > 255) return bp_fail(LUTLDPC_ERR_ARG, "check degree outside [2,255]"); ec += dc[c]; d->cn_ptr[(size_t)c + 1] = (int32_t)ec; }
    if (ev != ec || ev > (1ll << 28)) return bp_fail(LUTLDPC_ERR_ARG, "sum(dv) != sum(dc)");
    d->E = (int)ev;
    d->cn_idx.assign(cn_msg_idx, cn_msg_idx + d->E);
    std::vector<int32_t> edge_vn((size_t)d->E);
    for (int v = 0; v < nvar; v++) for (int e = d->vn_ptr[(size_t)v]; e < d->vn_ptr[(size_t)v + 1]; e++) edge_vn[(size_t)e] = v;
    std::vector<uint8_t> seen((size_t)d->E, 0);
    d->cn_vn.resize((size_t)d->E);
    for (int k = 0; k < d->E; k++) {
        const int e = cn_msg_idx[k];
        if (e < 0 || e >= d->E || seen[(size_t)e]) return bp_fail(LUTLDPC_ERR_ARG, "cn_msg_idx is not a permutation of the edges");
        seen[(size_t)e] = 1; d->cn_vn[(size_t)k] = edge_vn[(size_t)e];
    }
    d->table.resize((size_t)d2);
    for (int i = 0; i < d2; i++) d->table[(size_t)i] = to_qllr_host(d.get(), std::log(1.0 + std::exp(-std::ldexp(1.0, d3 - d1) * i)));
    d->device = device;
    if (device >= 0) {
        BP_TRY(hipSetDevice(device));
        BP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
        BP_TRY(d->d_vn_ptr.upload(d->vn_ptr)); BP_TRY(d->d_cn_ptr.upload(d->cn_ptr)); BP_TRY(d->d_cn_idx.upload(d->cn_idx));
        BP_TRY(d->d_cn_vn.upload(d->cn_vn)); BP_TRY(d->d_table.upload(d->table));
    }
    *out = d.release();
    return LUTLDPC_OK;
}

int lutldpc_bp_destroy(lutldpc_bp_decoder *d) {
    if (!d) return LUTLDPC_OK;
    if (d->device >= 0) {
        (void)hipSetDevice(d->device);
        if (d->stream) (void)hipStreamSynchronize(d->stream);
        d->d_vn_ptr.release(); d->d_cn_ptr.release(); d->d_cn_idx.release(); d->d_cn_vn.release(); d->d_table.release(); d->d_mvc.release(); d->d_mcv.release();
        d->d_in.release(); d->d_out.release(); d->d_iters.release(); d->d_q_in.release(); d->d_q_out.release(); d->d_llr.release(); d->d_active.release();
        d->d_bits.release(); d->d_fail.release();
        if (d->stream) (void)hipStreamDestroy(d->stream);
    }
    delete d;
    return LUTLDPC_OK;
}

int lutldpc_bp_set_exit_conditions(lutldpc_bp_decoder *d, int max_iters, int psc, int pisc) {
    if (!d || max_iters < 1) return bp_fail(LUTLDPC_ERR_ARG, "NULL decoder or max_iters < 1");
    d->max_iters = max_iters; d->psc = psc ? 1 : 0; d->pisc = pisc ? 1 : 0;
    return LUTLDPC_OK;
}

int lutldpc_bp_logexp_table(lutldpc_bp_decoder *d, int32_t *out, int cap) {
    if (!d) return bp_fail(LUTLDPC_ERR_ARG, "NULL decoder");
    if (out && cap >= d->d2 && d->d2 > 0) std::memcpy(out, d->table.data(), sizeof(int32_t) * (size_t)d->d2);
    return d->d2;
}

int lutldpc_bp_decode_llr_batch(lutldpc_bp_decoder *d, const double *llr, int B, uint8_t *out_bits, int32_t *out_iters, int32_t *out_qllr) {
    if (!llr) return bp_fail(LUTLDPC_ERR_ARG, "NULL argument");
    return decode_common(d, llr, nullptr, B, out_bits, out_iters, out_qllr);
}
int lutldpc_bp_decode_qllr_batch(lutldpc_bp_decoder *d, const int32_t *qllr, int B, uint8_t *out_bits, int32_t *out_iters, int32_t *out_qllr) {
    if (!qllr) return bp_fail(LUTLDPC_ERR_ARG, "NULL argument");
    return decode_common(d, nullptr, qllr, B, out_bits, out_iters, out_qllr);
}

}  // extern "C"
