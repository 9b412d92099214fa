// kernels_frontend.hpp -- Monte-Carlo front end on the device: the per-frame work of
// LDPC_BER_Sim::sim_snr_point (src/LDPC_BER_Sim.cpp:260-286) other than the decode itself.
//
// The reference draws N Gaussian samples per frame, forms LLR = 4x/N0 and quantises them
// (BPSK + AWGN_Channel + demodulate_soft_bits + quant_nonlin).  All the decoder ever sees of a
// received value x is (channel label, initial message label, slicer sign), i.e. which cell of the
// partition of the real line by the thresholds {qb_Cha*N0/4} u {qb_Msg*N0/4} u {0} it fell
// into.  The sampler therefore draws the CELL directly: one 64-bit uniform per code bit is
// compared against the integer cumulative cell probabilities (computed once per SNR point on the
// host from the Gaussian cdf).  This is an exact sampler of the same joint distribution (up to
// the 2^-64 resolution of the thresholds), needs no transcendental on the device and -- because
// everything after the thresholds is integer arithmetic -- gives bit-identical label streams on
// the GPU and in the CPU oracle.  (The IT++ random stream itself cannot be reproduced: SURVEY F5.)
//
// Random numbers: Philox4x32-10 (Salmon et al., SC'11), key = seed, counter =
// (frame index lo, hi, code-bit pair index, stream id); the four output words give the two
// 64-bit uniforms of code bits 2p and 2p+1.  Frames are therefore addressable: any sharding of
// the frame range over GPUs produces the same frames.
#pragma once
#include "kernels_common.hpp"

namespace lutldpc {

constexpr int kMaxCells = 72;      // 2*(Nq-1)+1 thresholds for Nq <= 32 ... generous

struct ChannelCells {
    int32_t n_cells;
    int32_t pad;
    uint64_t thr[kMaxCells];       // thr[j] = floor(2^64 * P(cell <= j | bit 0)), j < n_cells-1
    uint8_t cha[kMaxCells];        // labels of each cell when bit 0 was sent
    uint8_t msg[kMaxCells];
    uint8_t neg[kMaxCells];        // 1: x < 0 (slicer decides bit 1)
    uint8_t cha_m[kMaxCells];      // labels of the mirrored cell (bit 1 was sent: x -> -x)
    uint8_t msg_m[kMaxCells];
};

struct Philox {
    __host__ __device__ static inline void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    __host__ __device__ static inline void gen(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
        for (int r = 0; r < 10; r++) {
            round(c, k0, k1);
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
    }
};

// cell index = number of thresholds <= u.  The thresholds ascend (checked on the host), so this is an upper-bound
// binary search: 7 steps over the LDS copy of the table instead of up to 71 64-bit compares per sample.  (Used once per
// bucket in the prologue of the sampler; the samples themselves start from their bucket's first cell, below.)
__device__ __forceinline__ int cell_of(const uint64_t *thr_lds, int n_thr, uint64_t u) {
    int lo = 0, n = n_thr;                    // invariant: thr[0..lo) <= u, answer in [lo, lo + n]
#pragma unroll
    for (int step = 0; step < 7; step++) {    // 2^7 > kMaxCells
        const int half = n >> 1;
        const bool right = n > 0 && u >= thr_lds[lo + half];
        lo = right ? lo + half + 1 : lo;
        n = right ? n - half - 1 : half;
    }
    return lo;
}
// The same cell index, found from the sample's BUCKET: the top kBucketBits bits of u select one of 1024 equal slices of the
// unit interval, first_lds[bucket] = number of thresholds <= the slice's lower end, and the few thresholds inside the slice are
// walked one by one (most slices hold none or one; the dense ones sit in the far tails, hit once in ~500 samples).  The
// walk is a wave loop: it ends when no lane advances.  Same result as cell_of for every u.
constexpr int kBucketBits = 10;
__device__ __forceinline__ int cell_from_bucket(const uint64_t *thr_lds, const uint8_t *first_lds, int n_thr, uint64_t u) {
    int cell = first_lds[(uint32_t)(u >> (64 - kBucketBits))];
    for (;;) {
        const bool adv = cell < n_thr && u >= thr_lds[cell < n_thr ? cell : 0];
        if (!__any(adv)) break;
        cell += adv ? 1 : 0;
    }
    return cell;
}

// Writes cha_t / msg0_t rows (row layout) for B frames starting at global frame index frame0 and
// adds the slicer errors of each frame to stats[f][3].  One thread = 4*PACK frames x a run of
// code-bit pairs.  codewords (frame-major [B][N], may be null = all-zero codeword): the sent bits.
template <int PACK>
__global__ __launch_bounds__(256) void sample_labels_kernel(const ChannelCells *__restrict__ Cp, uint32_t seed_lo, uint32_t seed_hi, uint32_t stream, uint64_t frame0,
                                                            int B, int N, const uint8_t *__restrict__ codewords, uint8_t *__restrict__ cha_t,
                                                            uint8_t *__restrict__ msg_t, int32_t *__restrict__ stats, int pairs_per_thread)
{
    constexpr int F = 4 * PACK;                                     // frames per lane
    __shared__ uint64_t thr_lds[kMaxCells];
    __shared__ uint32_t attr_lds[kMaxCells];                        // per cell: cha | neg << 7 | msg << 8 | cha_m << 16 | msg_m << 24 (labels < 128)
    __shared__ uint8_t first_lds[1 << kBucketBits];
    const ChannelCells &C = *Cp;                                    // (device memory: the argument segment stays small)
    const int n_thr = C.n_cells - 1;
    if (threadIdx.x < kMaxCells) {
        const int t = threadIdx.x < C.n_cells ? threadIdx.x : 0;
        thr_lds[threadIdx.x] = C.thr[threadIdx.x < n_thr ? threadIdx.x : 0];
        attr_lds[threadIdx.x] = (uint32_t)C.cha[t] | ((uint32_t)C.neg[t] << 7) | ((uint32_t)C.msg[t] << 8) | ((uint32_t)C.cha_m[t] << 16) | ((uint32_t)C.msg_m[t] << 24);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < (1 << kBucketBits); t += 256)
        first_lds[t] = (uint8_t)cell_of(thr_lds, n_thr, (uint64_t)t << (64 - kBucketBits));
    __syncthreads();
    const int lane = threadIdx.x & 63, g = blockIdx.y;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int npairs = (N + 1) / 2;
    int p0 = w * pairs_per_thread, p1 = p0 + pairs_per_thread;
    if (p1 > npairs) p1 = npairs;
    int unc[F];
#pragma unroll
    for (int j = 0; j < F; j++) unc[j] = 0;
    for (int p = p0; p < p1; p++) {
        uint32_t ca[2][PACK], ms[2][PACK];
#pragma unroll
        for (int h = 0; h < PACK; h++) { ca[0][h] = ca[1][h] = 0; ms[0][h] = ms[1][h] = 0; }
#pragma unroll
        for (int j = 0; j < F; j++) {
            const int fl = (g * kWave + lane) * F + j;              // frame within the batch
            if (fl >= B) continue;                                  // pad frames keep label 0
            const uint64_t f = frame0 + (uint64_t)fl;
            uint32_t c[4] = {(uint32_t)f, (uint32_t)(f >> 32), (uint32_t)p, stream};
            Philox::gen(c, seed_lo, seed_hi);
            const uint64_t u[2] = {((uint64_t)c[1] << 32) | c[0], ((uint64_t)c[3] << 32) | c[2]};
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int v = 2 * p + hh;
                if (v >= N) continue;
                const int cell = cell_from_bucket(thr_lds, first_lds, n_thr, u[hh]);
                const int bit = codewords ? codewords[(size_t)fl * N + v] : 0;
                const uint32_t at = attr_lds[cell], sel = bit ? at >> 16 : at;
                const uint32_t a = sel & 0x7Fu, m = (sel >> 8) & 0xFFu;
                // slicer decision: neg for bit 0, its mirror for bit 1 -- wrong exactly when the cell lies on the negative side
                unc[j] += (int)((at >> 7) & 1u);
                ca[hh][j / 4] |= a << (8 * (j & 3));
                ms[hh][j / 4] |= m << (8 * (j & 3));
            }
        }
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int v = 2 * p + hh;
            if (v >= N) continue;
            *reinterpret_cast<uint32_t *>(cha_t + ((size_t)g * N + v) * kRowBytes + lane * 4) = pack_halves<PACK>(ca[hh]);
            *reinterpret_cast<uint32_t *>(msg_t + ((size_t)g * N + v) * kRowBytes + lane * 4) = pack_halves<PACK>(ms[hh]);
        }
    }
#pragma unroll
    for (int j = 0; j < F; j++) {
        const int fl = (g * kWave + lane) * F + j;
        if (fl < B && unc[j]) atomicAdd(&stats[(size_t)fl * 4 + 3], unc[j]);
    }
}

// stats[f] = {iteration code, frame error, data bit errors, uncoded bit errors}: compares the decided
// bits of the first K positions with the sent ones (BERC / BLERC of src/LDPC_BER_Sim.cpp:284-286)
template <int PACK>
__global__ __launch_bounds__(256) void count_errors_kernel(const uint8_t *__restrict__ hard, const uint8_t *__restrict__ codewords, int B, int N, int K,
                                                           const int32_t *__restrict__ iters, int32_t *__restrict__ stats, int rows_per_wave)
{
    constexpr int F = 4 * PACK;
    const int lane = threadIdx.x & 63, g = blockIdx.y;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    int v0 = w * rows_per_wave, v1 = v0 + rows_per_wave;
    if (v1 > K) v1 = K;
    int err[F];
#pragma unroll
    for (int j = 0; j < F; j++) err[j] = 0;
    for (int v = v0; v < v1; v++) {
        const uint32_t x = *reinterpret_cast<const uint32_t *>(hard + ((size_t)g * N + v) * kRowBytes + lane * 4);
#pragma unroll
        for (int j = 0; j < F; j++) {
            const int fl = (g * kWave + lane) * F + j;
            const int sent = (codewords && fl < B) ? codewords[(size_t)fl * N + v] : 0;
            const uint32_t hw = unpack_half<PACK>(x, j / 4);
            err[j] += (int)((hw >> (8 * (j & 3))) & 1u) != sent;
        }
    }
#pragma unroll
    for (int j = 0; j < F; j++) {
        const int fl = (g * kWave + lane) * F + j;
        if (fl < B && err[j]) { atomicAdd(&stats[(size_t)fl * 4 + 2], err[j]); atomicOr(&stats[(size_t)fl * 4 + 1], 1); }
        if (fl < B && v0 == 0) stats[(size_t)fl * 4 + 0] = iters[fl];
    }
}

}  // namespace lutldpc
