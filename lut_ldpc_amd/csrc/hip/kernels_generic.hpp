// kernels_generic.hpp -- generic (any tree, any degree) node-update kernels and the layout
// kernels.  These are the correctness baseline every specialised kernel is tested against.
#pragma once
#include "kernels_common.hpp"
#include "lut_program.hpp"

namespace lutldpc {

// ------------------------------------------------------------------------------------------
// Node-program interpreter.  KIND: TT_VAR (variable-node pass, src/LDPC_Code_LUT.cpp:404-414),
// TT_CHK (CHKTREE check pass, :416-426), TT_DEC (decision pass, :428-434,340-344).
// One wave per block; the program's value slots live in LDS as [slot][lane] dwords (four
// frames per dword), so a slot access is a conflict-free ds_read/write_b32.  Tables of the
// class are staged into LDS when they fit (LDS_TAB), else read through L1/L2.
// ------------------------------------------------------------------------------------------
template <int KIND, bool LDS_TAB>
__global__ __launch_bounds__(64) void tree_pass_kernel(
    PassParams P, uint8_t *__restrict__ msgs, const uint8_t *__restrict__ cha, uint8_t *__restrict__ hard,
    const uint32_t *__restrict__ state_w, uint32_t *__restrict__ vfail_w, const Op *__restrict__ ops,
    const uint8_t *__restrict__ tables, const int32_t *__restrict__ node_list,
    const int32_t *__restrict__ node_ptr,   // VAR/DEC: first edge of each VN ; CHK: offset of each CN in cn_idx
    const int32_t *__restrict__ cn_idx)
{
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
    const int g = blockIdx.x / P.blocks_per_group;
    const int b = blockIdx.x - g * P.blocks_per_group;
    const int si = find_seg(P, b);
    const PassSeg S = P.seg[si];

    const uint32_t st = state_w[g * kWave + lane];
    const uint32_t amask = swar_zero_mask(st);              // 0xFF for frames still decoding
    if (wave_all_zero(amask)) return;

    uint32_t *slot = lds;                                    // [n_slots][64]
    const uint8_t *tab;
    if (LDS_TAB) {
        uint32_t *lt = lds + P.slots_lds * kWave;
        const uint32_t *gt = reinterpret_cast<const uint32_t *>(tables + S.tab_off);
        for (int i = lane; i < S.tab_bytes / 4; i += kWave) lt[i] = gt[i];
        __syncthreads();
        tab = reinterpret_cast<const uint8_t *>(lt);
    } else {
        tab = tables + S.tab_off;
    }

    const size_t gE = (size_t)g * (size_t)P.E, gN = (size_t)g * (size_t)P.N;
    const int first = (b - S.block_begin) * P.nodes_per_block;
    int last = first + P.nodes_per_block;
    if (last > S.n_nodes) last = S.n_nodes;
    const Op *prog = ops + S.op_off;
    uint32_t fail = 0;

    for (int i = first; i < last; i++) {
        const int node = node_list[S.node_off + i];
        const int p0 = node_ptr[node];
        // ---- gather inputs into slots 0..n_in-1
        if (KIND == TT_CHK) {
            uint32_t par = 0;
            for (int k = 0; k < S.deg; k++) {
                const int e = cn_idx[p0 + k];
                const uint32_t x = *reinterpret_cast<const uint32_t *>(msgs + (gE + (size_t)e) * kTileFrames + lane * 4);
                slot[k * kWave + lane] = x;
                par ^= swar_lt(x, (uint32_t)P.nz);
            }
            if (P.check) fail |= par;
        } else {
            for (int k = 0; k < S.deg; k++)
                slot[k * kWave + lane] = *reinterpret_cast<const uint32_t *>(msgs + (gE + (size_t)(p0 + k)) * kTileFrames + lane * 4);
            slot[S.deg * kWave + lane] = *reinterpret_cast<const uint32_t *>(cha + (gN + (size_t)node) * kTileFrames + lane * 4);
        }
        // ---- run the program
        uint32_t neg_ref = 0, have_ref = 0;
        for (int o = 0; o < S.n_ops; o++) {
            const Op &op = prog[o];
            uint32_t lab[4] = {0, 0, 0, 0}, par[4] = {0, 0, 0, 0};
            for (int c = 0; c < op.nchild; c++) {
                const uint32_t x = slot[op.child[c] * kWave + lane];
                const uint32_t m = op.mult[c];
                if (op.kind == 1) {
                    const uint32_t h = op.childK[c] >> 1;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t v = (x >> (8 * j)) & 0xFFu;
                        const bool ng = v < h;
                        par[j] ^= ng ? 1u : 0u;
                        lab[j] += m * (ng ? (h - 1 - v) : (v - h));
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) lab[j] += m * ((x >> (8 * j)) & 0xFFu);
                }
            }
            uint32_t r = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t idx = lab[j];
                if (op.kind == 1 && !par[j]) idx += op.half_len;
                if (idx >= op.tab_len) idx = op.tab_len - 1;          // memory safety on corrupt labels
                r |= (uint32_t)tab[op.tab_off + idx] << (8 * j);
            }
            slot[op.dst * kWave + lane] = r;
            if (op.out_idx >= 0) {
                if (KIND == TT_DEC) {
                    // bit = (label < 1), src/LDPC_Code_LUT.cpp:342
                    const uint32_t bit = swar_lt(r, 1u);
                    uint8_t *hp = hard + (gN + (size_t)node) * kTileFrames + lane * 4;
                    if (amask == 0xFFFFFFFFu) *reinterpret_cast<uint32_t *>(hp) = bit;
                    else if (amask) *reinterpret_cast<uint32_t *>(hp) = bfi(amask, bit, *reinterpret_cast<uint32_t *>(hp));
                } else {
                    // in-place update; frames that already terminated keep their old message
                    size_t row;
                    if (KIND == TT_CHK) row = gE + (size_t)cn_idx[p0 + op.out_idx];
                    else row = gE + (size_t)(p0 + op.out_idx);
                    const uint32_t old = slot[op.out_idx * kWave + lane];
                    *reinterpret_cast<uint32_t *>(msgs + row * kTileFrames + lane * 4) = bfi(amask, r, old);
                    if (KIND == TT_VAR && (P.check || P.write_hard)) {
                        const uint32_t ng = swar_lt(r, (uint32_t)P.nz);
                        if (!have_ref) { neg_ref = ng; have_ref = 1; }
                        else fail |= ng ^ neg_ref;
                    }
                }
            }
        }
        if (KIND == TT_VAR && P.write_hard) {
            uint8_t *hp = hard + (gN + (size_t)node) * kTileFrames + lane * 4;
            if (amask == 0xFFFFFFFFu) *reinterpret_cast<uint32_t *>(hp) = neg_ref;
            else if (amask) *reinterpret_cast<uint32_t *>(hp) = bfi(amask, neg_ref, *reinterpret_cast<uint32_t *>(hp));
        }
    }
    if (KIND != TT_DEC && P.check) {
        fail &= amask;
        if (fail) atomicOr(&vfail_w[g * kWave + lane], fail);
    }
}

// ------------------------------------------------------------------------------------------
// Min-sum check pass, any degree (src/LDPC_Code_LUT.cpp:355-402).  Two sweeps over the
// check's rows: the first accumulates min1/min2/sign product, the second re-reads each row
// (L1/L2 hit) and writes the extrinsic message.  Byte-serial arithmetic; the specialised
// kernel in kernels_fast.hpp keeps the rows in registers and works on packed bytes.
// out magnitude for edge k is min over the others = (mag_k == min1 ? min2 : min1): when the
// minimum occurs twice min2 == min1, so the reference's min_idx bookkeeping is not needed.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void cn_minsum_generic_kernel(
    PassParams P, uint8_t *__restrict__ msgs, const uint32_t *__restrict__ state_w, uint32_t *__restrict__ vfail_w,
    const int32_t *__restrict__ node_list, const int32_t *__restrict__ cn_ptr, const int32_t *__restrict__ cn_idx)
{
    const int lane = threadIdx.x;
    const int g = blockIdx.x / P.blocks_per_group;
    const int b = blockIdx.x - g * P.blocks_per_group;
    const PassSeg S = P.seg[find_seg(P, b)];
    const uint32_t amask = swar_zero_mask(state_w[g * kWave + lane]);
    if (wave_all_zero(amask)) return;
    const size_t gE = (size_t)g * (size_t)P.E;
    const int first = (b - S.block_begin) * P.nodes_per_block;
    int last = first + P.nodes_per_block;
    if (last > S.n_nodes) last = S.n_nodes;
    const int nz = P.nz;
    uint32_t fail = 0;
    for (int i = first; i < last; i++) {
        const int c = node_list[S.node_off + i];
        const int p0 = cn_ptr[c];
        int min1[4], min2[4];
        uint32_t sp = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { min1[j] = nz; min2[j] = nz; }
        for (int k = 0; k < S.deg; k++) {
            const uint32_t x = *reinterpret_cast<const uint32_t *>(msgs + (gE + (size_t)cn_idx[p0 + k]) * kTileFrames + lane * 4);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int v = (int)((x >> (8 * j)) & 0xFFu);
                int t;
                if (v < nz) { sp ^= 1u << (8 * j); t = nz - 1 - v; } else t = v - nz;
                if (t < min1[j]) { min2[j] = min1[j]; min1[j] = t; }
                else if (t < min2[j]) min2[j] = t;
            }
        }
        if (P.check) fail |= sp;
        for (int k = 0; k < S.deg; k++) {
            uint32_t *row = reinterpret_cast<uint32_t *>(msgs + (gE + (size_t)cn_idx[p0 + k]) * kTileFrames + lane * 4);
            const uint32_t x = *row;
            uint32_t r = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int v = (int)((x >> (8 * j)) & 0xFFu);
                const bool ng = v < nz;
                const int t = ng ? nz - 1 - v : v - nz;
                const int m = (t == min1[j]) ? min2[j] : min1[j];
                const uint32_t s = ((sp >> (8 * j)) & 1u) ^ (ng ? 1u : 0u);
                const int o = s ? nz - 1 - m : nz + m;
                r |= (uint32_t)(o & 0xFF) << (8 * j);
            }
            *row = bfi(amask, r, x);
        }
    }
    if (P.check) {
        fail &= amask;
        if (fail) atomicOr(&vfail_w[g * kWave + lane], fail);
    }
}

// ------------------------------------------------------------------------------------------
// Layout kernels
// ------------------------------------------------------------------------------------------
// frame-major [B][N] -> tiles [G][N][256]; labels are clamped to < limit, pad frames get 0.
__global__ __launch_bounds__(256) void transpose_in_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                           int B, int N, int limit)
{
    __shared__ uint8_t tile[kTileFrames][64 + 4];
    const int g = blockIdx.y, n0 = blockIdx.x * 64, t = threadIdx.x;
    const int nn = t & 63, fq = t >> 6;
    for (int f = fq; f < kTileFrames; f += 4) {
        const int fr = g * kTileFrames + f, n = n0 + nn;
        uint8_t v = 0;
        if (fr < B && n < N) { v = src[(size_t)fr * N + n]; if (v >= limit) v = (uint8_t)(limit - 1); }
        tile[f][nn] = v;
    }
    __syncthreads();
    const int lane = t & 63;
    for (int r = t >> 6; r < 64; r += 4) {
        const int n = n0 + r;
        if (n >= N) break;
        const uint32_t w = (uint32_t)tile[4 * lane][r] | ((uint32_t)tile[4 * lane + 1][r] << 8) |
                           ((uint32_t)tile[4 * lane + 2][r] << 16) | ((uint32_t)tile[4 * lane + 3][r] << 24);
        *reinterpret_cast<uint32_t *>(dst + ((size_t)g * N + n) * kTileFrames + lane * 4) = w;
    }
}

// tiles [G][N][256] -> frame-major [B][N]
__global__ __launch_bounds__(256) void transpose_out_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int B, int N)
{
    __shared__ uint8_t tile[kTileFrames][64 + 4];
    const int g = blockIdx.y, n0 = blockIdx.x * 64, t = threadIdx.x;
    const int lane = t & 63;
    for (int r = t >> 6; r < 64; r += 4) {
        const int n = n0 + r;
        uint32_t w = 0;
        if (n < N) w = *reinterpret_cast<const uint32_t *>(src + ((size_t)g * N + n) * kTileFrames + lane * 4);
        tile[4 * lane][r] = (uint8_t)w; tile[4 * lane + 1][r] = (uint8_t)(w >> 8);
        tile[4 * lane + 2][r] = (uint8_t)(w >> 16); tile[4 * lane + 3][r] = (uint8_t)(w >> 24);
    }
    __syncthreads();
    const int nn = t & 63, fq = t >> 6;
    for (int f = fq; f < kTileFrames; f += 4) {
        const int fr = g * kTileFrames + f, n = n0 + nn;
        if (fr < B && n < N) dst[(size_t)fr * N + n] = tile[f][nn];
    }
}

// msgs[g][e][:] = msg0[g][v(e)][:]  (src/LDPC_Code_LUT.cpp:284-289); one wave per VN
__global__ __launch_bounds__(256) void init_edges_kernel(const uint8_t *__restrict__ msg0_t, uint8_t *__restrict__ msgs,
                                                         const int32_t *__restrict__ vn_ptr, int N, int E)
{
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6), g = blockIdx.y;
    if (v >= N) return;
    const uint32_t w = *reinterpret_cast<const uint32_t *>(msg0_t + ((size_t)g * N + v) * kTileFrames + lane * 4);
    const int e0 = vn_ptr[v], e1 = vn_ptr[v + 1];
    for (int e = e0; e < e1; e++) *reinterpret_cast<uint32_t *>(msgs + ((size_t)g * E + e) * kTileFrames + lane * 4) = w;
}

// hard[g][v][:] = cha[g][v][:] < nz   (src/LDPC_Code_LUT.cpp:275)
__global__ __launch_bounds__(256) void hard_from_labels_kernel(const uint8_t *__restrict__ cha_t, uint8_t *__restrict__ hard,
                                                               size_t n_words, int nz)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_words) reinterpret_cast<uint32_t *>(hard)[i] = swar_lt(reinterpret_cast<const uint32_t *>(cha_t)[i], (uint32_t)nz);
}

// parity of every check over the hard decisions (src/LDPC_Code_LUT.cpp:455-469): vfail |= syndrome
__global__ __launch_bounds__(256) void syndrome_bits_kernel(const uint8_t *__restrict__ hard, const uint32_t *__restrict__ state_w,
                                                            uint32_t *__restrict__ vfail_w, const int32_t *__restrict__ cn_ptr,
                                                            const int32_t *__restrict__ cn_vn, int M, int N, int checks_per_wave)
{
    const int lane = threadIdx.x & 63, g = blockIdx.y;
    const uint32_t amask = swar_zero_mask(state_w[g * kWave + lane]);
    if (wave_all_zero(amask)) return;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    int c0 = w * checks_per_wave, c1 = c0 + checks_per_wave;
    if (c1 > M) c1 = M;
    uint32_t fail = 0;
    for (int c = c0; c < c1; c++) {
        uint32_t s = 0;
        for (int k = cn_ptr[c]; k < cn_ptr[c + 1]; k++)
            s ^= *reinterpret_cast<const uint32_t *>(hard + ((size_t)g * N + cn_vn[k]) * kTileFrames + lane * 4);
        fail |= s;
    }
    fail &= amask & 0x01010101u;
    if (fail) atomicOr(&vfail_w[g * kWave + lane], fail);
}

// per-frame state machine between passes
//   mode 0 (start)      : state = f < B ? ACTIVE : PAD ; iters = 0 ; vfail = 0
//   mode 1 (pisc)       : ACTIVE && !vfail -> DONE_PISC, iters = 0
//   mode 2 (psc, value) : ACTIVE && !vfail -> DONE_PSC , iters = value
//   mode 3 (end, value) : ACTIVE -> iters = vfail ? -value : +value
// vfail is cleared in every mode.
__global__ __launch_bounds__(256) void frame_state_kernel(uint8_t *__restrict__ state, uint8_t *__restrict__ vfail,
                                                          int32_t *__restrict__ iters, int B, int Bpad, int mode, int value)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= Bpad) return;
    if (mode == 0) { state[f] = f < B ? ST_ACTIVE : ST_PAD; iters[f] = 0; vfail[f] = 0; return; }
    const uint8_t s = state[f], vf = vfail[f];
    vfail[f] = 0;
    if (s != ST_ACTIVE) return;
    if (mode == 1) { if (!vf) { state[f] = ST_DONE_PISC; iters[f] = 0; } }
    else if (mode == 2) { if (!vf) { state[f] = ST_DONE_PSC; iters[f] = value; } }
    else iters[f] = vf ? -value : value;
}

// quant_nonlin (src/common.cpp:120-138) for the decode(vec llr) entry: label = number of
// leading boundaries strictly below x; mode 0: msg label from qb_msg, mode 1: map[cha label]
__global__ __launch_bounds__(256) void quantize_llr_kernel(const double *__restrict__ llr, size_t n, const double *__restrict__ qb_cha,
                                                           int n_cha, const double *__restrict__ qb_msg, int n_msg, int mode,
                                                           const int32_t *__restrict__ map, uint8_t *__restrict__ cha, uint8_t *__restrict__ msg)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double x = llr[i];
    int a = 0;
    for (int k = 0; k < n_cha; k++) { if (x > qb_cha[k]) a++; else break; }
    cha[i] = (uint8_t)a;
    if (mode == 0) {
        int m = 0;
        for (int k = 0; k < n_msg; k++) { if (x > qb_msg[k]) m++; else break; }
        msg[i] = (uint8_t)m;
    } else msg[i] = (uint8_t)map[a];
}

}  // namespace lutldpc
