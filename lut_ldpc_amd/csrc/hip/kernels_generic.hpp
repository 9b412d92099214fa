// kernels_generic.hpp -- generic (any tree, any degree) node-update kernels and the layout
// kernels.  These are the correctness baseline every specialised kernel is tested against.
// All kernels are templated on PACK (1 = byte rows, 2 = nibble rows; kernels_common.hpp).
#pragma once
#include "kernels_common.hpp"
#include "lut_program.hpp"

namespace lutldpc {

// ------------------------------------------------------------------------------------------
// Node-program interpreter.  KIND: TT_VAR (variable-node pass, src/LDPC_Code_LUT.cpp:404-414),
// TT_CHK (CHKTREE check pass, :416-426), TT_DEC (decision pass, :428-434,340-344).
// One wave per block; the program's value slots live in LDS as [slot][lane] dwords (four
// frames per dword), so a slot access is a conflict-free ds_read/write_b32.  With nibble rows the
// program runs once per half and the first half's outputs wait in LDS.  Tables of the class are
// staged into LDS when they fit (LDS_TAB), else read through L1/L2.
// LDS: [slots_lds][64] value slots | [n_out_max][64] staged outputs (PACK = 2) | tables
// ------------------------------------------------------------------------------------------
template <int KIND, bool LDS_TAB, int PACK>
__global__ __launch_bounds__(64) void tree_pass_kernel(
    const PassParams *__restrict__ Pp, uint8_t *__restrict__ msgs, const uint8_t *__restrict__ cha, uint8_t *__restrict__ hard,
    const uint32_t *__restrict__ state_w, uint32_t *__restrict__ vfail_w, const Op *__restrict__ ops,
    const uint8_t *__restrict__ tables, const int32_t *__restrict__ node_list,
    const int32_t *__restrict__ node_ptr,   // VAR/DEC: first edge of each VN ; CHK: offset of each CN in cn_idx
    const int32_t *__restrict__ cn_idx, int out_slots)
{
    extern __shared__ uint32_t lds[];
    const PassParams &P = *Pp;           // (pass parameters in device memory: the argument segment stays a handful of pointers)
    const int lane = threadIdx.x;
    const int g = blockIdx.x / P.blocks_per_group;
    const int b = blockIdx.x - g * P.blocks_per_group;
    const int si = find_seg(P, b);
    const PassSeg S = P.seg[si];

    uint32_t amask[PACK];
    if (load_active<PACK>(state_w, g, lane, amask)) return;
    const uint32_t smask = pack_masks<PACK>(amask);

    uint32_t *slot = lds;                                    // [slots_lds][64]
    uint32_t *ostage = lds + P.slots_lds * kWave;            // [out_slots][64]
    const uint8_t *tab;
    if (LDS_TAB) {
        uint32_t *lt = ostage + out_slots * kWave;
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
    uint32_t fail[PACK];
#pragma unroll
    for (int h = 0; h < PACK; h++) fail[h] = 0;

    for (int i = first; i < last; i++) {
        const int node = node_list[S.node_off + i];
        const int p0 = node_ptr[node];
        uint32_t neg_ref[PACK], dec_bit[PACK];
#pragma unroll
        for (int h = 0; h < PACK; h++) {
            // ---- gather inputs of this half into slots 0..n_in-1
            if (KIND == TT_CHK) {
                uint32_t par = 0;
                for (int k = 0; k < S.deg; k++) {
                    const int e = cn_idx[p0 + k];
                    const uint32_t x = unpack_half<PACK>(*reinterpret_cast<const uint32_t *>(msgs + (gE + (size_t)e) * kRowBytes + lane * 4), h);
                    slot[k * kWave + lane] = x;
                    par ^= swar_lt(x, (uint32_t)P.nz);
                }
                if (P.check) fail[h] |= par;
            } else {
                for (int k = 0; k < S.deg; k++)
                    slot[k * kWave + lane] = unpack_half<PACK>(*reinterpret_cast<const uint32_t *>(msgs + (gE + (size_t)(p0 + k)) * kRowBytes + lane * 4), h);
                slot[S.deg * kWave + lane] = unpack_half<PACK>(*reinterpret_cast<const uint32_t *>(cha + (gN + (size_t)node) * kRowBytes + lane * 4), h);
            }
            // ---- run the program
            uint32_t have_ref = 0;
            neg_ref[h] = 0; dec_bit[h] = 0;
            for (int o = 0; o < S.n_ops; o++) {
                const Op &op = prog[o];
                uint32_t lab[4] = {0, 0, 0, 0}, par[4] = {0, 0, 0, 0};
                for (int c = 0; c < op.nchild; c++) {
                    const uint32_t x = slot[op.child[c] * kWave + lane];
                    const uint32_t m = op.mult[c];
                    if (op.kind == 1) {
                        const uint32_t hh = op.childK[c] >> 1;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t v = (x >> (8 * j)) & 0xFFu;
                            const bool ng = v < hh;
                            par[j] ^= ng ? 1u : 0u;
                            lab[j] += m * (ng ? (hh - 1 - v) : (v - hh));
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
                    if (KIND == TT_DEC) dec_bit[h] = swar_lt(r, 1u);       // bit = (label < 1), src/LDPC_Code_LUT.cpp:342
                    else {
                        ostage[op.out_idx * kWave + lane] = (PACK == 2 && h == 1) ? (ostage[op.out_idx * kWave + lane] | (r << 4)) : r;
                        if (KIND == TT_VAR && (P.check || P.write_hard)) {
                            const uint32_t ng = swar_lt(r, (uint32_t)P.nz);
                            if (!have_ref) { neg_ref[h] = ng; have_ref = 1; }
                            else fail[h] |= ng ^ neg_ref[h];
                        }
                    }
                }
            }
        }
        // ---- write back (in place; frames that already terminated keep their old value)
        if (KIND == TT_DEC) {
            store_row_masked<PACK>(reinterpret_cast<uint32_t *>(hard + (gN + (size_t)node) * kRowBytes + lane * 4), pack_halves<PACK>(dec_bit), smask);
        } else {
            for (int k = 0; k < S.n_out; k++) {
                size_t row;
                if (KIND == TT_CHK) row = gE + (size_t)cn_idx[p0 + k];
                else row = gE + (size_t)(p0 + k);
                uint32_t *p = reinterpret_cast<uint32_t *>(msgs + row * kRowBytes + lane * 4);
                store_row_masked<PACK>(p, ostage[k * kWave + lane], smask);
            }
            if (KIND == TT_VAR && P.write_hard)
                store_row_masked<PACK>(reinterpret_cast<uint32_t *>(hard + (gN + (size_t)node) * kRowBytes + lane * 4), pack_halves<PACK>(neg_ref), smask);
        }
    }
    if (KIND != TT_DEC && P.check) flag_frames<PACK>(vfail_w, P.vfail_stride_w, g, lane, fail, amask);
}

// ------------------------------------------------------------------------------------------
// Min-sum check pass, any degree (src/LDPC_Code_LUT.cpp:355-402).  Two sweeps over the
// check's rows: the first accumulates min1/min2/sign product, the second re-reads each row
// (L1/L2 hit) and writes the extrinsic message.  Byte-serial arithmetic; the specialised
// kernel in kernels_fast.hpp keeps the rows in registers and works on packed bytes.
// out magnitude for edge k is min over the others = (mag_k == min1 ? min2 : min1): when the
// minimum occurs twice min2 == min1, so the reference's min_idx bookkeeping is not needed.
// ------------------------------------------------------------------------------------------
template <int PACK>
__global__ __launch_bounds__(64) void cn_minsum_generic_kernel(
    const PassParams *__restrict__ Pp, uint8_t *__restrict__ msgs, const uint32_t *__restrict__ state_w, uint32_t *__restrict__ vfail_w,
    const int32_t *__restrict__ node_list, const int32_t *__restrict__ cn_ptr, const int32_t *__restrict__ cn_idx)
{
    const PassParams &P = *Pp;
    const int lane = threadIdx.x;
    const int g = blockIdx.x / P.blocks_per_group;
    const int b = blockIdx.x - g * P.blocks_per_group;
    const PassSeg S = P.seg[find_seg(P, b)];
    uint32_t amask[PACK];
    if (load_active<PACK>(state_w, g, lane, amask)) return;
    const uint32_t smask = pack_masks<PACK>(amask);
    const size_t gE = (size_t)g * (size_t)P.E;
    const int first = (b - S.block_begin) * P.nodes_per_block;
    int last = first + P.nodes_per_block;
    if (last > S.n_nodes) last = S.n_nodes;
    const int nz = P.nz;
    uint32_t fail[PACK];
#pragma unroll
    for (int h = 0; h < PACK; h++) fail[h] = 0;
    for (int i = first; i < last; i++) {
        const int c = node_list[S.node_off + i];
        const int p0 = cn_ptr[c];
        int min1[PACK][4], min2[PACK][4];
        uint32_t sp[PACK];
#pragma unroll
        for (int h = 0; h < PACK; h++) {
            sp[h] = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { min1[h][j] = nz; min2[h][j] = nz; }
        }
        for (int k = 0; k < S.deg; k++) {
            const uint32_t xr = *reinterpret_cast<const uint32_t *>(msgs + (gE + (size_t)cn_idx[p0 + k]) * kRowBytes + lane * 4);
#pragma unroll
            for (int h = 0; h < PACK; h++) {
                const uint32_t x = unpack_half<PACK>(xr, h);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int v = (int)((x >> (8 * j)) & 0xFFu);
                    int t;
                    if (v < nz) { sp[h] ^= 1u << (8 * j); t = nz - 1 - v; } else t = v - nz;
                    if (t < min1[h][j]) { min2[h][j] = min1[h][j]; min1[h][j] = t; }
                    else if (t < min2[h][j]) min2[h][j] = t;
                }
            }
        }
        if (P.check) {
#pragma unroll
            for (int h = 0; h < PACK; h++) fail[h] |= sp[h];
        }
        for (int k = 0; k < S.deg; k++) {
            uint32_t *row = reinterpret_cast<uint32_t *>(msgs + (gE + (size_t)cn_idx[p0 + k]) * kRowBytes + lane * 4);
            const uint32_t xr = *row;
            uint32_t rr[PACK];
#pragma unroll
            for (int h = 0; h < PACK; h++) {
                const uint32_t x = unpack_half<PACK>(xr, h);
                uint32_t r = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int v = (int)((x >> (8 * j)) & 0xFFu);
                    const bool ng = v < nz;
                    const int t = ng ? nz - 1 - v : v - nz;
                    const int m = (t == min1[h][j]) ? min2[h][j] : min1[h][j];
                    const uint32_t s = ((sp[h] >> (8 * j)) & 1u) ^ (ng ? 1u : 0u);
                    const int o = s ? nz - 1 - m : nz + m;
                    r |= (uint32_t)(o & (PACK == 2 ? 0x0F : 0xFF)) << (8 * j);
                }
                rr[h] = r;
            }
            *row = bfi(smask, pack_halves<PACK>(rr), xr);
        }
    }
    if (P.check) flag_frames<PACK>(vfail_w, P.vfail_stride_w, g, lane, fail, amask);
}

// ------------------------------------------------------------------------------------------
// Layout kernels
// ------------------------------------------------------------------------------------------
// frame-major [B][N] -> rows [G][N][256 B]; labels are clamped to < limit, pad frames get 0.
template <int PACK>
__global__ __launch_bounds__(256) void transpose_in_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                           int B, int N, int limit)
{
    constexpr int F = kRowBytes * PACK;                     // frames per group
    __shared__ uint8_t tile[F][32 + 4];
    const int g = blockIdx.y, n0 = blockIdx.x * 32, t = threadIdx.x;
    const int nn = t & 31, fq = t >> 5;
    for (int f = fq; f < F; f += 8) {
        const int fr = g * F + f, n = n0 + nn;
        uint8_t v = 0;
        if (fr < B && n < N) { v = src[(size_t)fr * N + n]; if (v >= limit) v = (uint8_t)(limit - 1); }
        tile[f][nn] = v;
    }
    __syncthreads();
    const int lane = t & 63;
    for (int r = t >> 6; r < 32; r += 4) {
        const int n = n0 + r;
        if (n >= N) break;
        uint32_t w[PACK];
#pragma unroll
        for (int h = 0; h < PACK; h++) {
            const int f0 = lane * 4 * PACK + 4 * h;
            w[h] = (uint32_t)tile[f0][r] | ((uint32_t)tile[f0 + 1][r] << 8) | ((uint32_t)tile[f0 + 2][r] << 16) | ((uint32_t)tile[f0 + 3][r] << 24);
        }
        *reinterpret_cast<uint32_t *>(dst + ((size_t)g * N + n) * kRowBytes + lane * 4) = pack_halves<PACK>(w);
    }
}

// rows [G][N][256 B] -> frame-major [B][N]
template <int PACK>
__global__ __launch_bounds__(256) void transpose_out_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int B, int N)
{
    constexpr int F = kRowBytes * PACK;
    __shared__ uint8_t tile[F][32 + 4];
    const int g = blockIdx.y, n0 = blockIdx.x * 32, t = threadIdx.x;
    const int lane = t & 63;
    for (int r = t >> 6; r < 32; r += 4) {
        const int n = n0 + r;
        uint32_t x = 0;
        if (n < N) x = *reinterpret_cast<const uint32_t *>(src + ((size_t)g * N + n) * kRowBytes + lane * 4);
#pragma unroll
        for (int h = 0; h < PACK; h++) {
            const uint32_t w = unpack_half<PACK>(x, h);
            const int f0 = lane * 4 * PACK + 4 * h;
            tile[f0][r] = (uint8_t)w; tile[f0 + 1][r] = (uint8_t)(w >> 8);
            tile[f0 + 2][r] = (uint8_t)(w >> 16); tile[f0 + 3][r] = (uint8_t)(w >> 24);
        }
    }
    __syncthreads();
    const int nn = t & 31, fq = t >> 5;
    for (int f = fq; f < F; f += 8) {
        const int fr = g * F + f, n = n0 + nn;
        if (fr < B && n < N) dst[(size_t)fr * N + n] = tile[f][nn];
    }
}

// ------------------------------------------------------------------------------------------
// Vectorised transposes (N % 4 == 0, 4-byte aligned frame-major buffer): a block moves 128 nodes x one
// frame group through LDS as DWORDS (4 nodes of one frame each).
//   global side : 32 lanes x 4 B = 128 contiguous bytes per frame;
//   LDS         : tile[f][c ^ key(f)], key(f) = (f / frames-per-lane) & 31: the XOR swizzle makes both the
//                 frame-major side (32 lanes, one frame, all quads c) and the row side (lane L reads its own
//                 frames 4*PACK*L + k of one quad c) hit 32 different banks;
//   registers   : 4x4 byte transposes with v_perm_b32 turn "4 nodes of 4 frames" into "4 frames of one node",
//                 the dword a lane owns in a row (PACK = 2: two of them merged as low / high nibbles).
// XCD-aware block order of the transposes.  A block moves 128 nodes x one frame group; its frame-major side is one 128-byte
// segment per frame at a stride of N bytes, so unless N is a multiple of 128 every segment straddles two 128-byte lines and
// shares each with the block of the neighbouring node range.  Blocks are dealt round-robin to the 8 XCDs (one L2 each): with the
// plain (x, y) order the two halves of a line are fetched (written) by two different L2s -- measured 3.72 GB read for 2.12 GB of
// input.  Remapped, the blocks of one XCD cover a CONTIGUOUS range of (group, node block): the neighbour that shares a line runs
// on the same XCD a few blocks later and finds it in that L2.
__device__ __forceinline__ void xcd_block(int &bx, int &by) {
    const int gx = (int)gridDim.x, total = gx * (int)gridDim.y, lin = (int)blockIdx.y * gx + (int)blockIdx.x;
    const int per = total / 8;
    int nl = lin;
    if (lin < per * 8) nl = (lin & 7) * per + (lin >> 3);         // (the last total % 8 blocks keep their place)
    bx = nl % gx; by = nl / gx;
}

__device__ __forceinline__ void transpose4x4(uint32_t (&d)[4]) {
    const uint32_t a = __builtin_amdgcn_perm(d[1], d[0], 0x05010400u), b = __builtin_amdgcn_perm(d[1], d[0], 0x07030602u);
    const uint32_t c = __builtin_amdgcn_perm(d[3], d[2], 0x05010400u), e = __builtin_amdgcn_perm(d[3], d[2], 0x07030602u);
    d[0] = __builtin_amdgcn_perm(c, a, 0x05040100u); d[1] = __builtin_amdgcn_perm(c, a, 0x07060302u);
    d[2] = __builtin_amdgcn_perm(e, b, 0x05040100u); d[3] = __builtin_amdgcn_perm(e, b, 0x07060302u);
}

template <int PACK>
__global__ __launch_bounds__(256) void transpose_in_vec_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int B, int N, int limit)
{
    constexpr int F = kRowBytes * PACK, FPL = 4 * PACK;     // frames per group / per lane
    __shared__ uint32_t tile[F * 32];
    int bx, by;
    xcd_block(bx, by);
    const int g = by, n0 = bx * 128, t = threadIdx.x;
    const int c = t & 31, fq = t >> 5;
    const uint32_t lim = (uint32_t)(limit - 1);
    for (int f0 = fq; f0 < F; f0 += 64) {                   // eight loads in flight per thread
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int fr = g * F + f0 + 8 * j, n = n0 + 4 * c;
            v[j] = (fr < B && n < N) ? *reinterpret_cast<const uint32_t *>(src + (size_t)fr * N + n) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int f = f0 + 8 * j;
            const uint32_t b0 = min(v[j] & 0xFFu, lim), b1 = min((v[j] >> 8) & 0xFFu, lim), b2 = min((v[j] >> 16) & 0xFFu, lim), b3 = min(v[j] >> 24, lim);
            tile[f * 32 + (c ^ ((f / FPL) & 31))] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
        }
    }
    __syncthreads();
    const int lane = t & 63;
    for (int q = t >> 6; q < 32; q += 4) {                  // node quad q: nodes n0 + 4q .. +3
        if (n0 + 4 * q >= N) break;
        uint32_t out[4] = {0, 0, 0, 0};
#pragma unroll
        for (int h = 0; h < PACK; h++) {
            uint32_t d[4];
#pragma unroll
            for (int k = 0; k < 4; k++) d[k] = tile[(FPL * lane + 4 * h + k) * 32 + (q ^ (lane & 31))];
            transpose4x4(d);
#pragma unroll
            for (int k = 0; k < 4; k++) out[k] |= d[k] << (4 * h);
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            *reinterpret_cast<uint32_t *>(dst + ((size_t)g * N + n0 + 4 * q + k) * kRowBytes + lane * 4) = out[k];
    }
}

template <int PACK>
__global__ __launch_bounds__(256) void transpose_out_vec_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int B, int N)
{
    constexpr int F = kRowBytes * PACK, FPL = 4 * PACK;
    __shared__ uint32_t tile[F * 32];
    int bx, by;
    xcd_block(bx, by);
    const int g = by, n0 = bx * 128, t = threadIdx.x;
    const int lane = t & 63;
    for (int q = t >> 6; q < 32; q += 4) {
        uint32_t x[4] = {0, 0, 0, 0};
        if (n0 + 4 * q < N) {
#pragma unroll
            for (int k = 0; k < 4; k++) x[k] = *reinterpret_cast<const uint32_t *>(src + ((size_t)g * N + n0 + 4 * q + k) * kRowBytes + lane * 4);
        }
#pragma unroll
        for (int h = 0; h < PACK; h++) {
            uint32_t d[4];
#pragma unroll
            for (int k = 0; k < 4; k++) d[k] = unpack_half<PACK>(x[k], h);
            transpose4x4(d);                                 // d[k] = 4 nodes of frame FPL*lane + 4h + k
#pragma unroll
            for (int k = 0; k < 4; k++) tile[(FPL * lane + 4 * h + k) * 32 + (q ^ (lane & 31))] = d[k];
        }
    }
    __syncthreads();
    const int c = t & 31, fq = t >> 5;
    for (int f = fq; f < F; f += 8) {
        const int fr = g * F + f, n = n0 + 4 * c;
        if (fr < B && n < N) *reinterpret_cast<uint32_t *>(dst + (size_t)fr * N + n) = tile[f * 32 + (c ^ ((f / FPL) & 31))];
    }
}

// msgs[g][e][:] = msg0[g][v(e)][:]  (src/LDPC_Code_LUT.cpp:284-289); one wave per VN (rows are copied whole)
__global__ __launch_bounds__(256) void init_edges_kernel(const uint8_t *__restrict__ msg0_t, uint8_t *__restrict__ msgs,
                                                         const int32_t *__restrict__ vn_ptr, int N, int E)
{
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6), g = blockIdx.y;
    if (v >= N) return;
    const uint32_t w = *reinterpret_cast<const uint32_t *>(msg0_t + ((size_t)g * N + v) * kRowBytes + lane * 4);
    const int e0 = vn_ptr[v], e1 = vn_ptr[v + 1];
    for (int e = e0; e < e1; e++) *reinterpret_cast<uint32_t *>(msgs + ((size_t)g * E + e) * kRowBytes + lane * 4) = w;
}

// hard[g][v][:] = cha[g][v][:] < nz   (src/LDPC_Code_LUT.cpp:275)
template <int PACK>
__global__ __launch_bounds__(256) void hard_from_labels_kernel(const uint8_t *__restrict__ cha_t, uint8_t *__restrict__ hard,
                                                               size_t n_words, int nz)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_words) return;
    const uint32_t x = reinterpret_cast<const uint32_t *>(cha_t)[i];
    uint32_t r[PACK];
#pragma unroll
    for (int h = 0; h < PACK; h++) r[h] = swar_lt(unpack_half<PACK>(x, h), (uint32_t)nz);
    reinterpret_cast<uint32_t *>(hard)[i] = pack_halves<PACK>(r);
}

// hard[g][v][:] = cha[g][v][:] < nz for the frames that passed the test on the channel decisions (ST_DONE_PISC) only: used at
// the end of a decode whose compaction moved frames (the channel rows travel with their frame, the decided-bit rows do not)
template <int PACK>
__global__ __launch_bounds__(256) void hard_from_labels_masked_kernel(const uint8_t *__restrict__ cha_t, uint8_t *__restrict__ hard,
                                                                      const uint32_t *__restrict__ state_w, int N, int nz, int g0)
{
    const int lane = threadIdx.x & 63, g = g0 + blockIdx.y;
    uint32_t m[PACK], any = 0;
#pragma unroll
    for (int h = 0; h < PACK; h++) { m[h] = swar_zero_mask(state_w[frame_word<PACK>(g, lane, h)] ^ (ST_DONE_PISC * 0x01010101u)); any |= m[h]; }
    if (wave_all_zero(any)) return;
    const uint32_t smask = pack_masks<PACK>(m);
    const int w0 = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), step = (int)gridDim.x * 4;
    for (int v = w0; v < N; v += step) {
        const uint32_t x = *reinterpret_cast<const uint32_t *>(cha_t + ((size_t)g * N + (size_t)v) * kRowBytes + lane * 4);
        uint32_t r[PACK];
#pragma unroll
        for (int h = 0; h < PACK; h++) r[h] = swar_lt(unpack_half<PACK>(x, h), (uint32_t)nz);
        store_row_masked<PACK>(reinterpret_cast<uint32_t *>(hard + ((size_t)g * N + (size_t)v) * kRowBytes + lane * 4), pack_halves<PACK>(r), smask);
    }
}

// Decided bits of the frames that left through the exit test (ST_DONE_PSC), recovered ONCE at the end of the decode from
// their frozen messages instead of being stored by every variable pass.  The reference returns the unanimous signs of the
// variable-to-check messages of iteration ii (src/LDPC_Code_LUT.cpp:327-329,437-452); here the frame is frozen one check
// pass later, with the check-to-variable messages of iteration ii+1 in place.  For a frame that PASSED the test every
// check is satisfied, and the min-sum output on an edge then carries the sign of its own input on that edge
// (chk_update_minsum, :355-402: sign_msg = sign_prod ^ own sign, sign_prod = 0), i.e. the unanimous sign of the variable
// node: bit v = (label on the node's first edge < nz).  Exact for min-sum check updates with one message alphabet; the
// nodes updated inside the check pass (chain fusion, `skip`) keep the bits that pass stored for them.
// g0: first frame group of the launch (grid.y groups from there); ctl: when given (a compaction check point), ctl[1] != 0
// means "no permutation at this check point": nothing to save yet.
template <int PACK>
__global__ __launch_bounds__(256) void hard_from_frozen_kernel(const uint8_t *__restrict__ msgs, uint8_t *__restrict__ hard,
                                                               const uint32_t *__restrict__ state_w, const int32_t *__restrict__ vn_ptr,
                                                               const uint8_t *__restrict__ skip, int N, int E, int nz, int g0, const int32_t *__restrict__ ctl)
{
    if (ctl && ctl[1]) return;
    const int lane = threadIdx.x & 63, g = g0 + blockIdx.y;
    // which frames of this lane left through the exit test: the same for every node of the group
    uint32_t m[PACK], any = 0;
#pragma unroll
    for (int h = 0; h < PACK; h++) { m[h] = swar_zero_mask(state_w[frame_word<PACK>(g, lane, h)] ^ (ST_DONE_PSC * 0x01010101u)); any |= m[h]; }
    if (wave_all_zero(any)) return;
    const uint32_t smask = pack_masks<PACK>(m);
    // a small fixed grid walks the nodes (a check point that does not permute must cost microseconds)
    const int w0 = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), step = (int)gridDim.x * 4;
#pragma unroll 4
    for (int v = w0; v < N; v += step) {
        if (skip && skip[v]) continue;
        const uint32_t x = *reinterpret_cast<const uint32_t *>(msgs + ((size_t)g * E + (size_t)vn_ptr[v]) * kRowBytes + lane * 4);
        uint32_t r[PACK];
#pragma unroll
        for (int h = 0; h < PACK; h++) r[h] = swar_lt(unpack_half<PACK>(x, h), (uint32_t)nz);
        store_row_masked<PACK>(reinterpret_cast<uint32_t *>(hard + ((size_t)g * N + (size_t)v) * kRowBytes + lane * 4), pack_halves<PACK>(r), smask);
    }
}

// Decided bits of the variable nodes that are updated INSIDE the check pass (chain fusion), for the frames that left through
// the exit test: their edges hold new variable-to-check messages, not the frozen check-to-variable ones, so their bits
// cannot be read off them -- but a frame that passed the test satisfies every check, and each such node is the only unknown
// of one check once the run is walked in order: bit(forward node of check j) = XOR of the bits of all other nodes of check j
// (the accumulator of a dual-diagonal code, read backwards).  One wave per run of `npw` checks of a class (the same runs as
// the check pass); the bit row of the node shared with the next check travels in a register.  Runs after
// hard_from_frozen_kernel (which supplies every other node), with the same frame mask; replaces the per-pass stores.
//   edges: dense [n_nodes][deg] edge ids of the class, links: [n_nodes][2] = {back node + 1, forward node + 1},
//   edge_vn[e] = variable node of edge e.
template <int PACK>
__global__ __launch_bounds__(256) void chain_hard_kernel(uint8_t *__restrict__ hard, const uint32_t *__restrict__ state_w, const int32_t *__restrict__ edges,
                                                         const int32_t *__restrict__ links, const int32_t *__restrict__ edge_vn, int n_nodes, int deg, int npw, int N,
                                                         int g0, const int32_t *__restrict__ ctl)
{
    if (ctl && ctl[1]) return;
    const int lane = threadIdx.x & 63, g = g0 + blockIdx.y;
    uint32_t m[PACK], any = 0;
#pragma unroll
    for (int h = 0; h < PACK; h++) { m[h] = swar_zero_mask(state_w[frame_word<PACK>(g, lane, h)] ^ (ST_DONE_PSC * 0x01010101u)); any |= m[h]; }
    if (wave_all_zero(any)) return;
    const uint32_t smask = pack_masks<PACK>(m);
    const int runs = (n_nodes + npw - 1) / npw;
    const int w0 = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), step = (int)gridDim.x * 4;
    uint8_t *hb = hard + (size_t)g * (size_t)N * kRowBytes + lane * 4;
    for (int r = w0; r < runs; r += step) {
        const int j0 = r * npw, j1 = min(j0 + npw, n_nodes);
        uint32_t carry = 0;                                   // bit row of the node shared with the previous check of the run
        for (int j = j0; j < j1; j++) {
            const int lb = links[2 * (size_t)j], lf = links[2 * (size_t)j + 1];
            if (!lf) continue;                                // no node of this check is updated inside the check pass towards the next one
            uint32_t acc = lb ? carry : 0u;                   // (lb != 0: the back node was computed one step ago, same run by construction)
            for (int k = 0; k < deg; k++) {
                const int v = edge_vn[edges[(size_t)j * deg + k]];
                if (v == lf - 1 || (lb && v == lb - 1)) continue;
                acc ^= *reinterpret_cast<const uint32_t *>(hb + (size_t)v * kRowBytes);
            }
            store_row_masked<PACK>(reinterpret_cast<uint32_t *>(hb + (size_t)(lf - 1) * kRowBytes), acc, smask);
            carry = acc;
        }
    }
}

// parity of every check over the hard decisions (src/LDPC_Code_LUT.cpp:455-469): vfail |= syndrome.
// cn_vnf[k] = variable node of check-edge k, bit 31 set on the last edge of its check; padded with 8
// zero entries.  A wave walks the run of edges of its checks eight at a time: ONE vector load fetches
// the eight indices (lanes 0..7), v_readlane turns them into wave-uniform row offsets, the eight row
// loads go out together (a lane offset past the end of the run is out of range: returns 0, neutral
// for the XOR), and the running parity is closed branch-free where the flag is set.
// label_sbit >= 0: the rows are LABEL rows (channel labels), the decided bit of a label is its sign bit inverted (label < nz,
// nz = 1 << label_sbit: src/LDPC_Code_LUT.cpp:275) -- the test on the channel decisions needs no decided-bit rows.
// grp_skip (optional): groups whose frames have ALL failed already are skipped; grp_out (optional, one wave per group: the probe
// launch over the first checks): set to 1 when every active frame of the group has failed.  At an SNR where the channel
// decisions never form a codeword the probe fails every frame within its 64 checks and the full launch returns at once.
template <int PACK>
__global__ __launch_bounds__(256) void syndrome_bits_kernel(const uint8_t *__restrict__ hard, const uint32_t *__restrict__ state_w,
                                                            uint32_t *__restrict__ vfail_w, const int32_t *__restrict__ cn_ptr,
                                                            const uint32_t *__restrict__ cn_vnf, int M, int N, int checks_per_wave, int vfail_stride_w,
                                                            int label_sbit = -1, const int32_t *__restrict__ grp_skip = nullptr, int32_t *__restrict__ grp_out = nullptr)
{
    const int lane = threadIdx.x & 63, g = blockIdx.y;
    if (grp_skip && grp_skip[g]) return;
    uint32_t amask[PACK];
    if (load_active<PACK>(state_w, g, lane, amask)) return;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    int c0 = w * checks_per_wave, c1 = c0 + checks_per_wave;
    if (c1 > M) c1 = M;
    if (c0 >= c1) return;
    const rsrc_t hb = make_rsrc(hard + (size_t)g * (size_t)N * kRowBytes, (uint32_t)N * kRowBytes);
    const uint32_t lane4 = (uint32_t)lane * 4u;
    const int k0 = cn_ptr[c0], k1 = cn_ptr[c1];
    uint32_t acc = 0, s = 0;
    for (int k = k0; k < k1; k += 8) {
        const uint32_t mine = cn_vnf[k + (lane & 7)];
        uint32_t x[8], last[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)mine, j);
            last[j] = (k + j < k1) ? (uint32_t)((int32_t)e >> 31) : 0u;                      // 0 or ~0, wave-uniform
            x[j] = ld_row(hb, (e & 0x7FFFFFFFu) * kRowBytes, lane4 | (k + j < k1 ? 0u : 0x80000000u));
            if (label_sbit >= 0) x[j] = (k + j < k1) ? ((~x[j] >> label_sbit) & (PACK == 2 ? 0x11111111u : 0x01010101u)) : 0u;     // (wave-uniform)
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            s ^= x[j];
            acc |= s & last[j];
            s &= ~last[j];
        }
    }
    uint32_t fail[PACK];
#pragma unroll
    for (int h = 0; h < PACK; h++) fail[h] = unpack_half<PACK>(acc, h);
    flag_frames<PACK>(vfail_w, vfail_stride_w, g, lane, fail, amask);
    if (grp_out) {
        bool all_failed = true;
#pragma unroll
        for (int h = 0; h < PACK; h++) all_failed = all_failed && (((fail[h] | ~amask[h]) & 0x01010101u) == 0x01010101u);
        if (__ballot(!all_failed) == 0ull && lane == 0) grp_out[g] = 1;
    }
}

// per-frame state machine between passes
//   mode 0 (start)      : state = f < B ? ACTIVE : PAD ; iters = 0 ; vfail = 0
//   mode 1 (pisc)       : ACTIVE && !vfail -> DONE_PISC, iters = 0
//   mode 2 (psc, value) : ACTIVE && !vfail -> DONE_PSC , iters = value
//   mode 3 (end, value) : ACTIVE -> iters = vfail ? -value : +value
// vfail is cleared in every mode.
__global__ __launch_bounds__(256) void frame_state_kernel(uint8_t *__restrict__ state, uint8_t *__restrict__ vfail,
                                                          int32_t *__restrict__ iters, int B, int f0, int f1, int mode, int value, int vfail_stride)
{
    const int f = f0 + blockIdx.x * 256 + threadIdx.x;      // frames f0 .. f1-1 of the padded batch
    if (f >= f1) return;
    uint8_t vf = 0;                                         // OR of the kVfailSlots copies (flag_frames), then clear them
    if (mode != 0) for (int s = 0; s < kVfailSlots; s++) vf |= vfail[(size_t)s * vfail_stride + f];
    for (int s = 0; s < kVfailSlots; s++) vfail[(size_t)s * vfail_stride + f] = 0;
    if (mode == 0) { state[f] = f < B ? ST_ACTIVE : ST_PAD; iters[f] = 0; return; }
    const uint8_t s = state[f];
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
