// kernels_compact.hpp -- compaction of the surviving frames (as-shipped mode, skewed pipeline).
//
// Early termination freezes a finished frame in place; its lane keeps riding along until the whole
// 512-frame group is done, which rarely happens while a few stragglers need 40+ iterations.  Every few
// iterations the frame SLOTS of a half are therefore permuted -- active frames to the front (stable), all
// others behind them -- so that whole groups fall idle and their waves exit on the first look at the state
// words (load_active).  Nothing is lost: iteration codes, states and pending flags move with their frame,
// `frame_of[slot]` remembers where each frame came from, and at the end of the decode the decided bits and
// iteration codes are put back into the original order.
//
// Default flow (rows kept): the message and channel rows of EVERY frame of the groups that were live move -- the
// frames that just left keep their frozen rows behind the active ones (frames that left earlier sit further back; a
// stable partition never touches them) -- so the decided bits are recovered once at the end of the decode, exactly
// as without compaction, and no decided-bit row exists (or moves) before that.
// One permutation = compact_apply_kernel (one block per half: prefix sum, small per-slot arrays) +
// permute_rows_kernel over the E message rows and the N channel rows (one launch).
// Earlier flow (LUTLDPC_COMPACT_KEEP=0, also used when the decided bits are stored by every variable pass): the
// decided bits of the frames that left since the last permutation are recovered at the check point
// (hard_from_frozen_kernel, frames marked ST_DONE_SAVED), only the ACTIVE frames' rows move, and the N decided-bit
// rows are permuted along.
//
// Whether a permutation pays is decided ON THE DEVICE at every check point (the launch sequence is a fixed hipGraph): it
// reads and rewrites the rows of the live groups -- about  1.3 x live  group-iterations (earlier flow:
// 0.7 (live + new) + 0.3 GH) -- and saves  (live - new) groups x remaining iterations.  compact_decide_kernel permutes
// only when a sizeable share of the live groups falls idle at once and the saving exceeds the cost; otherwise the
// other kernels of the check point return at once (5 us each).
#pragma once
#include "kernels_common.hpp"

namespace lutldpc {

// decode start: frame_of = identity, ctl[half] = {n_active, skip, live groups, live groups before the last permutation}
__global__ __launch_bounds__(256) void compact_init_kernel(int32_t *__restrict__ frame_of, int n, int32_t *__restrict__ ctl, int gh0, int gh1) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) frame_of[i] = i;
    if (i == 0) { ctl[0] = 0; ctl[1] = 1; ctl[2] = gh0; ctl[3] = gh0; ctl[4] = 0; ctl[5] = 1; ctl[6] = gh1; ctl[7] = gh1; }
}

// Check point, step 1 (one block): count the active frames of the half and decide.  ctl = {n_active, skip, live groups, live
// groups before this permutation}.  iters_left: message-passing iterations still to run; a permutation happens when at least
// `min_share` of the live groups fall idle AND the group-iterations saved reach `margin` x the rows moved (header of this
// file; margin 0 = whenever a group falls idle, for the tests).
__global__ __launch_bounds__(1024) void compact_decide_kernel(const uint8_t *__restrict__ state, int s0, int n, int tile_frames, int32_t *__restrict__ ctl,
                                                               int iters_left, float margin, float min_share, int rows_keep)
{
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    int cnt = 0;
    for (int i = t; i < n; i += 1024) cnt += state[s0 + i] == ST_ACTIVE ? 1 : 0;
    for (int o = 32; o; o >>= 1) cnt += __shfl_down(cnt, o);
    if (lane == 0) wsum[w] = cnt;
    __syncthreads();
    if (t == 0) {
        int s = 0;
        for (int k = 0; k < 16; k++) s += wsum[k];
        const int gnew = (s + tile_frames - 1) / tile_frames, live = ctl[2], gh = n / tile_frames;
        const float gain = (float)(live - gnew) * (float)iters_left;
        const float cost = rows_keep ? 1.3f * (float)live : 0.7f * (float)(live + gnew) + 0.3f * (float)gh;
        ctl[0] = s;
        if (gnew < live && (float)(live - gnew) >= min_share * (float)live && gain >= margin * cost) { ctl[1] = 0; ctl[3] = live; ctl[2] = gnew; }
        else ctl[1] = 1;
    }
}

// Check point, step 3 (one block; step 2 is hard_from_frozen_kernel on the frames that left since the last permutation):
// slots [s0, s0 + n): perm[new] = old (absolute slot numbers), active frames first (stable).  Applies the permutation to
// state / iters / frame_of and to the pending flag buffer (its kVfailSlots copies are ORed into copy 0).  Frames that left
// through the exit test are marked ST_DONE_SAVED first: their decided bits are in the hard rows now, their messages may go.
// tmp: 3 * n int32 of scratch.
__global__ __launch_bounds__(1024) void compact_apply_kernel(uint8_t *__restrict__ state, int32_t *__restrict__ iters, int32_t *__restrict__ frame_of,
                                                              uint8_t *__restrict__ vfail_pending, int vfail_stride, int s0, int n,
                                                              int32_t *__restrict__ perm, int32_t *__restrict__ tmp, const int32_t *__restrict__ ctl, int mark_saved)
{
    __shared__ int wsum[16];
    __shared__ int base_act, base_rest;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (ctl[1]) return;                                       // nothing to gain at this check point
    if (t == 0) { base_act = 0; base_rest = ctl[0]; }
    if (mark_saved)
        for (int i = t; i < n; i += 1024) if (state[s0 + i] == ST_DONE_PSC) state[s0 + i] = ST_DONE_SAVED;
    __syncthreads();
    // stable partition, 1024 slots at a time
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i = c0 + t;
        const int act = (i < n && state[s0 + i] == ST_ACTIVE) ? 1 : 0, val = i < n ? 1 : 0;
        // inclusive scan of `act` over the block (wave scan + wave totals)
        int x = act;
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; k++) woff += wsum[k];
        int blk = 0;
        for (int k = 0; k < 16; k++) blk += wsum[k];
        const int before_act = woff + x - act;                  // active frames before slot i within this chunk
        if (val) {
            const int dst = act ? base_act + before_act : base_rest + (t - before_act);
            perm[s0 + dst] = s0 + i;
        }
        __syncthreads();
        if (t == 0) { base_act += blk; base_rest += (n - c0 < 1024 ? n - c0 : 1024) - blk; }
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    // move the per-slot data (through tmp: the permutation is not in place)
    for (int i = t; i < n; i += 1024) {
        const int o = perm[s0 + i];
        uint8_t vf = 0;
        for (int c = 0; c < kVfailSlots; c++) vf |= vfail_pending[(size_t)c * vfail_stride + o];
        tmp[i] = (int)state[o] | ((int)vf << 8);
        tmp[n + i] = iters[o];
        tmp[2 * n + i] = frame_of[o];
    }
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
        state[s0 + i] = (uint8_t)(tmp[i] & 0xFF);
        iters[s0 + i] = tmp[n + i];
        frame_of[s0 + i] = tmp[2 * n + i];
        vfail_pending[s0 + i] = (uint8_t)(tmp[i] >> 8);
        for (int c = 1; c < kVfailSlots; c++) vfail_pending[(size_t)c * vfail_stride + s0 + i] = 0;
    }
}

// slot_of[frame_of[slot]] = slot over [s0, s0 + n): the way back at the end of the decode
__global__ __launch_bounds__(256) void invert_map_kernel(const int32_t *__restrict__ frame_of, int32_t *__restrict__ slot_of, int s0, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) slot_of[frame_of[s0 + i]] = s0 + i;
}
__global__ __launch_bounds__(256) void gather_i32_kernel(const int32_t *__restrict__ src, const int32_t *__restrict__ map, int32_t *__restrict__ dst, int s0, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[s0 + i] = src[map[s0 + i]];
}

// rows[g][r][256 B], groups g0 .. g0+GH-1 (one half): row r of every group is rebuilt IN PLACE as
//     new slot s  <-  old slot perm[s]        (slots relative to the half: perm values are absolute, s0 = g0 * tile)
// for the first `limit` new slots only (limit = n_active[0] when gather_active, else all).
// The permutation is the same for every row.  A block of 16 waves walks row indices, kPermuteRows at a time: wave w fetches
// the old rows of groups w and w + 16 (prefetched one step ahead), parks them in the block's LDS tile, and -- after the
// barrier that also makes the in-place update safe: every old row of the step has been read before any new row is stored --
// builds the new rows of groups w and w + 16.  The picks of a lane's new dword are the same for every row index and live in
// registers: for each of its F labels {byte address in the tile, bit offset}, so one label costs ds_read_b32 + v_bfe_u32 +
// v_lshl_or_b32.  Labels without a source read a row of zeros.  The tile is double-buffered (one barrier per step).
// Memory-bound: (gold + gnew) rows of traffic per row index, each old row fetched once.  Two row arrays (message rows,
// channel rows) share one launch: same permutation, same picks.
// LDS: [2 buffers][kPermuteRows][kPermuteMaxGroups + 1][64] dwords = 66 KB (fixed strides: the offsets are immediates; two blocks
// per CU; above the 64 KB a launch gets by default: decoder.hip raises hipFuncAttributeMaxDynamicSharedMemorySize once).
constexpr int kPermuteMaxGroups = 32;
constexpr int kPermuteRows = 4;
constexpr int kPermuteRowStride = (kPermuteMaxGroups + 1) * 64;          // dwords per row index in the tile
constexpr int kPermuteLdsBytes = 2 * kPermuteRows * kPermuteRowStride * 4;
template <int PACK>
__global__ __launch_bounds__(1024) void permute_rows_kernel(uint8_t *__restrict__ rows_a, int n_rows_a, uint8_t *__restrict__ rows_b, int n_rows_b, int g0, int GH,
                                                            const int32_t *__restrict__ perm, const int32_t *__restrict__ ctl, int gather_active)
{
    constexpr int F = 4 * PACK, BITS = 8 / PACK, T = kRowBytes * PACK, R = kPermuteRows;
    extern __shared__ uint32_t tile[];                        // [2][R][kPermuteMaxGroups + 1][64]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (ctl && ctl[1]) return;                                 // the plan kernel found nothing to gain (block-uniform)
    const int s0 = g0 * T;
    // gather_active 1: only the active frames move (the first ctl[0] new slots; the rest of their rows is dropped)
    //               2: every frame of the groups that were live before this permutation moves -- the frames that just left keep
    //                  their (frozen) rows, now behind the active ones; frames that left earlier sit further back and stay put
    //               0: every slot of the half
    const int gold = (ctl && gather_active) ? ctl[3] : GH;     // groups that still held active frames before this permutation
    const int limit = gather_active == 1 ? ctl[0] : gather_active == 2 ? gold * T : GH * T;
    const int gnew = (limit + T - 1) / T;                      // groups that receive frames
    if (gnew == 0) return;

    // the picks of this lane's dwords of new groups w and w + 16
    uint32_t addr[2][F], sh[2][F];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int g = w + 16 * h;
#pragma unroll
        for (int j = 0; j < F; j++) {
            const int i = g * T + lane * F + j;                // new slot within the half
            int go = kPermuteMaxGroups, lo = lane, jo = 0;     // no source: the row of zeros
            if (g < gnew && i < limit) {
                const int o = perm[s0 + i] - s0;               // old slot within the half
                const int og = o / T, fo = o - og * T;
                if (og < gold) { go = og; lo = fo / F; jo = fo - lo * F; }
            }
            addr[h][j] = (uint32_t)((go * 64 + lo) * 4);
            // label jo of a lane sits at half jo / 4, byte jo % 4 (kernels_common.hpp): bit offset 8 * (jo % 4) + BITS * (jo / 4)
            sh[h][j] = (uint32_t)(8 * (jo & 3) + BITS * (jo >> 2));
        }
    }
    for (int i = threadIdx.x; i < 2 * R * 64; i += 1024)       // the rows of zeros
        tile[(i / 64) * kPermuteRowStride + kPermuteMaxGroups * 64 + (i & 63)] = 0u;

    const int n_rows = n_rows_a + n_rows_b;
    const bool load0 = w < gold, load1 = w + 16 < gold, own0 = w < gnew, own1 = w + 16 < gnew;      // wave-uniform
    auto row_ptr = [&](int r, int g) -> uint8_t * {            // row r of the two arrays taken as one, group g (r < n_rows)
        const bool in_a = r < n_rows_a;
        return (in_a ? rows_a + ((size_t)(g0 + g) * n_rows_a + r) * kRowBytes : rows_b + ((size_t)(g0 + g) * n_rows_b + (r - n_rows_a)) * kRowBytes) + lane * 4;
    };
    uint32_t pv[R][2];
    auto prefetch = [&](int r0) {
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int r = r0 + q;
            pv[q][0] = (load0 && r < n_rows) ? *reinterpret_cast<const uint32_t *>(row_ptr(r, w)) : 0u;
            pv[q][1] = (load1 && r < n_rows) ? *reinterpret_cast<const uint32_t *>(row_ptr(r, w + 16)) : 0u;
        }
    };
    auto step = [&](auto BUF, int r0, int r_next) {
        constexpr int buf = decltype(BUF)::value;
        uint32_t *tb = tile + buf * R * kPermuteRowStride;
#pragma unroll
        for (int q = 0; q < R; q++) {
            if (load0) tb[q * kPermuteRowStride + w * 64 + lane] = pv[q][0];
            if (load1) tb[q * kPermuteRowStride + (w + 16) * 64 + lane] = pv[q][1];
        }
        prefetch(r_next);                                      // in flight while this step's rows are assembled
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (!(h ? own1 : own0)) continue;
#pragma unroll
            for (int q = 0; q < R; q++) {
                if (r0 + q >= n_rows) continue;
                const uint8_t *tq = reinterpret_cast<const uint8_t *>(tb + q * kPermuteRowStride);
                uint32_t out = 0;
#pragma unroll
                for (int j = 0; j < F; j++)
                    out = lshl_or(__builtin_amdgcn_ubfe(*reinterpret_cast<const uint32_t *>(tq + addr[h][j]), sh[h][j], (uint32_t)BITS),
                                  (uint32_t)(8 * (j & 3) + BITS * (j >> 2)), out);
                *reinterpret_cast<uint32_t *>(row_ptr(r0 + q, w + 16 * h)) = out;
            }
        }
    };
    // a fixed, small grid walks the rows (an empty check point must cost microseconds, not one block per row)
    const int stride = (int)gridDim.x * R;
    int r0 = (int)blockIdx.x * R;
    prefetch(r0);
    __syncthreads();                                           // the rows of zeros are in place
    while (r0 < n_rows) {                                      // block-uniform
        step(std::integral_constant<int, 0>{}, r0, r0 + stride);
        r0 += stride;
        if (r0 >= n_rows) break;
        step(std::integral_constant<int, 1>{}, r0, r0 + stride);
        r0 += stride;
    }
}

}  // namespace lutldpc
