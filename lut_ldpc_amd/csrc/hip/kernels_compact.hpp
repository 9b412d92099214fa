// kernels_compact.hpp -- compaction of the surviving frames (as-shipped mode, skewed pipeline).
//
// Early termination freezes a finished frame in place; its lane keeps riding along until the whole
// 512-frame group is done, which rarely happens while a few stragglers need 40+ iterations.  Every few
// iterations the frame SLOTS of a half are therefore permuted -- active frames to the front (stable), all
// others behind them -- so that whole groups fall idle and their waves exit on the first look at the state
// words (load_active).  Nothing is lost: the decided bits, iteration codes, states and pending flags move
// with their frame, `frame_of[slot]` remembers where each frame came from, and at the end of the decode the
// decided bits and iteration codes are put back into the original order.  Only the message and channel rows
// of the ACTIVE frames are moved (finished frames never read theirs again).
//
// One permutation = compact_apply_kernel (one block per half: prefix sum, small per-slot arrays) +
// permute_rows_kernel over the E message rows, the N channel rows and the N decided-bit rows.
//
// Whether a permutation pays is decided ON THE DEVICE at every check point (the launch sequence is a fixed hipGraph): it
// moves  (live + new) groups of message / channel rows  and all decided-bit rows -- measured 1.4 ms for half a 16384-frame
// DVB-S2 batch, about  0.7 (live + new) + 0.3 GH  group-iterations -- and saves  (live - new) groups x remaining
// iterations.  compact_decide_kernel permutes only when a sizeable share of the live groups falls idle at once and the
// saving exceeds the cost; otherwise the other kernels of the check point return at once (5 us each).
//
// A check point = compact_decide_kernel -> hard_from_frozen_kernel (the decided bits of the frames that left since the
// last permutation are read off their frozen messages BEFORE those are dropped, kernels_generic.hpp) ->
// compact_apply_kernel -> permute_rows_kernel x 3.
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
                                                               int iters_left, float margin, float min_share)
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
        const float cost = 0.7f * (float)(live + gnew) + 0.3f * (float)gh;
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

// rows[g][r][256 B], groups g0 .. g0+GH-1 (one half): row r of every group is rebuilt as
//     new slot s  <-  old slot perm[s]        (slots relative to the half: perm values are absolute, s0 = g0 * tile)
// for the first `limit` new slots only (limit = n_active[0] when gather_active, else all).
// The permutation is the same for every row, so a block first turns it into DESCRIPTORS in LDS -- for every label of
// every new dword {dword index in the wave's tile of old rows, bit offset}, 16 bits -- and then streams rows: a wave loads
// the `gold` old dwords of its lane (eight loads in flight), parks them in its LDS tile, and assembles every new dword
// from F picks {LDS read, shift, mask, merge}.  Memory-bound: (gold + gnew) rows of traffic per row index.
// LDS: [4 waves][GH][64] dwords + [gnew][64][F] descriptors (GH <= 32: dword index < 2048 = 11 bits, bit offset 5 bits).
constexpr int kPermuteMaxGroups = 32;
template <int PACK>
__global__ __launch_bounds__(256) void permute_rows_kernel(uint8_t *__restrict__ rows, int n_rows, int rows_per_group, int g0, int GH,
                                                           const int32_t *__restrict__ perm, const int32_t *__restrict__ ctl, int gather_active)
{
    constexpr int F = 4 * PACK, BITS = 8 / PACK, T = kRowBytes * PACK;
    extern __shared__ uint32_t tile[];                        // [4 waves][GH][64] | descriptors
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (ctl && ctl[1]) return;                                 // the plan kernel found nothing to gain (block-uniform)
    uint32_t *my = tile + (size_t)w * GH * 64;
    uint16_t *desc = reinterpret_cast<uint16_t *>(tile + (size_t)4 * GH * 64);
    const int s0 = g0 * T;
    const int limit = gather_active ? ctl[0] : GH * T;
    const int gnew = (limit + T - 1) / T;                      // groups that receive frames
    const int gold = (ctl && gather_active) ? ctl[3] : GH;     // groups that still held active frames before this permutation
    if (gnew == 0) return;
    for (int i = threadIdx.x; i < gnew * T; i += 256) {        // new slot i of the half = (group i / T, lane (i % T) / F, label i % F)
        const int o = perm[s0 + i] - s0;                       // old slot within the half
        const int go = o / T, fo = o - go * T, lo = fo / F, jo = fo - lo * F;
        // label jo of a lane sits at half jo / 4, byte jo % 4 (kernels_common.hpp): bit offset 8 * (jo % 4) + BITS * (jo / 4)
        desc[i] = go < gold ? (uint16_t)((go * 64 + lo) | ((8 * (jo & 3) + BITS * (jo >> 2)) << 11)) : (uint16_t)0xFFFFu;
    }
    __syncthreads();
    // a fixed, small grid walks the rows (an empty check point must cost microseconds, not one block per row)
    for (int r = blockIdx.x * 4 + w; r < n_rows; r += gridDim.x * 4) {
        for (int gb = 0; gb < gold; gb += 8) {                 // eight row loads in flight per lane
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; k++)
                v[k] = gb + k < gold ? *reinterpret_cast<const uint32_t *>(rows + ((size_t)(g0 + gb + k) * rows_per_group + r) * kRowBytes + lane * 4) : 0u;
#pragma unroll
            for (int k = 0; k < 8; k++) if (gb + k < gold) my[(gb + k) * 64 + lane] = v[k];
        }
        // (wave-private tile: no barrier needed, the LDS queue is in order within a wave)
        for (int g = 0; g < gnew; g++) {
            uint32_t out = 0;
            const uint16_t *dd = desc + ((size_t)g * 64 + lane) * F;
#pragma unroll
            for (int j = 0; j < F; j++) {
                const uint32_t de = dd[j];
                const uint32_t v = de != 0xFFFFu ? (my[de & 0x7FFu] >> (de >> 11)) & ((1u << BITS) - 1u) : 0u;
                out |= v << (8 * (j & 3) + BITS * (j >> 2));
            }
            *reinterpret_cast<uint32_t *>(rows + ((size_t)(g0 + g) * rows_per_group + r) * kRowBytes + lane * 4) = out;
        }
    }
}

}  // namespace lutldpc
