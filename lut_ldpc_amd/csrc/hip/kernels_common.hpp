// kernels_common.hpp -- device-side conventions shared by all gfx950 kernels.
//
// Data layout in HBM ("frame tiles"): a batch of B frames is cut into G = ceil(B/256) groups
// of 256 frames.  Every per-edge / per-node quantity is stored as one 256-byte row per
// (group, edge|node):
//      msgs[g][e][f]   uint8, e = VN-major edge id of the reference (src/LDPC_Code_LUT.cpp:513-521)
//      cha [g][v][f]   uint8 channel labels,   hard[g][v][f] uint8 decided bits
// One wavefront (64 lanes) owns one row segment: lane L holds frames 4L..4L+3 of the group
// packed in one dword, so every global access of a wave is a single fully coalesced 256-byte
// row and all node/edge indices are wave-uniform (scalar registers, scalar loads).
// Per-frame state lives in byte arrays indexed [g*256 + f] and is read as the same dwords.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lutldpc {

constexpr int kTileFrames = 256;     // frames per group (= 64 lanes x 4 packed bytes)
constexpr int kWave = 64;
constexpr int kMaxSeg = 32;          // degree classes handled by one pass launch

// frame states
constexpr uint32_t ST_ACTIVE = 0, ST_DONE_PSC = 1, ST_DONE_PISC = 2, ST_PAD = 3;

struct PassSeg {
    int32_t block_begin;   // first block (within one frame group) of this degree class
    int32_t n_nodes;       // nodes in the class
    int32_t node_off;      // offset of the class in the node list
    int32_t deg;           // node degree
    int32_t op_off;        // first op of the class program (generic kernels)
    int32_t n_ops;
    int32_t tab_off;       // byte offset of the class tables in the table blob
    int32_t tab_bytes;
    int32_t n_in, n_out, n_slots;
    int32_t fast;          // specialised kernel id (0 = generic)
};

struct PassParams {
    int32_t n_seg;
    int32_t blocks_per_group;
    int32_t nodes_per_block;
    int32_t G;
    int32_t E;             // edges  (rows of msgs per group)
    int32_t N;             // variable nodes (rows of cha/hard per group)
    int32_t nz;            // sign threshold of the messages this pass WRITES (VN) / READS (CN)
    int32_t check;         // 1: accumulate the early-termination test into vfail
    int32_t write_hard;    // VN pass: store the sign of the new messages as hard decisions
    int32_t slots_lds;     // dwords per lane reserved for program slots
    PassSeg seg[kMaxSeg];
};

// ---- SWAR helpers on four packed bytes (all byte values < 128) --------------------------------
// 0x01 in every byte whose value is <  t   (t wave-uniform, 0 <= t <= 128)
__device__ __forceinline__ uint32_t swar_lt(uint32_t x, uint32_t t) {
    uint32_t d = (x | 0x80808080u) - t * 0x01010101u;   // byte = 0x80 + x - t, no borrow
    return (~d >> 7) & 0x01010101u;
}
// 0xFF in every byte that is zero
__device__ __forceinline__ uint32_t swar_zero_mask(uint32_t x) {
    uint32_t nz = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;     // bit7 set iff byte != 0
    uint32_t z = (~nz >> 7) & 0x01010101u;
    return z * 0xFFu;
}
// expand 0x01 flags to 0xFF masks
__device__ __forceinline__ uint32_t swar_flag_to_mask(uint32_t f) { return f * 0xFFu; }
// (a & m) | (b & ~m)
__device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }

__device__ __forceinline__ bool wave_all_zero(uint32_t x) { return __ballot(x != 0) == 0ull; }

// locate the degree class of a block: linear scan over <= kMaxSeg scalar entries
__device__ __forceinline__ int find_seg(const PassParams &P, int b) {
    int s = 0;
#pragma unroll 1
    for (int i = 1; i < P.n_seg; i++) if (b >= P.seg[i].block_begin) s = i;
    return s;
}

}  // namespace lutldpc
