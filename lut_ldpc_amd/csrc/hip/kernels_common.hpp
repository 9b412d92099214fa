// kernels_common.hpp -- device-side conventions shared by all gfx950 kernels.
//
// Data layout in HBM ("frame tiles"): a batch of B frames is cut into groups of 256*PACK frames.
// Every per-edge / per-node quantity is stored as one 256-byte row per (group, edge|node):
//      msgs[g][e][256 B]  e = VN-major edge id of the reference (src/LDPC_Code_LUT.cpp:513-521)
//      cha [g][v][256 B]  channel labels,   hard[g][v][256 B] decided bits
// PACK = 1: one byte per (row, frame)   -- any alphabet up to 128 labels, b = 1 byte per message
// PACK = 2: one NIBBLE per (row, frame) -- alphabets up to 16 labels (the 3- and 4-bit decoders the
//           toolkit is about), b = 0.5 byte per message: half the HBM traffic of both passes.
// One wavefront (64 lanes) owns one row: lane L holds frames 4*PACK*L .. 4*PACK*L + 4*PACK-1 of the
// group in ONE dword.  With PACK = 2 the dword is split into two "halves" of four frames each,
//      half 0 = low nibbles  of bytes 0..3  -> frames 8L+0..3
//      half 1 = high nibbles of bytes 0..3  -> frames 8L+4..7
// so that `x & 0x0F0F0F0F` / `(x >> 4) & 0x0F0F0F0F` yield the same "four frames, one per byte"
// format as PACK = 1 and all byte-parallel (SWAR) arithmetic is shared.
// Every global access of a wave is a single fully coalesced 256-byte row and all node / edge
// indices are wave-uniform (scalar registers, scalar loads).
// Per-frame state lives in byte arrays indexed by frame; a lane's frames are PACK dwords of them.
#pragma once
#ifndef __HIPCC_RTC__      // hiprtc (jit.hpp compiles this header at run time) has the HIP runtime and the fixed-width types built in
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
typedef unsigned long long uint64_t;
typedef short int16_t;
typedef int int32_t;
typedef long long int64_t;
#endif

namespace lutldpc {

constexpr int kRowBytes = 256;       // bytes per row (= 64 lanes x one dword)
constexpr int kTileFrames = kRowBytes;   // frames per group when PACK = 1 (kept for the host code)
constexpr int kWave = 64;
constexpr int kMaxSeg = 32;          // degree classes handled by one pass launch

// frame states
constexpr uint32_t ST_ACTIVE = 0, ST_DONE_PSC = 1, ST_DONE_PISC = 2, ST_PAD = 3;
constexpr uint32_t ST_DONE_SAVED = 4;   // left through the exit test, decided bits already recovered into the hard rows (compaction)

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
    int32_t vfail_stride_w;   // words between two copies of the early-termination flags (see flag_frames)
    PassSeg seg[kMaxSeg];
};

// ---- packing ------------------------------------------------------------------------------------
// half h of a row dword as four frames, one per byte
template <int PACK>
__device__ __forceinline__ uint32_t unpack_half(uint32_t x, int h) {
    if constexpr (PACK == 1) return x;
    else return h ? ((x >> 4) & 0x0F0F0F0Fu) : (x & 0x0F0F0F0Fu);
}
template <int PACK>
__device__ __forceinline__ uint32_t pack_halves(const uint32_t (&r)[PACK]) {
    if constexpr (PACK == 1) return r[0];
    else return r[0] | (r[1] << 4);
}
// byte mask (0xFF / 0x00 per frame) of the two halves -> mask over the packed dword
template <int PACK>
__device__ __forceinline__ uint32_t pack_masks(const uint32_t (&m)[PACK]) {
    if constexpr (PACK == 1) return m[0];
    else return (m[0] & 0x0F0F0F0Fu) | (m[1] & 0xF0F0F0F0u);
}
// dword index of half h of lane `lane` in group g inside a per-frame byte array
template <int PACK>
__device__ __forceinline__ int frame_word(int g, int lane, int h) { return (g * kWave + lane) * PACK + h; }

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
// (a & m) | (b & ~m)
// (one v_bitop3_b32 / v_bfi_b32; written out, the compiler hoists ~m of a loop-invariant mask and then needs two instructions per use)
__device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(m, a, b, (0xCC & 0xF0) | (0xAA & 0x0F)); }

// Row access through a buffer resource: the descriptor (4 SGPRs) covers the rows of one frame group,
// the row offset is wave-uniform (SGPR soffset) and the lane supplies only its byte offset within the
// row (one VGPR): no 64-bit address arithmetic in vector registers, and an out-of-range row reads 0 /
// drops the store instead of faulting.  One group's rows must stay below 4 GiB (checked at create).
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
// Cache policy of the row traffic (the aux immediate of the buffer instructions on gfx950: 1 = sc0, 2 = nt, 16 = sc1).
// Every row is read once and written once per pass and the batch is far larger than L2 + Infinity Cache, so there is no
// reuse to protect: non-temporal loads AND stores.  Measured with tools/ab_variants.sh (DVB-S2, 16384 frames, six
// interleaved processes per variant on one box, profiles/r02_ab_cache_policy.txt): nt/nt +3.7 % over the default
// policy, nt loads alone +0.3 %, nt stores alone +0.5 %, sc1 (write-through) stores +2.8 %.
#ifndef LUTLDPC_LD_AUX
#define LUTLDPC_LD_AUX 2
#endif
#ifndef LUTLDPC_ST_AUX
#define LUTLDPC_ST_AUX 2
#endif
__device__ __forceinline__ uint32_t ld_row(rsrc_t r, uint32_t row_off, uint32_t lane4) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)lane4, (int)row_off, LUTLDPC_LD_AUX);
}
__device__ __forceinline__ void st_row(rsrc_t r, uint32_t row_off, uint32_t lane4, uint32_t v) {
    __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)lane4, (int)row_off, LUTLDPC_ST_AUX);
}

// Entry of a software-pipelined loop whose first step was peeled: drain the vector-memory queue
// (s_waitcnt vmcnt(0), expcnt / lgkmcnt untouched) with scheduling barriers on both sides.  The
// compiler places its waits per basic block from the merged state of all predecessors; entering the
// loop with nothing outstanding makes the entry edge agree with the back edge, so that inside the loop
// the only wait is "prefetched loads done, the stores issued after them still in flight".
__device__ __forceinline__ void pipeline_entry_fence() {
    asm volatile("" ::: "memory");          // IR level: no load may sink below the fence
    __builtin_amdgcn_sched_barrier(0);      // machine scheduler: nothing moves across
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ bool wave_all_zero(uint32_t x) { return __ballot(x != 0) == 0ull; }

// per-half "still decoding" byte masks of a lane; returns true when the whole wave is finished
template <int PACK>
__device__ __forceinline__ bool load_active(const uint32_t *__restrict__ state_w, int g, int lane, uint32_t (&amask)[PACK]) {
    uint32_t any = 0;
#pragma unroll
    for (int h = 0; h < PACK; h++) { amask[h] = swar_zero_mask(state_w[frame_word<PACK>(g, lane, h)]); any |= amask[h]; }
    return wave_all_zero(any);
}

// store `val` into a row dword, keeping the old content for frames that already terminated
template <int PACK>
__device__ __forceinline__ void store_blend(uint32_t *p, uint32_t val, uint32_t old, uint32_t smask) {
    *p = bfi(smask, val, old);
}
template <int PACK>
__device__ __forceinline__ void store_row_masked(uint32_t *p, uint32_t val, uint32_t smask) {
    if (smask == 0xFFFFFFFFu) *p = val;
    else if (smask) *p = bfi(smask, val, *p);
}

// Early-termination flags.  A failing frame fails in thousands of waves of one launch and all waves of a
// frame group own the same frames, so a single flag word per four frames would serialise ~10^4 atomics
// per launch on a handful of L2 lines (measured: +60 % on the whole pass).  The flags are therefore
// kept in kVfailSlots copies (slot = block index mod kVfailSlots, `stride_w` words apart); the atomics
// are fire-and-forget (no return value) and frame_state_kernel ORs the copies together.
constexpr int kVfailSlots = 32;
template <int PACK>
__device__ __forceinline__ void flag_frames(uint32_t *__restrict__ vfail_w, int stride_w, int g, int lane, const uint32_t (&fail)[PACK], const uint32_t (&amask)[PACK]) {
    uint32_t *slot = vfail_w + (size_t)(blockIdx.x & (kVfailSlots - 1)) * (size_t)stride_w;
#pragma unroll
    for (int h = 0; h < PACK; h++) {
        const uint32_t f = fail[h] & amask[h] & 0x01010101u;
        if (f) atomicOr(&slot[frame_word<PACK>(g, lane, h)], f);
    }
}

// ---- parameters of the specialised kernels (kernels_fast.hpp and the run-time generated ones of jit.hpp)
constexpr int kFastMaxTables = 32;     // LUT nodes of one balanced tree (degree <= 33)
constexpr int kFastTableStride = 256;  // bytes per table slot in LDS

struct FastParams {
    int32_t n_nodes, node_off, nodes_per_wave, waves_per_group;
    int32_t idx_off;       // offset of the class in the dense index blob (see decoder.hip: build_fast_index)
    int32_t G, E, N;
    int32_t g0;            // first frame group of the launch (the G groups g0 .. g0+G-1 are processed)
    int32_t nz;            // sign threshold (see PassParams)
    int32_t shift_msg;     // log2 of the message alphabet feeding the tables (label = a | b << shift)
    int32_t check, write_hard;
    int32_t deg;
    int32_t n_tables;
    int32_t tab_off[kFastMaxTables];    // byte offsets into the table blob, canonical node order
    int32_t tab_len[kFastMaxTables];
    int32_t tab_shift[kFastMaxTables];  // log2 alphabet of each table's first child
    int32_t nib;                        // (unused: nibble-packed LDS tables were measured slower and removed)
    int32_t vfail_stride_w;             // words between two copies of the early-termination flags (see flag_frames)
    int32_t vfail_off_w;                // word offset of the flag buffer this pass reports to (skewed pipeline: two buffers)
};
// chain fusion of the check pass (cn_minsum_body<..., CHAIN>): which table, which links
struct ChainParams {
    int32_t on;            // 1: update the linked degree-2 variable nodes inside this check pass
    int32_t idx_off;       // dense [n_nodes][2] = {back node + 1, forward node + 1} (0 = none)
    int32_t tab_off, tab_len, tab_shift;    // the degree-2 class' root table: label = message | channel << tab_shift
    // early termination (parity_check_iter): the unanimity test of the nodes updated here belongs to the NEXT exit test
    // (other flag buffer); their decided bits -- the signs of the messages they sent LAST iteration -- are stored by the
    // check pass that reads those messages (hard = 1), because the variable pass skipped them
    int32_t check, hard, vfail_off_w, sbit_out;
};

// (x << s) | y in one instruction, s wave-uniform
__device__ __forceinline__ uint32_t lshl_or(uint32_t x, int s, uint32_t y) {
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(s), "v"(y));
    return r;
}

// (a ^ b) + c in one instruction (the compiler splits it whenever a ^ b has a second use)
__device__ __forceinline__ uint32_t xad(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
// three-input bit operations of gfx950 (v_bitop3_b32; truth tables from a = 0xF0, b = 0xCC, c = 0xAA)
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xF0 ^ 0xCC ^ 0xAA); }
__device__ __forceinline__ uint32_t xor_or(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, (0xF0 ^ 0xCC) | 0xAA); }   // (a ^ b) | c

#ifndef __HIPCC_RTC__
// Every kernel of the library is launched through this wrapper: its argument segment stays within 128 bytes -- pointers and a few
// integers; anything larger (role lists, class parameters, channel cells) lives in DEVICE memory and is passed by pointer.  The one
// device fault this library has shown pointed at the stale tail of a 3472-byte by-value argument segment;
// the static_assert keeps such a segment from coming back.
constexpr size_t kMaxKernelArgBytes = 128;
template <typename... KArgs, typename... Args>
inline void launch_k(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t stream, Args &&...args) {
    static_assert((((sizeof(KArgs) + 7) / 8 * 8) + ... + 0) <= kMaxKernelArgBytes, "kernel argument segment above 128 bytes: pass the structure through device memory");
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, static_cast<KArgs>(args)...);
}
#endif

// locate the degree class of a block: linear scan over <= kMaxSeg scalar entries
__device__ __forceinline__ int find_seg(const PassParams &P, int b) {
    int s = 0;
#pragma unroll 1
    for (int i = 1; i < P.n_seg; i++) if (b >= P.seg[i].block_begin) s = i;
    return s;
}

}  // namespace lutldpc
