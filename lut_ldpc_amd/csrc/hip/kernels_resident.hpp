// kernels_resident.hpp -- device helpers of the LDS-RESIDENT decoder (jit_resident.hpp generates the kernel around them).
//
// A code whose edge messages fit the 160 KB of LDS of one compute unit -- E * 4 bytes per SET of 8 frames (nibble labels) --
// is decoded without touching HBM between the channel labels and the decided bits: one workgroup owns S sets, keeps their E
// edge messages in LDS as dwords (eight nibble frames or four byte frames per dword: the same per-lane format as one lane of
// a 256-byte row of the streaming kernels, kernels_common.hpp), and runs ALL iterations (src/LDPC_Code_LUT.cpp:301-338) with
// workgroup barriers between the passes.  A thread handles one (set, node) item per round: the arithmetic of an item is the
// arithmetic one LANE of the streaming kernels does (same SWAR min-sum, same frame-by-frame tree look-ups), only the
// operands come from LDS instead of rows in HBM and the node indices are per lane instead of wave-uniform.
// This header is embedded into the library as text and prepended to the generated source (like kernels_common.hpp).
#pragma once
#include "kernels_common.hpp"

namespace lutldpc {

// Min-sum check update on the packed labels of one check (src/LDPC_Code_LUT.cpp:355-402): x[k] = dword of edge k (8 nibble
// frames / 4 byte frames), r[k] = extrinsic output.  Same arithmetic as cn_minsum_body (kernels_fast.hpp): the extrinsic
// magnitude `mag == min1 ? min2 : min1` is the minimum over the OTHER edges, and all DEG of them come out of suffix minima,
// a running prefix minimum and one combination per inner edge -- 3 (DEG - 2) two-input minima -- on complemented magnitudes
// (minima become maxima, the sign step stays one three-input bit operation); sign bit `sbit` is the flag bit of every comparison.
// Returns the parity of the negative inputs in bit `sbit` of every element (the check's syndrome bit per frame).
template <int DEG, int PACK>
__device__ __forceinline__ uint32_t res_minsum(const uint32_t (&x)[DEG], uint32_t (&r)[DEG], int sbit, uint32_t SB, uint32_t LOW) {
    const uint32_t odd = (DEG & 1) ? SB : 0u;
    // max of two complemented magnitudes: (b ^ LOW) + a carries into bit sbit <=> a > b
    auto mx = [&](uint32_t a, uint32_t b) -> uint32_t {
        const uint32_t gt = xad(b, LOW, a) & SB;
        return bfi(gt - (gt >> sbit), a, b);
    };
    uint32_t mcs[DEG], spp = 0;
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const uint32_t xh = x[k];
        const uint32_t pos = xh & SB;
        mcs[k] = (xh ^ (pos - (pos >> sbit))) & LOW;              // positive: LOW - magnitude code; negative: the code itself
        if (k & 1) spp = xor3(spp, x[k - 1], xh);
        else if (k == DEG - 1) spp ^= xh;
    }
    const uint32_t tn = (spp ^ odd) & SB;
    uint32_t oc[DEG];
    if constexpr (DEG == 1) oc[0] = 0u;                           // (no other edge: magnitude nz-1, as min1 = LOW gave before)
    else if constexpr (DEG == 2) { oc[0] = mcs[1]; oc[1] = mcs[0]; }
    else {
        uint32_t suf[DEG];
        suf[DEG - 1] = mcs[DEG - 1];
#pragma unroll
        for (int k = DEG - 2; k >= 1; k--) suf[k] = mx(mcs[k], suf[k + 1]);
        uint32_t pre = mcs[0];
        oc[0] = suf[1];
#pragma unroll
        for (int k = 1; k <= DEG - 2; k++) { oc[k] = mx(pre, suf[k + 1]); pre = mx(pre, mcs[k]); }
        oc[DEG - 1] = pre;
    }
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const uint32_t po = (tn ^ x[k]) & SB;
        const uint32_t kp = po - (po >> sbit);
        r[k] = xor_or(oc[k], kp, po);
    }
    return tn;
}

// 0x1 in every element (nibble / byte) of a packed label dword whose value is < t (any t up to the alphabet size): the
// decided bit of a label, src/LDPC_Code_LUT.cpp:275
template <int PACK>
__device__ __forceinline__ uint32_t res_lt(uint32_t x, uint32_t t) {
    uint32_t r[PACK];
#pragma unroll
    for (int h = 0; h < PACK; h++) r[h] = swar_lt(unpack_half<PACK>(x, h), t);
    return pack_halves<PACK>(r);
}

// Raise the failed-frame flags `f` (one bit per element) of set `s` in the workgroup's flag words.  In the first iterations nearly
// every item has a failing frame, and the lanes of a wave mostly work on the same set: 64 lanes doing an LDS atomic on ONE word
// serialise (64 LDS cycles per wave-instruction, a third of a degree-3 item's whole LDS time).  So the wave first ORs its flags
// together -- one ballot per frame position -- and a single lane posts the result; a wave that straddles two sets falls back to
// per-lane atomics.
template <int PACK>
__device__ __forceinline__ void res_flag(uint32_t *fail_words, int s, uint32_t f) {
    const unsigned long long any = __ballot(f != 0u);
    if (!any) return;
    const int first = (int)__builtin_ctzll(any);               // the first lane with a flag to raise (wave-uniform)
    const int s0 = __builtin_amdgcn_readlane(s, first);
    if (__ballot(f != 0u && s != s0) == 0ull) {
        constexpr int BITS = 8 / PACK, F = 4 * PACK;
        uint32_t r = 0u;
#pragma unroll
        for (int n = 0; n < F; n++) if (__ballot((f >> (n * BITS)) & 1u)) r |= 1u << (n * BITS);
        if ((int)(threadIdx.x & 63) == first) atomicOr(&fail_words[s0], r);
    } else if (f) atomicOr(&fail_words[s], f);
}

// element mask (0xF / 0xFF per frame) from one flag bit per element
template <int PACK>
__device__ __forceinline__ uint32_t res_mask(uint32_t one_bits) { return one_bits * (PACK == 2 ? 0xFu : 0xFFu); }

// frame offset inside its set of element n of a dword (kernels_common.hpp: low nibbles of bytes 0..3 = frames 0..3, high
// nibbles = frames 4..7; byte rows: byte b = frame b)
template <int PACK>
__device__ __forceinline__ int res_frame_of_element(int n) { return PACK == 2 ? ((n & 1) * 4 + (n >> 1)) : n; }

}  // namespace lutldpc
