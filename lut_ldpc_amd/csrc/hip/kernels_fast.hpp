// kernels_fast.hpp -- specialised gfx950 kernels for the shapes ber_sim actually produces:
//   * min-sum check nodes (src/LDPC_Code_LUT.cpp:355-402): exact-degree straight-line code, the check's
//     rows in registers, SWAR arithmetic on the eight nibble frames (or four byte frames) of each dword;
//   * variable / decision nodes whose tree is the balanced binary tree of
//     LUT_Tree_Node::gen_bin_balanced_tree (src/LUT_Tree.cpp:200-237) -- the only shape
//     ber_sim designs in its auto modes (src/LDPC_BER_Sim.cpp:487-489) -- expanded at compile time with
//     every shared sub-expression computed once and evaluated frame by frame on unpacked labels;
//   * pass_fused_kernel: the check pass of one half of the frame groups and the variable pass of the
//     other half in one launch (the skewed two-half pipeline of decoder.hip).
// Other tree shapes get kernels generated at run time (jit.hpp); kernels_generic.hpp is the fallback.
//
// Roofline: HBM-bound streaming of 256-byte rows.  Algorithmic bytes per launch with b bytes per label:
// check pass 2*E*b*B, variable pass (2*E + N)*b*B (+N*b*B when the hard decisions are written for the
// early-termination test); DESIGN.md section 3 has the measured on-chip limits (VALU issue, LDS banks).
#pragma once
#include "kernels_common.hpp"
#include "lut_program.hpp"

#include <utility>

namespace lutldpc {

// ------------------------------------------------------------------------------------------
// Min-sum check pass.  DMAX: register budget (rows kept per check); the true degree is the
// wave-uniform runtime value P.deg <= DMAX.  One wave = one 256-byte row; UNR checks are in flight.
//
// Byte-parallel arithmetic on four frames per register, full-rate integer ops only.  With
// nz = Nq/2 = 1 << sbit a label is  [sign bit `sbit` (1 = positive LLR)] [magnitude code in the bits
// below]: positive -> magnitude = label & LOW, negative -> magnitude = ~label & LOW (this IS the
// reference's label-nz / nz-1-label, src/LDPC_Code_LUT.cpp:368-374).  Magnitudes never use bit
// `sbit`, so it serves as the per-byte flag bit of every comparison:
//      (a | SB) - b  has bit sbit set  <=>  a >= b           (no borrow between bytes)
//      flag - (flag >> sbit)           =    LOW in flagged bytes (a select mask for magnitudes)
// min1/min2 start at LOW (= nz-1, the largest magnitude) instead of the reference's nz: identical for
// every check of degree >= 2 (both are replaced by real magnitudes after two inputs).
// DEG is the exact check degree (straight-line code, all row loads of UNR checks issued up front);
// `edges` is the dense [n_nodes][DEG] table of edge ids of this degree class, read with scalar loads.
// `block` = index of the 4-wave block within this degree class; PT = FastParams or RoleParams.
// CHAIN (UNR = 1): a degree-2 variable node shared by two consecutive checks of this wave is updated here -- its two
// incoming messages are this check's output r[0] and the previous check's output r[1], held back in `pend` -- and
// the NEW variable-to-check messages are stored instead of the check-to-variable ones: one write and one read less per
// such edge and iteration (decoder.hip: build_fast_index puts the chain edges at slots 0 / 1 of the edge table).
// FIRST (the check pass of iteration 0 in the fused pipeline): every edge still carries its variable node's initial message
// (src/LDPC_Code_LUT.cpp:284-289), so the inputs are read from the N initial-message rows through a second table holding the
// NODE of every check edge -- the E edge rows are written for the first time by this pass, no separate copy kernel.
constexpr int kCnAllButOneMaxDeg = 16;     // checks up to this degree: all-but-one minima from prefix / suffix minima (registers: DEG - 2 suffixes)
// (ABO: the largest degree that takes the all-but-one form in this instantiation -- the fused kernels of the lean degree buckets keep
// their few wide checks on the running-minima form, whose registers do not grow with the degree)
template <int DEG, int UNR, int PACK, bool CHAIN, typename PT, bool FIRST = false, int ABO = kCnAllButOneMaxDeg>
__device__ __forceinline__ void cn_minsum_body(
    const PT &P, int block, uint8_t *msgs, const uint32_t *__restrict__ state_w, uint32_t *__restrict__ vfail_w,
    const int32_t *__restrict__ fast_idx, const ChainParams CH = ChainParams{}, uint8_t *lds_tab = nullptr,
    const uint8_t *cha = nullptr, const uint8_t *__restrict__ tables = nullptr, uint8_t *hard = nullptr, const uint8_t *msg0 = nullptr)
{
    static_assert(!CHAIN || (UNR == 1 && DEG >= 2), "chain fusion: one check per step");
    if constexpr (CHAIN) {      // the degree-2 root table (block-uniform call)
        const int i = threadIdx.x;
        if (CH.on && i < CH.tab_len / 4) reinterpret_cast<uint32_t *>(lds_tab)[i] = reinterpret_cast<const uint32_t *>(tables + CH.tab_off)[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(block * 4 + (threadIdx.x >> 6));    // wave-uniform -> SGPR
    const int gl = wave / P.waves_per_group;
    if (gl >= P.G) return;
    const int chunk = wave - gl * P.waves_per_group;
    const int g = gl + P.g0;
    uint32_t amask[PACK];
    if (load_active<PACK>(state_w, g, lane, amask)) return;
    const uint32_t smask = pack_masks<PACK>(amask);
    const bool all_active = __ballot(smask != 0xFFFFFFFFu) == 0ull;      // wave-uniform: no frozen frame, plain stores
    const rsrc_t base = make_rsrc(msgs + (size_t)g * (size_t)P.E * kRowBytes, (uint32_t)P.E * kRowBytes);   // this group's edge rows
    const uint32_t lane4 = (uint32_t)lane * 4u;
    const int32_t *edges = fast_idx + P.idx_off;
    const int32_t *nodes = edges;
    rsrc_t nbase = base;
    if constexpr (FIRST) {
        nodes = fast_idx + P.nidx_off;
        nbase = make_rsrc(msg0 + (size_t)g * (size_t)P.N * kRowBytes, (uint32_t)P.N * kRowBytes);       // this group's initial-message rows
    }
    const int first = chunk * P.nodes_per_wave;
    int last = first + P.nodes_per_wave;
    if (last > P.n_nodes) last = P.n_nodes;
    // SWAR element = one byte (PACK = 1) or one NIBBLE (PACK = 2): with nibble rows the arithmetic runs
    // on all eight frames of the dword at once, no unpacking (nz <= 8, so sign bit + magnitude fit 4 bits)
    constexpr uint32_t ONE = PACK == 2 ? 0x11111111u : 0x01010101u;
    const int sbit = __builtin_ctz((unsigned)P.nz);
    const uint32_t SB = (uint32_t)P.nz * ONE, LOW = SB - ONE;
    const uint32_t odd = (DEG & 1) ? SB : 0u;
    uint32_t failw = 0;

    // Software pipeline.  UNR checks are evaluated per step and the rows of the NEXT step are requested
    // before the current one is evaluated, so a wave keeps UNR*DEG loads in flight while it computes (the
    // pass needs ~150 rows in flight per CU to cover the HBM latency).  Structure (first step peeled)
    //     fetch(0) fetch(1) eval(0) { x <- next; fetch(i+2); eval(i+1) }
    // gives both edges into the loop the same queue of outstanding accesses [loads, stores], which lets the
    // compiler wait with the exact vmcnt (loads done, the stores behind them still in flight).  A fetch
    // past the end of the chunk uses an out-of-range lane offset: the buffer unit returns 0 without
    // touching memory.
    // chain state: links of the check being evaluated / fetched, the held-back output of the previous check
    const int32_t *links = CHAIN ? fast_idx + CH.idx_off : nullptr;
    const rsrc_t cbase = CHAIN ? make_rsrc(cha + (size_t)g * (size_t)P.N * kRowBytes, (uint32_t)P.N * kRowBytes) : base;
    int lb = 0, lf = 0, lbn = 0, lfn = 0;          // back / forward node + 1 of the current and of the next check
    uint32_t xc = 0, xcn = 0;                      // channel row of the back node
    uint32_t pend = 0, pend_old = 0, chainfail = 0;
    int pend_e = 0;
    auto fetch = [&](int i, uint32_t (&xx)[UNR][DEG], int (&ee)[UNR][DEG]) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int ii = (i + u < last) ? i + u : last - 1;
#pragma unroll
            for (int k = 0; k < DEG; k++) ee[u][k] = edges[(size_t)ii * DEG + k];
            if constexpr (CHAIN) { lbn = links[2 * (size_t)ii]; lfn = links[2 * (size_t)ii + 1]; }
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const uint32_t off = lane4 | ((i + u < last) ? 0u : 0x80000000u);
#pragma unroll
            for (int k = 0; k < DEG; k++) {
                if constexpr (FIRST) {
                    const int ii = (i + u < last) ? i + u : last - 1;
                    xx[u][k] = ld_row(nbase, (uint32_t)nodes[(size_t)ii * DEG + k] * kRowBytes, off);
                } else xx[u][k] = ld_row(base, (uint32_t)ee[u][k] * kRowBytes, off);
            }
            if constexpr (CHAIN) xcn = ld_row(cbase, (uint32_t)(lbn > 0 ? lbn - 1 : 0) * kRowBytes, (i + u < last && lbn > 0 && CH.on) ? lane4 : (lane4 | 0x80000000u));
        }
    };
    // complemented magnitude (nz-1-|.|) of a stored message, and the stored form of (complemented magnitude, positive flag)
    // (Storing the messages as [sign | complemented magnitude] instead -- re-labelled root tables on the variable side, one AND here
    // and one OR on the way out -- saves 7 of these instructions per edge and was measured: slower in both directions (the four hot
    // table entries of a converged decoder, both operands saturated with either sign, land pairwise 128 bytes apart = on one LDS
    // bank, where the labels 0 and 2 nz - 1 differ in every bit: DVB-S2 265 -> 248 k codewords/s) and no faster with only the
    // variable-to-check direction re-labelled (the pass streams at the HBM rate by now: 253.1 against 255.8 k, four interleaved runs).)
    auto to_mc = [&](uint32_t xh) -> uint32_t {
        const uint32_t pos = xh & SB;
        return (xh ^ (pos - (pos >> sbit))) & LOW;           // positive label: LOW - magnitude code; negative: the code itself
    };
    auto from_mc = [&](uint32_t mc, uint32_t po) -> uint32_t {
        return xor_or(mc, po - (po >> sbit), po);            // positive nz+m = ((m^LOW)^LOW)|SB, negative nz-1-m = m^LOW
    };
    auto eval = [&](int i, const uint32_t (&x)[UNR][DEG], const int (&e)[UNR][DEG]) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            if (i + u >= last) break;
            uint32_t r[DEG];
            uint32_t tn;
            if constexpr (DEG <= ABO) {
            // The extrinsic magnitude of an edge, `mag == min1 ? min2 : min1` (src/LDPC_Code_LUT.cpp:367-390), IS the minimum over the
            // OTHER edges of the check.  All DEG of them come out of suffix minima, a running prefix minimum and one combination
            // per inner edge -- 3 (DEG - 2) two-input minima, the least any scheme needs -- instead of two compare-selects per edge on
            // the way in plus an equality test and a select per edge on the way out: 135 instead of 215 vector instructions for a
            // degree-7 check (eight frames).  Everything runs on COMPLEMENTED magnitudes (nz-1-|.|: a negative label's low bits as
            // they are), so the minima are maxima and the sign step at the end stays one three-input bit operation.
            // max: (b ^ LOW) + a carries into bit sbit <=> a > b (one v_xad_u32; fields never overflow: a + (LOW - b) <= 2 LOW)
            auto mx = [&](uint32_t a, uint32_t b) -> uint32_t {
                const uint32_t gt = xad(b, LOW, a) & SB;
                return bfi(gt - (gt >> sbit), a, b);
            };
            uint32_t mcs[DEG], spp = 0;
#pragma unroll
            for (int k = 0; k < DEG; k++) {
                const uint32_t xh = x[u][k];
                mcs[k] = to_mc(xh);
                if (k & 1) spp = xor3(spp, x[u][k - 1], xh);                  // sign bits add up in bit sbit (masked below)
                else if (k == DEG - 1) spp ^= xh;
            }
            tn = (spp ^ odd) & SB;                                            // parity of the negative inputs (bit sbit)
            uint32_t oc[DEG];
            if constexpr (DEG == 1) oc[0] = 0u;                               // (never launched: fill_cn_fast refuses degree 1)
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
            for (int k = 0; k < DEG; k++) r[k] = from_mc(oc[k], (tn ^ x[u][k]) & SB);      // extrinsic sign: positive flag
            } else {
            // wide checks: the suffix array would not fit the registers.  Input sweep: magnitudes, running two smallest, sign
            // parity.  The first two edges need no comparison against the initial values (after them min1 <= min2 are simply the sorted pair).
            uint32_t min1 = LOW, min2 = LOW, spp = 0;
            uint32_t mg[DEG];
#pragma unroll
            for (int k = 0; k < DEG; k++) {
                const uint32_t xh = x[u][k];
                const uint32_t mag = to_mc(xh) ^ LOW;
                spp ^= xh;                                                // sign bits add up in bit sbit (masked below)
                mg[k] = mag;
                if (k == 0) {
                    min1 = mag;
                } else {
                    const uint32_t g1 = ((mag | SB) - min1) & SB;         // mag >= min1
                    const uint32_t k1 = g1 - (g1 >> sbit);
                    const uint32_t lo = bfi(k1, min1, mag);
                    const uint32_t hi = mag ^ min1 ^ lo;
                    if (k == 1) {
                        min2 = hi;
                    } else {
                        const uint32_t g2 = ((min2 | SB) - hi) & SB;      // min2 >= hi
                        const uint32_t k2 = g2 - (g2 >> sbit);
                        min2 = bfi(k2, hi, min2);
                    }
                    min1 = lo;
                }
            }
            tn = (spp ^ odd) & SB;                                            // parity of the negative inputs (bit sbit)
            // output sweep: magnitude = (mag == min1 ? min2 : min1), selected directly in complemented form
            // (x ^ LOW = nz-1-x), so that the sign step is one op: positive nz+m = ((m^LOW)^LOW)|SB, negative nz-1-m = m^LOW
            const uint32_t m1c = min1 ^ LOW, m2c = min2 ^ LOW;
#pragma unroll
            for (int k = 0; k < DEG; k++) {
                const uint32_t eq = ~(((mg[k] ^ min1) | SB) - ONE) & SB;      // this edge holds the minimum
                const uint32_t ke = eq - (eq >> sbit);
                r[k] = from_mc(bfi(ke, m2c, m1c), (tn ^ x[u][k]) & SB);       // extrinsic sign: positive flag
            }
            }
            if (P.check) failw |= tn >> sbit;
            if constexpr (CHAIN) {
                [[maybe_unused]] constexpr int F = 4 * PACK, BITS = 8 / PACK;
                if (lb && CH.hard) {
                    // decided bit of the node shared with the previous check = sign of the message it sent to this check last
                    // iteration (unanimous whenever the frame passes the exit test, src/LDPC_Code_LUT.cpp:437-452)
                    const uint32_t neg = (~x[u][0] >> sbit) & ONE;
                    store_row_masked<PACK>(reinterpret_cast<uint32_t *>(hard + ((size_t)g * (size_t)P.N + (size_t)(lb - 1)) * kRowBytes + lane4), neg, smask);
                }
                if (lb && CH.on) {       // wave-uniform: the node shared with the previous check
                    uint32_t o_prev = 0, o_this = 0, dif = 0;
                    // src/LUT_Tree.cpp:774-790 for two inputs: the message to one check is ROOT(message from the other, channel)
                    if constexpr (PACK == 2) {
                        // the table labels (message | channel << shift, below 256) of four frames are formed at once, one per byte --
                        // low nibbles of the rows = frames 0..3, high nibbles = frames 4..7 -- and a frame costs one extraction
                        // per look-up instead of three extractions and two merges for its two look-ups
                        constexpr uint32_t NIB = 0x0F0F0F0Fu;
                        const uint32_t c_lo = xc & NIB, c_hi = (xc >> 4) & NIB;
                        const uint32_t ia_lo = lshl_or(c_lo, CH.tab_shift, pend & NIB), ia_hi = lshl_or(c_hi, CH.tab_shift, (pend >> 4) & NIB);
                        const uint32_t ib_lo = lshl_or(c_lo, CH.tab_shift, r[0] & NIB), ib_hi = lshl_or(c_hi, CH.tab_shift, (r[0] >> 4) & NIB);
#pragma unroll 1
                        for (int s = 0; s < 32; s += 8) {          // byte s / 8 of the label words: frames s / 8 and 4 + s / 8
                            const uint32_t vp0 = lds_tab[__builtin_amdgcn_ubfe(ib_lo, (uint32_t)s, 8u)], vt0 = lds_tab[__builtin_amdgcn_ubfe(ia_lo, (uint32_t)s, 8u)];
                            const uint32_t vp1 = lds_tab[__builtin_amdgcn_ubfe(ib_hi, (uint32_t)s, 8u)], vt1 = lds_tab[__builtin_amdgcn_ubfe(ia_hi, (uint32_t)s, 8u)];
                            o_prev = lshl_or(vp0, s, o_prev);
                            o_this = lshl_or(vt0, s, o_this);
                            o_prev = lshl_or(vp1, s + 4, o_prev);
                            o_this = lshl_or(vt1, s + 4, o_this);
                        }
                    } else {
#pragma unroll 1
                    for (int s = 0; s < F * BITS; s += 2 * BITS) {
#pragma unroll
                        for (int t = 0; t < 2; t++) {
                            const int sb = s + t * BITS;
                            const uint32_t a = __builtin_amdgcn_ubfe(pend, (uint32_t)sb, (uint32_t)BITS), b = __builtin_amdgcn_ubfe(r[0], (uint32_t)sb, (uint32_t)BITS);
                            const uint32_t c = __builtin_amdgcn_ubfe(xc, (uint32_t)sb, (uint32_t)BITS);
                            const uint32_t vp = lds_tab[lshl_or(c, CH.tab_shift, b)], vt = lds_tab[lshl_or(c, CH.tab_shift, a)];
                            o_prev = lshl_or(vp, sb, o_prev);
                            o_this = lshl_or(vt, sb, o_this);
                        }
                    }
                    }
                    if (CH.check) dif = ((o_prev ^ o_this) >> CH.sbit_out) & ONE;    // the two outgoing signs differ (all frames of the dword at once)
                    chainfail |= dif;
                    st_row(base, (uint32_t)__builtin_amdgcn_readfirstlane(pend_e) * kRowBytes, lane4, bfi(smask, o_prev, pend_old));      // (wave-uniform: keeps the row offset scalar)
                    r[0] = o_this;
                }
                if (lf && CH.on) { pend = r[1]; pend_e = e[u][1]; pend_old = x[u][1]; }
#pragma unroll
                for (int k = 0; k < DEG; k++)
                    if (!(k == 1 && lf && CH.on)) st_row(base, (uint32_t)e[u][k] * kRowBytes, lane4, bfi(smask, r[k], x[u][k]));
            } else if (all_active) {
#pragma unroll
                for (int k = 0; k < DEG; k++) st_row(base, (uint32_t)e[u][k] * kRowBytes, lane4, r[k]);
            } else {                                                          // frames that already terminated keep their value
#pragma unroll
                for (int k = 0; k < DEG; k++) st_row(base, (uint32_t)e[u][k] * kRowBytes, lane4, bfi(smask, r[k], x[u][k]));
            }
        }
    };
    if constexpr (DEG <= 16) {
        uint32_t x[UNR][DEG], xn[UNR][DEG];
        int e[UNR][DEG], en[UNR][DEG];
        fetch(first, x, e);
        if constexpr (CHAIN) { lb = lbn; lf = lfn; xc = xcn; }
        fetch(first + UNR, xn, en);
        eval(first, x, e);
        pipeline_entry_fence();
        for (int i = first + UNR; i < last; i += UNR) {
#pragma unroll
            for (int u = 0; u < UNR; u++)
#pragma unroll
                for (int k = 0; k < DEG; k++) { x[u][k] = xn[u][k]; e[u][k] = en[u][k]; }
            if constexpr (CHAIN) { lb = lbn; lf = lfn; xc = xcn; }
            fetch(i + UNR, xn, en);
            eval(i, x, e);
        }
    } else {
        // wide checks: a second set of rows in flight would cost DEG more registers per lane than the
        // occupancy it buys; the DEG loads of one check already cover the latency
        for (int i = first; i < last; i += UNR) {
            uint32_t x[UNR][DEG];
            int e[UNR][DEG];
            fetch(i, x, e);
            if constexpr (CHAIN) { lb = lbn; lf = lfn; xc = xcn; }
            eval(i, x, e);
        }
    }
    if (P.check) {
        uint32_t fail[PACK];
#pragma unroll
        for (int h = 0; h < PACK; h++) fail[h] = unpack_half<PACK>(failw, h);
        flag_frames<PACK>(vfail_w + P.vfail_off_w, P.vfail_stride_w, g, lane, fail, amask);
    }
    if constexpr (CHAIN) {
        if (CH.check && CH.on) {        // unanimity of the nodes updated here: part of the next exit test
            uint32_t fail[PACK];
#pragma unroll
            for (int h = 0; h < PACK; h++) fail[h] = unpack_half<PACK>(chainfail, h);
            flag_frames<PACK>(vfail_w + CH.vfail_off_w, P.vfail_stride_w, g, lane, fail, amask);
        }
    }
}

template <int DEG, int UNR, int PACK>
__global__ __launch_bounds__(256) void cn_minsum_fast_kernel(
    const FastParams *__restrict__ Pp, uint8_t *__restrict__ msgs, const uint32_t *__restrict__ state_w, uint32_t *__restrict__ vfail_w,
    const int32_t *__restrict__ fast_idx)
{
    const FastParams &P = *Pp;           // (class parameters in device memory: read with scalar loads as they are needed)
    cn_minsum_body<DEG, UNR, PACK, false>(P, (int)blockIdx.x, msgs, state_w, vfail_w, fast_idx);
}

// ------------------------------------------------------------------------------------------
// Balanced binary trees at compile time.
// N message leaves are paired front-to-back through a FIFO (src/LUT_Tree.cpp:203-235); leaves are
// then renumbered by depth-first position, because a leaf at position p consumes queue element p
// (src/LUT_Tree.cpp:402-407).  Internal nodes are numbered in creation order (children first).
template <int N>
struct BalShape {
    static constexpr int NI = N > 1 ? N - 1 : 0;      // internal (LUT) nodes below the root
    int left[NI > 0 ? NI : 1] = {};                   // child ids: < N leaf (creation id), >= N internal
    int right[NI > 0 ? NI : 1] = {};
    int size[2 * N] = {};                              // leaves below each node
    int first[2 * N] = {};                             // DFS position of its first leaf
    int off[NI > 0 ? NI : 1] = {};                    // offset of its variants in the value array
    int top = 0;                                       // id of the subtree root
    int total = 0;                                     // sum over internal nodes of (size + 1)
};

template <int N>
constexpr BalShape<N> make_bal_shape() {
    BalShape<N> S{};
    int fifo[2 * N + 2] = {};
    int head = 0, tail = 0, next = N;
    for (int l = 0; l < N; l++) { fifo[tail++] = l; S.size[l] = 1; }
    while (tail - head > 1) {
        const int a = fifo[head++], b = fifo[head++];
        S.left[next - N] = a; S.right[next - N] = b;
        S.size[next] = S.size[a] + S.size[b];
        fifo[tail++] = next++;
    }
    S.top = fifo[head];
    // DFS positions (explicit stack)
    int stack[2 * N + 2] = {};
    int sp = 0, pos = 0;
    stack[sp++] = S.top;
    // pre-order walk assigning `first`; children pushed right then left
    while (sp > 0) {
        const int nd = stack[--sp];
        S.first[nd] = pos;
        if (nd < N) { pos += 1; continue; }
        // the first leaf of an internal node is the first leaf of its left child: handled when
        // the left child is popped next (pos unchanged until a leaf is met)
        stack[sp++] = S.right[nd - N];
        stack[sp++] = S.left[nd - N];
    }
    int total = 0;
    for (int j = 0; j < BalShape<N>::NI; j++) { S.off[j] = total; total += S.size[N + j] + 1; }
    S.total = total;
    return S;
}


// One look-up of ONE frame: label = a | b << sh, table slot `t` in LDS (one v_lshl_or_b32 + one
// ds_read_u8 with the slot offset as immediate).  The tree of a node is evaluated frame by frame on
// unpacked labels: keeping four frames packed per register (SWAR) would cost four extractions and three
// merges around every group of four look-ups -- 2.6 VALU instructions per frame-look-up instead of 1 --
// and the pass is VALU-bound (gfx950 issues one wave64 VALU instruction per CU per clock).
// (Packing table entries as nibbles makes the reads bank-conflict free but costs 7 more VALU
// instructions per four look-ups; measured slower, removed.)
// The label is formed by an explicit v_lshl_or_b32: written as `a | (b << sh)` the compiler hoists the
// shared shift `b << sh` of a value that feeds several variants and ends up with MORE instructions
// (one shift per value plus one OR per label instead of one fused op per label).
__device__ __forceinline__ uint32_t lut1(const uint8_t *lds_tab, int t, uint32_t a, uint32_t b, int sh) {
    return lds_tab[t * kFastTableStride + lshl_or(b, sh, a)];
}

// value of child `c` of the balanced tree in variant kc (the first kc leaves of the subtree read
// the queue unshifted, the others read one element further: the removed message lies before them)
// (all shape look-ups go through constexpr LOCALS so that they fold to immediates; reading the
// static member through a reference would be a run-time load in device code)
template <int N, int C, int KC>
__device__ __forceinline__ uint32_t bal_child(const uint32_t *in, const uint32_t *v) {
    constexpr BalShape<N> S = make_bal_shape<N>();
    if constexpr (C < N) { constexpr int idx = S.first[C] + (KC ? 0 : 1); return in[idx]; }
    else { constexpr int idx = S.off[C - N] + KC; return v[idx]; }
}

template <int N, int J, int K>
__device__ __forceinline__ void bal_node_variant(const uint32_t *in, uint32_t *v, const uint8_t *tab, int sh) {
    constexpr BalShape<N> S = make_bal_shape<N>();
    constexpr int L = S.left[J], R = S.right[J], sl = S.size[L];
    constexpr int kl = K < sl ? K : sl, kr = K > sl ? K - sl : 0;
    constexpr int dst = S.off[J] + K;
    v[dst] = lut1(tab, J, bal_child<N, L, kl>(in, v), bal_child<N, R, kr>(in, v), sh);
}

template <int N, int J, int... Ks>
__device__ __forceinline__ void bal_node_all(const uint32_t *in, uint32_t *v, const uint8_t *tab, int sh, std::integer_sequence<int, Ks...>) {
    (bal_node_variant<N, J, Ks>(in, v, tab, sh), ...);
}
// VAR: all variants of every node; DEC (ALL = false): only the unshifted variant K = size
template <int N, bool ALL, int... Js>
__device__ __forceinline__ void bal_all_nodes(const uint32_t *in, uint32_t *v, const uint8_t *tab, int sh, std::integer_sequence<int, Js...>) {
    if constexpr (ALL) (bal_node_all<N, Js>(in, v, tab, sh, std::make_integer_sequence<int, make_bal_shape<N>().size[N + Js] + 1>{}), ...);
    else (bal_node_variant<N, Js, make_bal_shape<N>().size[N + Js]>(in, v, tab, sh), ...);
}

// Variable-node (KIND = TT_VAR) / decision (TT_DEC) pass for degree-DV nodes with balanced trees.
// Tables: slots 0..NI-1 = internal nodes in creation order, slot NI = root.
// DV == 1 (VAR only): ROOT(CHA), the build's degree-1 extension.
// `block` = index of the 4-wave block within this degree class; lds_tab: >= (NI + 1) * 256 bytes,
// staged here by the whole block (the call must be block-uniform).
template <int DV, int KIND, bool CHECK, int PACK, typename PT>
__device__ __forceinline__ void vn_balanced_body(
    const PT &P, int block, uint8_t *lds_tab, uint8_t *msgs, const uint8_t *cha, uint8_t *__restrict__ hard,
    const uint32_t *__restrict__ state_w, uint32_t *__restrict__ vfail_w, const uint8_t *__restrict__ tables,
    const int32_t *__restrict__ fast_idx)
{
    constexpr int N = (KIND == TT_DEC) ? DV : DV - 1;          // message leaves
    constexpr int NI = N > 1 ? N - 1 : 0;
    constexpr int NB = N > 1 ? N : 2;                          // shape used for array sizes when N <= 1
    // stage the class tables (canonical order, fixed 256-byte slots)
    // (table offsets and lengths are multiples of 4: lut_program.hpp pads every table)
    {
        // all table loads first, then all LDS writes: one memory latency per block instead of one per table
        uint32_t tmp[NI + 1];
        const int i = threadIdx.x;
#pragma unroll
        for (int t = 0; t <= NI; t++) tmp[t] = (i < P.tab_len[t] / 4) ? reinterpret_cast<const uint32_t *>(tables + P.tab_off[t])[i] : 0u;
#pragma unroll
        for (int t = 0; t <= NI; t++) if (i < P.tab_len[t] / 4) reinterpret_cast<uint32_t *>(lds_tab + t * kFastTableStride)[i] = tmp[t];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(block * 4 + (threadIdx.x >> 6));    // wave-uniform -> SGPR
    const int gl = wave / P.waves_per_group;
    if (gl >= P.G) return;
    const int chunk = wave - gl * P.waves_per_group;
    const int g = gl + P.g0;
    uint32_t amask[PACK];
    if (load_active<PACK>(state_w, g, lane, amask)) return;
    const uint32_t smask = pack_masks<PACK>(amask);
    const int32_t *vtab = fast_idx + P.idx_off;                 // dense [n_nodes][2] = {node id, first edge}
    const rsrc_t mbase = make_rsrc(msgs + (size_t)g * (size_t)P.E * kRowBytes, (uint32_t)P.E * kRowBytes);   // this group's rows
    const rsrc_t cbase = make_rsrc(cha + (size_t)g * (size_t)P.N * kRowBytes, (uint32_t)P.N * kRowBytes);
    uint8_t *hbase = hard + (size_t)g * (size_t)P.N * kRowBytes;
    const uint32_t lane4 = (uint32_t)lane * 4u;
    const int first = chunk * P.nodes_per_wave;
    int last = first + P.nodes_per_wave;
    if (last > P.n_nodes) last = P.n_nodes;
    const int sh = P.shift_msg, shr = P.tab_shift[NI];
    const int sbit = (KIND == TT_VAR && CHECK) ? __builtin_ctz((unsigned)P.nz | 0x100u) : 0;
    uint32_t failw = 0;                                          // failed-unanimity flags, one per frame position of the dword

    // software pipeline, same structure as cn_minsum_body: the rows of node i+1 are requested before node
    // i is evaluated, so a wave keeps DV+1 loads in flight during its LDS look-up phase
    auto fetch = [&](int i, int &vv, int &ee, uint32_t (&r)[DV + 1]) {
        const int ic = i < last ? i : last - 1;
        vv = vtab[2 * (size_t)ic]; ee = vtab[2 * (size_t)ic + 1];
        const uint32_t off = lane4 | (i < last ? 0u : 0x80000000u);      // past the end: out of range, returns 0, no access
#pragma unroll
        for (int k = 0; k < DV; k++) r[k] = ld_row(mbase, (uint32_t)(ee + k) * kRowBytes, off);
        r[DV] = ld_row(cbase, (uint32_t)vv * kRowBytes, off);
    };
    auto eval = [&](const uint32_t (&raw)[DV + 1], int v, int e0) {
        constexpr int F = 4 * PACK, BITS = 8 / PACK;             // frames per dword, bits per label
        constexpr int U = DV <= 8 ? 2 : 1;                       // frames evaluated per trip (independent: ILP for the LDS latency)
        constexpr BalShape<NB> SH = make_bal_shape<NB>();
        constexpr int top_off = SH.off[SH.top - NB];
        uint32_t out[DV], hardw = 0;
#pragma unroll
        for (int o = 0; o < DV; o++) out[o] = 0;
        // a real loop over the frames of the dword (not unrolled: one frame's tree keeps ~DV + sum(size+1)
        // values live, unrolling all frames lets the scheduler interleave them and blows the register budget)
#pragma unroll 1
        for (int sp = 0; sp < F * BITS; sp += U * BITS) {
#pragma unroll
            for (int u = 0; u < U; u++) {                        // frame at bits [s, s + BITS) of every row dword
                const int s = sp + u * BITS;
                uint32_t in[DV + 1];
#pragma unroll
                for (int k = 0; k <= DV; k++) in[k] = __builtin_amdgcn_ubfe(raw[k], (uint32_t)s, (uint32_t)BITS);
                const uint32_t ch = in[DV];
                uint32_t val[(N > 1 ? SH.total : 1)];
                if constexpr (N > 1) bal_all_nodes<N, KIND == TT_VAR>(in, val, lds_tab, sh, std::make_integer_sequence<int, NI>{});
                if constexpr (KIND == TT_DEC) {
                    uint32_t top;
                    if constexpr (N > 1) top = val[top_off + N];
                    else top = in[0];
                    hardw = lshl_or(lut1(lds_tab, NI, top, ch, shr) < 1u ? 1u : 0u, s, hardw);     // src/LDPC_Code_LUT.cpp:342
                } else {
#pragma unroll
                    for (int o = 0; o < DV; o++) {
                        uint32_t r;
                        if constexpr (N == 0) {
                            // degree 1: the only leaf is the channel label (a one-input table: label = ch)
                            r = lut1(lds_tab, 0, ch, 0u, 0);
                        } else {
                            uint32_t top;
                            if constexpr (N > 1) top = val[top_off + o];
                            else top = in[o == 0 ? 1 : 0];                     // N == 1: the other message
                            r = lut1(lds_tab, NI, top, ch, shr);
                        }
                        out[o] = lshl_or(r, s, out[o]);
                    }
                }
            }
        }
        if constexpr (KIND == TT_VAR && CHECK) {
            // sign of a label = bit sbit (nz = 1 << sbit): negative (bit 1 decided) <=> bit clear; the node fails the
            // unanimity test when any two outgoing signs differ -- on the packed rows, all frames of the dword at once
            constexpr uint32_t ONE = PACK == 2 ? 0x11111111u : 0x01010101u;
            uint32_t diff = 0;
#pragma unroll
            for (int o = 1; o < DV; o++) diff |= out[o] ^ out[0];
            failw |= (diff >> sbit) & ONE;
            if (P.write_hard) hardw = (~out[0] >> sbit) & ONE;      // (wave-uniform; off when the bits are recovered at the end)
        }
        if constexpr (KIND == TT_DEC) {
            store_row_masked<PACK>(reinterpret_cast<uint32_t *>(hbase + (size_t)v * kRowBytes + lane4), hardw, smask);
        } else {
#pragma unroll
            for (int o = 0; o < DV; o++) st_row(mbase, (uint32_t)(e0 + o) * kRowBytes, lane4, bfi(smask, out[o], raw[o]));      // (one instruction: cheaper than selecting the plain value for an all-active wave)
            if (CHECK && P.write_hard)
                store_row_masked<PACK>(reinterpret_cast<uint32_t *>(hbase + (size_t)v * kRowBytes + lane4), hardw, smask);
        }
    };
    int v, e0, vn, en;
    uint32_t raw[DV + 1], nxt[DV + 1];
    fetch(first, v, e0, raw);
    fetch(first + 1, vn, en, nxt);
    eval(raw, v, e0);
    pipeline_entry_fence();
    for (int i = first + 1; i < last; i++) {
#pragma unroll
        for (int k = 0; k <= DV; k++) raw[k] = nxt[k];
        v = vn; e0 = en;
        fetch(i + 1, vn, en, nxt);
        eval(raw, v, e0);
    }
    if (KIND == TT_VAR && CHECK) {
        uint32_t fail[PACK];
#pragma unroll
        for (int h = 0; h < PACK; h++) fail[h] = unpack_half<PACK>(failw, h);
        flag_frames<PACK>(vfail_w + P.vfail_off_w, P.vfail_stride_w, g, lane, fail, amask);
    }
}

template <int DV, int KIND, bool CHECK, int PACK>
__global__ __launch_bounds__(256) void vn_balanced_fast_kernel(
    const FastParams *__restrict__ Pp, uint8_t *msgs, const uint8_t *cha, uint8_t *__restrict__ hard,
    const uint32_t *__restrict__ state_w, uint32_t *__restrict__ vfail_w, const uint8_t *__restrict__ tables,
    const int32_t *__restrict__ fast_idx)
{
    constexpr int NT = (KIND == TT_DEC ? DV : DV - 1) > 1 ? (KIND == TT_DEC ? DV : DV - 1) : 1;   // LUT nodes incl. root
    __shared__ __attribute__((aligned(16))) uint8_t lds_tab[NT * kFastTableStride];
    const FastParams &P = *Pp;
    vn_balanced_body<DV, KIND, CHECK, PACK>(P, (int)blockIdx.x, lds_tab, msgs, cha, hard, state_w, vfail_w, tables, fast_idx);
}

// ------------------------------------------------------------------------------------------
// Skewed two-half pipeline: ONE launch that runs a check pass on one half of the frame groups and a
// variable pass on the other half.  The variable pass is bound by LDS look-ups (34 per frame for a
// degree-8 node), the check pass by HBM streaming and a little VALU work; taken alone each leaves the
// other resource idle.  The two halves of a batch are independent decodes, so the host runs them half
// an iteration out of phase (decoder.hip: decode_tiles_skewed) and every launch mixes both kinds of
// work on every CU: blocks are handed to roles (degree class x pass kind x half) through an
// interleaved item table, each block stays homogeneous (one role) so its tables sit at LDS offset 0.
// occupancy window of the first bucket's fused kernel (build-time knobs for tools/ab_variants.sh; 7..8 measured best)
#ifndef LUTLDPC_B0_WAVES_MIN
#define LUTLDPC_B0_WAVES_MIN 7
#endif
#ifndef LUTLDPC_B0_WAVES_MAX
#define LUTLDPC_B0_WAVES_MAX 8
#endif
constexpr int kFusedMaxRoles = 10;
constexpr int kFusedMaxTables = 20;
// Degree buckets of the fused kernel.  Its register count is the maximum over all the cases it contains
// (every degree up to the bucket limits), so codes with small degrees get their own, leaner instantiation:
//   bucket 0: variable degrees <= 8,  check degrees <= 8   (64 VGPRs, 8 waves per SIMD)
//   bucket 1:                  <= 12,               <= 16
//   bucket 2:                  <= 20,               <= 32
//   bucket 3:                  <= 8,                <= 10   (added last, ranked second: kFusedBucketOrder)
constexpr int kFusedBuckets = 4;
constexpr int kFusedVnDeg[kFusedBuckets] = {8, 12, 20, 8};
constexpr int kFusedCnDeg[kFusedBuckets] = {8, 16, 32, 10};
// checks up to this degree use the all-but-one minima inside the fused kernel of a bucket (cn_minsum_body: ABO): the suffix array of a
// degree-16 check costs bucket 1 two waves per SIMD (79 -> 104 registers), of a degree-10 check bucket 3 one (63 -> 72)
constexpr int kFusedAboDeg[kFusedBuckets] = {8, 10, 16, 8};
// the bucket a code runs in: the leanest one that holds its degrees -- bucket 3 (variable degrees <= 8, check degrees <= 10, 8 waves
// per SIMD like bucket 0 but with SGPR spills in the widest check bodies) sits between buckets 0 and 1: the irregular N = 64800
// code of the reference has five checks of degree 9 among 32400 (+4.5 % over bucket 1, which runs 5 waves per SIMD)
constexpr int kFusedBucketOrder[kFusedBuckets] = {0, 3, 1, 2};
inline int fused_bucket(int max_vn_deg, int max_cn_deg) {
    for (int b : kFusedBucketOrder) if (max_vn_deg <= kFusedVnDeg[b] && max_cn_deg <= kFusedCnDeg[b]) return b;
    return -1;
}
inline int fused_bucket_rank(int bucket) { for (int i = 0; i < kFusedBuckets; i++) if (kFusedBucketOrder[i] == bucket) return i; return -1; }
inline constexpr int fused_max_cn_deg() { int m = 0; for (int b = 0; b < kFusedBuckets; b++) m = kFusedCnDeg[b] > m ? kFusedCnDeg[b] : m; return m; }

struct RoleParams {
    int32_t kind;          // 0: min-sum check class, 1: variable class
    int32_t deg;
    int32_t g0, G;         // frame groups g0 .. g0+G-1
    int32_t n_nodes, nodes_per_wave, waves_per_group, idx_off;
    int32_t E, N, nz, shift_msg, check, write_hard;
    int32_t vfail_stride_w, vfail_off_w;
    int32_t first, nidx_off;   // check roles of iteration 0: inputs from the initial-message rows, through the node table at nidx_off
    ChainParams chain;     // check roles only
    int32_t tab_off[kFusedMaxTables], tab_len[kFusedMaxTables], tab_shift[kFusedMaxTables];
};
// host-side staging of the roles of one launch (decoder.hip builds these once per (batch shape, exit conditions) and keeps
// them in DEVICE memory: the kernel gets a pointer, not the 3.4 KB by value)
struct FusedParams {
    int32_t n_roles;
    int32_t prio;          // 1: raise the issue priority of the look-up-heavy waves (s_setprio)
    RoleParams role[kFusedMaxRoles];
};

template <int PACK, bool CHAIN, bool FIRST, int ABO, int... Ds>
__device__ __forceinline__ void fused_cn_switch(const RoleParams &P, int block, std::integer_sequence<int, Ds...>, uint8_t *msgs, const uint32_t *state_w,
                                                uint32_t *vfail_w, const int32_t *fast_idx, uint8_t *lds_tab, const uint8_t *cha, const uint8_t *tables, uint8_t *hard,
                                                const uint8_t *msg0) {
    ((P.deg == Ds + 2 ? (cn_minsum_body<Ds + 2, 1, PACK, CHAIN, RoleParams, FIRST, ABO>(P, block, msgs, state_w, vfail_w, fast_idx, P.chain, lds_tab, cha, tables, hard, msg0), 0) : 0), ...);
}
template <int PACK, bool CHECK, int... Ds>
__device__ __forceinline__ void fused_vn_switch(const RoleParams &P, int block, std::integer_sequence<int, Ds...>, uint8_t *lds_tab, uint8_t *msgs,
                                                const uint8_t *cha, uint8_t *hard, const uint32_t *state_w, uint32_t *vfail_w, const uint8_t *tables,
                                                const int32_t *fast_idx) {
    ((P.deg == Ds + 1 ? (vn_balanced_body<Ds + 1, TT_VAR, CHECK, PACK>(P, block, lds_tab, msgs, cha, hard, state_w, vfail_w, tables, fast_idx), 0) : 0), ...);
}

// items[b] = {role, block index within the role}; roles[] = the roles of THIS launch, in device memory (read-only
// for the whole decode: the role is picked with a wave-uniform run-time index, its fields come in through scalar
// loads as they are needed).  The kernel-argument segment stays a handful of pointers.
template <int PACK, bool CHECK, int BUCKET>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((BUCKET == 0 || BUCKET == 3) ? LUTLDPC_B0_WAVES_MIN : BUCKET == 1 ? 4 : 3, (BUCKET == 0 || BUCKET == 3) ? LUTLDPC_B0_WAVES_MAX : 8))) void pass_fused_kernel(
    const RoleParams *__restrict__ roles, const int2 *__restrict__ items, int prio, uint8_t *msgs, const uint8_t *cha, uint8_t *__restrict__ hard,
    const uint32_t *__restrict__ state_w, uint32_t *__restrict__ vfail_w, const uint8_t *__restrict__ tables, const int32_t *__restrict__ fast_idx,
    const uint8_t *__restrict__ msg0)
{
    constexpr int MAXVN = kFusedVnDeg[BUCKET], MAXCN = kFusedCnDeg[BUCKET];
    __shared__ __attribute__((aligned(16))) uint8_t lds_tab[MAXVN * kFastTableStride];
    const int2 it = items[blockIdx.x];
    const int r = __builtin_amdgcn_readfirstlane(it.x), rb = __builtin_amdgcn_readfirstlane(it.y);
    const RoleParams &P = roles[r];
    if (prio && P.kind) {                          // LUT-heavy waves first: they are the long ones
        if (P.deg >= 4) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);
    }
    if (P.kind == 0) {
        // chain fusion (dual-diagonal codes: DVB-S2 at every rate, IRA): every bucket has the chained check bodies
        constexpr auto degs = std::make_integer_sequence<int, MAXCN - 1>{};
        if (!P.first) {
            if (P.chain.on || P.chain.hard) fused_cn_switch<PACK, true, false, kFusedAboDeg[BUCKET]>(P, rb, degs, msgs, state_w, vfail_w, fast_idx, lds_tab, cha, tables, hard, msg0);
            else fused_cn_switch<PACK, false, false, kFusedAboDeg[BUCKET]>(P, rb, degs, msgs, state_w, vfail_w, fast_idx, lds_tab, cha, tables, hard, msg0);
        } else {                           // iteration 0 (two launches per decode): inputs from the initial-message rows
            if (P.chain.on) fused_cn_switch<PACK, true, true, kFusedAboDeg[BUCKET]>(P, rb, degs, msgs, state_w, vfail_w, fast_idx, lds_tab, cha, tables, hard, msg0);
            else fused_cn_switch<PACK, false, true, kFusedAboDeg[BUCKET]>(P, rb, degs, msgs, state_w, vfail_w, fast_idx, lds_tab, cha, tables, hard, msg0);
        }
    }
    else fused_vn_switch<PACK, CHECK>(P, rb, std::make_integer_sequence<int, MAXVN>{}, lds_tab, msgs, cha, hard, state_w, vfail_w, tables, fast_idx);
}

// ------------------------------------------------------------------------------------------
// host side: does a runtime tree have the balanced shape for degree d?  If so return its LUT
// nodes in canonical order (internal nodes in creation order, then the root).
inline bool match_balanced(const Tree &t, int kind, int d, std::vector<const TreeNode *> &canon) {
    canon.clear();
    const int n = (kind == TT_DEC) ? d : d - 1;
    const TreeNode *root = t.root.get();
    if (!root || root->type != NT_ROOT) return false;
    if (n == 0) {      // degree-1 extension: ROOT(CHA)
        if (root->child.size() != 1 || root->child[0]->type != NT_CHA) return false;
        canon.push_back(root);
        return true;
    }
    if (root->child.size() != 2 || root->child[1]->type != NT_CHA || !root->child[1]->child.empty()) return false;
    // rebuild the canonical shape at run time (same FIFO pairing as make_bal_shape)
    std::vector<int> left, right, fifo;
    for (int l = 0; l < n; l++) fifo.push_back(l);
    size_t head = 0;
    int next = n;
    while (fifo.size() - head > 1) {
        left.push_back(fifo[head]); right.push_back(fifo[head + 1]);
        head += 2;
        fifo.push_back(next++);
    }
    const int top = fifo[head];
    std::vector<const TreeNode *> node_of((size_t)(2 * n), nullptr);
    // simultaneous walk
    struct Item { int id; const TreeNode *tn; };
    std::vector<Item> stack{{top, root->child[0].get()}};
    while (!stack.empty()) {
        Item it = stack.back(); stack.pop_back();
        if (it.id < n) {
            if (it.tn->type != NT_MSG || !it.tn->child.empty()) return false;
            continue;
        }
        if (it.tn->type != NT_IM || it.tn->child.size() != 2) return false;
        node_of[(size_t)it.id] = it.tn;
        stack.push_back({right[(size_t)(it.id - n)], it.tn->child[1].get()});
        stack.push_back({left[(size_t)(it.id - n)], it.tn->child[0].get()});
    }
    for (int j = n; j < 2 * n - 1; j++) { if (!node_of[(size_t)j]) return false; canon.push_back(node_of[(size_t)j]); }
    canon.push_back(root);
    return true;
}

inline bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

// Per-class plan of a fast variable/decision pass (filled at create time)
struct FastClassPlan {
    bool ok = false;
    FastParams P{};
};

// tab_of: LUT node -> (offset in the global blob, length)
inline FastClassPlan plan_fast_vn(const Tree &t, int kind, int d, const std::map<const TreeNode *, std::pair<uint32_t, uint32_t>> &tab_of, int node_off, int n_nodes) {
    FastClassPlan fp;
    std::vector<const TreeNode *> canon;
    if (d < 1 || d > 20 || !match_balanced(t, kind, d, canon) || (int)canon.size() > kFastMaxTables) return fp;
    int shift_msg = -1;
    for (size_t j = 0; j < canon.size(); j++) {
        const TreeNode *nd = canon[j];
        auto it = tab_of.find(nd);
        if (it == tab_of.end() || it->second.second > 256u) return fp;
        for (auto &c : nd->child) if (!is_pow2(c->K) || c->K > 128) return fp;
        if (!is_pow2(nd->K) || nd->K > 128) return fp;
        const int sh0 = __builtin_ctz((unsigned)nd->child[0]->K);
        if (j + 1 < canon.size()) {                     // internal node: both children carry messages
            if (nd->child[0]->K != nd->child[1]->K) return fp;
            if (shift_msg < 0) shift_msg = sh0; else if (shift_msg != sh0) return fp;
        }
        fp.P.tab_off[j] = (int32_t)it->second.first;
        fp.P.tab_len[j] = (int32_t)it->second.second;
        fp.P.tab_shift[j] = sh0;
    }
    fp.P.shift_msg = shift_msg < 0 ? 0 : shift_msg;
    fp.P.n_tables = (int)canon.size();
    fp.P.nib = 0;      // nibble-packed LDS tables (lut4<true>): conflict-free but measured slower on MI355X, not instantiated
    fp.P.deg = d; fp.P.node_off = node_off; fp.P.n_nodes = n_nodes;
    fp.ok = true;
    return fp;
}

// ------------------------------------------------------------------------------------------
// Host-side launchers.  They are ordinary (non-inline) function templates, explicitly instantiated in
// their own translation units (fast_vn_var.hip, fast_vn_dec.hip, fast_cn.hip, fused.hip) so that the
// ~250 kernel instantiations compile in parallel; decoder.hip sees `extern template` declarations.
constexpr int kFastMaxDeg = 20;      // variable / decision nodes
constexpr int kFastMaxCnDeg = 32;    // check nodes

template <int KIND, bool CHECK, int PACK, int DV>
void launch_vn_fast_one(hipStream_t s, const FastParams &P, const FastParams *dP, uint8_t *msgs, const uint8_t *cha, uint8_t *hard, const uint32_t *state_w,
                        uint32_t *vfail_w, const uint8_t *tables, const int32_t *fast_idx) {
    const int waves = P.waves_per_group * P.G;
    launch_k(vn_balanced_fast_kernel<DV, KIND, CHECK, PACK>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, dP, msgs, cha, hard, state_w, vfail_w,
             tables, fast_idx);
}

template <int KIND, bool CHECK, int PACK, int... DVs>
bool dispatch_vn_fast(int deg, std::integer_sequence<int, DVs...>, hipStream_t s, const FastParams &P, const FastParams *dP, uint8_t *msgs, const uint8_t *cha,
                      uint8_t *hard, const uint32_t *state_w, uint32_t *vfail_w, const uint8_t *tables, const int32_t *fast_idx) {
    bool done = false;
    ((deg == DVs + 1 ? (launch_vn_fast_one<KIND, CHECK, PACK, DVs + 1>(s, P, dP, msgs, cha, hard, state_w, vfail_w, tables, fast_idx), done = true) : false), ...);
    return done;
}

// launch one class; returns false when the degree has no instantiation.  P is complete (fill_vn_fast), dP its copy in device memory.
inline void fill_vn_fast(FastParams &P, int G, int nz, int check, int write_hard, int nodes_per_wave, int E, int N, int vfail_stride_w) {
    P.vfail_stride_w = vfail_stride_w;
    P.G = G; P.E = E; P.N = N; P.nz = nz; P.check = check; P.write_hard = write_hard;
    P.nodes_per_wave = nodes_per_wave;
    P.waves_per_group = (P.n_nodes + nodes_per_wave - 1) / nodes_per_wave;
}
template <int KIND, int PACK>
bool launch_vn_fast(hipStream_t s, const FastParams &P, const FastParams *dP, uint8_t *msgs, const uint8_t *cha,
                    uint8_t *hard, const uint32_t *state_w, uint32_t *vfail_w, const uint8_t *tables, const int32_t *fast_idx) {
    constexpr auto seq = std::make_integer_sequence<int, kFastMaxDeg>{};
    if (KIND == TT_VAR && P.check) return dispatch_vn_fast<KIND, true, PACK>(P.deg, seq, s, P, dP, msgs, cha, hard, state_w, vfail_w, tables, fast_idx);
    return dispatch_vn_fast<KIND, false, PACK>(P.deg, seq, s, P, dP, msgs, cha, hard, state_w, vfail_w, tables, fast_idx);
}

template <int PACK, int DEG>
void launch_cn_fast_one(hipStream_t s, const FastParams &P, const FastParams *dP, uint8_t *msgs, const uint32_t *state_w, uint32_t *vfail_w, const int32_t *fast_idx) {
    // checks evaluated per pipeline step: more = more loads outstanding per wave, fewer = fewer VGPRs = more waves
    constexpr int UNR = DEG <= 4 ? 4 : DEG <= 10 ? 2 : 1;
    const int waves = P.waves_per_group * P.G;
    launch_k(cn_minsum_fast_kernel<DEG, UNR, PACK>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, dP, msgs, state_w, vfail_w, fast_idx);
}
template <int PACK, int... Ds>
bool dispatch_cn_fast(int deg, std::integer_sequence<int, Ds...>, hipStream_t s, const FastParams &P, const FastParams *dP, uint8_t *msgs, const uint32_t *state_w,
                      uint32_t *vfail_w, const int32_t *fast_idx) {
    bool done = false;
    ((deg == Ds + 1 ? (launch_cn_fast_one<PACK, Ds + 1>(s, P, dP, msgs, state_w, vfail_w, fast_idx), done = true) : false), ...);
    return done;
}

// min-sum: one launch per degree class (P from fill_cn_fast, dP its copy in device memory)
inline bool fill_cn_fast(FastParams &P, int deg, int n_nodes, int idx_off, int G, int E, int nz, int check, int nodes_per_wave, int vfail_stride_w) {
    if (!is_pow2(nz) || nz > 64 || deg < 2 || deg > kFastMaxCnDeg) return false;
    P = FastParams{};
    P.n_nodes = n_nodes; P.idx_off = idx_off; P.G = G; P.E = E; P.nz = nz; P.check = check; P.deg = deg; P.vfail_stride_w = vfail_stride_w;
    P.nodes_per_wave = nodes_per_wave;
    P.waves_per_group = (n_nodes + nodes_per_wave - 1) / nodes_per_wave;
    return true;
}
template <int PACK>
bool launch_cn_fast(hipStream_t s, const FastParams &P, const FastParams *dP, uint8_t *msgs, const uint32_t *state_w, uint32_t *vfail_w, const int32_t *fast_idx) {
    return dispatch_cn_fast<PACK>(P.deg, std::make_integer_sequence<int, kFastMaxCnDeg>{}, s, P, dP, msgs, state_w, vfail_w, fast_idx);
}

// code-object preload of the per-class translation units (see preload_fused)
template <int KIND, int PACK>
hipError_t preload_vn_fast() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&vn_balanced_fast_kernel<2, KIND, false, PACK>));
}
template <int PACK>
hipError_t preload_cn_fast() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&cn_minsum_fast_kernel<2, 4, PACK>));
}

// skewed pipeline: one launch of pass_fused_kernel over n_blocks items
template <int PACK, int BUCKET>
void launch_fused(hipStream_t s, const RoleParams *d_roles, const int32_t *items, int n_blocks, int prio, bool vn_check, uint8_t *msgs, const uint8_t *cha, uint8_t *hard,
                  const uint32_t *state_w, uint32_t *vfail_w, const uint8_t *tables, const int32_t *fast_idx, const uint8_t *msg0) {
    if (vn_check)
        launch_k(pass_fused_kernel<PACK, true, BUCKET>, dim3((unsigned)n_blocks), dim3(256), 0, s, d_roles, reinterpret_cast<const int2 *>(items), prio, msgs, cha, hard,
                 state_w, vfail_w, tables, fast_idx, msg0);
    else
        launch_k(pass_fused_kernel<PACK, false, BUCKET>, dim3((unsigned)n_blocks), dim3(256), 0, s, d_roles, reinterpret_cast<const int2 *>(items), prio, msgs, cha, hard,
                 state_w, vfail_w, tables, fast_idx, msg0);
}
// force the code object of this translation unit onto the current device now (HIP loads code objects lazily, at the first
// launch of one of their kernels): decoder.hip calls these at decoder creation, see preload_code_objects
template <int PACK, int BUCKET>
hipError_t preload_fused() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&pass_fused_kernel<PACK, false, BUCKET>));
}

#define LUTLDPC_FUSED_SIG (hipStream_t, const RoleParams *, const int32_t *, int, int, bool, uint8_t *, const uint8_t *, uint8_t *, const uint32_t *, uint32_t *, const uint8_t *, const int32_t *, const uint8_t *)
#define LUTLDPC_FAST_LAUNCHERS(X)                                                                                                           \
    X template bool launch_vn_fast<TT_VAR, 1>(hipStream_t, const FastParams &, const FastParams *, uint8_t *, const uint8_t *, uint8_t *, const uint32_t *, uint32_t *, const uint8_t *, const int32_t *); \
    X template bool launch_vn_fast<TT_VAR, 2>(hipStream_t, const FastParams &, const FastParams *, uint8_t *, const uint8_t *, uint8_t *, const uint32_t *, uint32_t *, const uint8_t *, const int32_t *); \
    X template bool launch_vn_fast<TT_DEC, 1>(hipStream_t, const FastParams &, const FastParams *, uint8_t *, const uint8_t *, uint8_t *, const uint32_t *, uint32_t *, const uint8_t *, const int32_t *); \
    X template bool launch_vn_fast<TT_DEC, 2>(hipStream_t, const FastParams &, const FastParams *, uint8_t *, const uint8_t *, uint8_t *, const uint32_t *, uint32_t *, const uint8_t *, const int32_t *); \
    X template bool launch_cn_fast<1>(hipStream_t, const FastParams &, const FastParams *, uint8_t *, const uint32_t *, uint32_t *, const int32_t *);    \
    X template bool launch_cn_fast<2>(hipStream_t, const FastParams &, const FastParams *, uint8_t *, const uint32_t *, uint32_t *, const int32_t *);    \
    X template void launch_fused<1, 0> LUTLDPC_FUSED_SIG; X template void launch_fused<2, 0> LUTLDPC_FUSED_SIG; \
    X template void launch_fused<1, 1> LUTLDPC_FUSED_SIG; X template void launch_fused<2, 1> LUTLDPC_FUSED_SIG; \
    X template void launch_fused<1, 2> LUTLDPC_FUSED_SIG; X template void launch_fused<2, 2> LUTLDPC_FUSED_SIG; \
    X template void launch_fused<1, 3> LUTLDPC_FUSED_SIG; X template void launch_fused<2, 3> LUTLDPC_FUSED_SIG; \
    X template hipError_t preload_fused<2, 0>(); X template hipError_t preload_fused<2, 1>(); X template hipError_t preload_fused<2, 2>(); X template hipError_t preload_fused<2, 3>(); \
    X template hipError_t preload_vn_fast<TT_VAR, 1>(); X template hipError_t preload_vn_fast<TT_VAR, 2>(); X template hipError_t preload_vn_fast<TT_DEC, 2>(); \
    X template hipError_t preload_cn_fast<2>();

}  // namespace lutldpc
