// kernels_fast.hpp -- specialised gfx950 kernels for the shapes ber_sim actually produces:
//   * min-sum check nodes (src/LDPC_Code_LUT.cpp:355-402) with the check's rows held in
//     registers and byte-parallel (SWAR) arithmetic on the four frames packed in each dword;
//   * variable / decision nodes whose tree is the balanced binary tree of
//     LUT_Tree_Node::gen_bin_balanced_tree (src/LUT_Tree.cpp:200-237) -- the only shape
//     ber_sim designs in its auto modes (src/LDPC_BER_Sim.cpp:487-489) -- evaluated as
//     straight-line code with every shared sub-expression computed once.
// Everything else (file trees, CHKTREE checks, odd alphabets) goes to kernels_generic.hpp.
//
// Roofline: both passes are HBM-bound streaming of 256-byte rows (measured ceiling for this
// in-place gather/scatter pattern on MI355X: ~5.0-5.2 TB/s, tests/microbench/rows.hip).
// Algorithmic bytes per launch: check pass 2*E*B, variable pass (2*E + N)*B (+N*B when the
// hard decisions are written for the early-termination test).
#pragma once
#include "kernels_common.hpp"
#include "lut_program.hpp"

#include <utility>

namespace lutldpc {

constexpr int kFastMaxTables = 32;     // LUT nodes of one balanced tree (degree <= 33)
constexpr int kFastTableStride = 256;  // bytes per table slot in LDS (byte tables; nibble tables use the first half)

struct FastParams {
    int32_t n_nodes, node_off, nodes_per_wave, waves_per_group;
    int32_t idx_off;       // offset of the class in the dense index blob (see decoder.hip: build_fast_index)
    int32_t G, E, N;
    int32_t nz;            // sign threshold (see PassParams)
    int32_t shift_msg;     // log2 of the message alphabet feeding the tables (label = a | b << shift)
    int32_t check, write_hard;
    int32_t deg;
    int32_t n_tables;
    int32_t tab_off[kFastMaxTables];    // byte offsets into the table blob, canonical node order
    int32_t tab_len[kFastMaxTables];
    int32_t tab_shift[kFastMaxTables];  // log2 alphabet of each table's first child
    int32_t nib;                        // 1: tables staged as nibbles (two entries per byte), see lut4
};

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
template <int DEG, int UNR, int PACK>
__global__ __launch_bounds__(256) void cn_minsum_fast_kernel(
    FastParams P, uint8_t *__restrict__ msgs, const uint32_t *__restrict__ state_w, uint32_t *__restrict__ vfail_w,
    const int32_t *__restrict__ fast_idx)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));    // wave-uniform -> SGPR
    const int g = wave / P.waves_per_group;
    if (g >= P.G) return;
    const int chunk = wave - g * P.waves_per_group;
    uint32_t amask[PACK];
    if (load_active<PACK>(state_w, g, lane, amask)) return;
    const uint32_t smask = pack_masks<PACK>(amask);
    uint8_t *base = msgs + (size_t)g * (size_t)P.E * kRowBytes + lane * 4;
    const int32_t *edges = fast_idx + P.idx_off;
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

    for (int i = first; i < last; i += UNR) {
        uint32_t x[UNR][DEG];
        int e[UNR][DEG];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int ii = (i + u < last) ? i + u : last - 1;
#pragma unroll
            for (int k = 0; k < DEG; k++) e[u][k] = edges[(size_t)ii * DEG + k];
        }
#pragma unroll
        for (int u = 0; u < UNR; u++)
#pragma unroll
            for (int k = 0; k < DEG; k++) x[u][k] = *reinterpret_cast<const uint32_t *>(base + (size_t)e[u][k] * kRowBytes);
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            if (i + u >= last) break;
            uint32_t min1 = LOW, min2 = LOW, spp = 0;
            uint32_t pk[DEG];                                     // magnitude | positive flag (bit sbit)
#pragma unroll
            for (int k = 0; k < DEG; k++) {
                const uint32_t xh = x[u][k];
                const uint32_t pos = xh & SB;
                const uint32_t pm = pos - (pos >> sbit);                  // LOW where positive
                const uint32_t mag = (xh ^ pm ^ LOW) & LOW;
                spp ^= pos;
                const uint32_t g1 = ((mag | SB) - min1) & SB;             // mag >= min1
                const uint32_t k1 = g1 - (g1 >> sbit);
                const uint32_t lo = bfi(k1, min1, mag);
                const uint32_t hi = mag ^ min1 ^ lo;
                const uint32_t g2 = ((min2 | SB) - hi) & SB;              // min2 >= hi
                const uint32_t k2 = g2 - (g2 >> sbit);
                min2 = bfi(k2, hi, min2);
                min1 = lo;
                pk[k] = mag | pos;
            }
            const uint32_t tn = spp ^ odd;                                    // parity of the negative inputs (bit sbit)
            if (P.check) failw |= tn >> sbit;
#pragma unroll
            for (int k = 0; k < DEG; k++) {
                const uint32_t mag = pk[k] & LOW, pos = pk[k] & SB;
                const uint32_t eq = ~(((mag ^ min1) | SB) - ONE) & SB;   // this edge holds the minimum
                const uint32_t ke = eq - (eq >> sbit);
                const uint32_t m = bfi(ke, min2, min1);
                const uint32_t po = tn ^ pos;                             // extrinsic sign: positive flag
                const uint32_t nf = po ^ SB;
                const uint32_t kn = nf - (nf >> sbit);                    // LOW where the result is negative
                const uint32_t r = (m ^ kn) | po;                         // positive: nz+m ; negative: nz-1-m
                *reinterpret_cast<uint32_t *>(base + (size_t)e[u][k] * kRowBytes) = bfi(smask, r, x[u][k]);
            }
        }
    }
    if (P.check) {
        uint32_t fail[PACK];
#pragma unroll
        for (int h = 0; h < PACK; h++) fail[h] = unpack_half<PACK>(failw, h);
        flag_frames<PACK>(vfail_w, g, lane, fail, amask);
    }
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

template <int N>
struct Bal {
    static constexpr BalShape<N> S = make_bal_shape<N>();
};

// One look-up for the four packed frames: label = a | b << sh, table slot `t` in LDS.
// A 256-entry byte table would span 64 dwords = every LDS bank twice, and two lanes of a 32-lane
// group reading different dwords of one bank cost an extra cycle (measured: 51 % of the LDS cycles of
// the degree-8 kernel were such 2-way conflicts).  Tables whose outputs fit 4 bits are therefore staged
// as NIBBLES: entry i in byte i>>1, low nibble for even i.  128 bytes = 32 dwords = one dword per bank:
// every access is conflict-free.  The nibble is selected for all four frames at once (SWAR).
template <bool NIB>
__device__ __forceinline__ uint32_t lut4(const uint8_t *lds_tab, int t, uint32_t a, uint32_t b, int sh) {
    const uint32_t L = a | (b << sh);
    const uint8_t *tb = lds_tab + t * kFastTableStride;
    if constexpr (NIB) {
        const uint32_t r0 = tb[(L >> 1) & 0x7Fu], r1 = tb[(L >> 9) & 0x7Fu], r2 = tb[(L >> 17) & 0x7Fu], r3 = tb[L >> 25];
        const uint32_t R = r0 | (r1 << 8) | (r2 << 16) | (r3 << 24);
        const uint32_t od = L & 0x01010101u;                    // odd entry -> high nibble
        const uint32_t M = (od << 4) - od;                      // 0x0F where odd
        return bfi(M, R >> 4, R) & 0x0F0F0F0Fu;
    } else {
        const uint32_t r0 = tb[L & 0xFFu], r1 = tb[(L >> 8) & 0xFFu], r2 = tb[(L >> 16) & 0xFFu], r3 = tb[L >> 24];
        return r0 | (r1 << 8) | (r2 << 16) | (r3 << 24);
    }
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

template <int N, int J, int K, bool NIB>
__device__ __forceinline__ void bal_node_variant(const uint32_t *in, uint32_t *v, const uint8_t *tab, int sh) {
    constexpr BalShape<N> S = make_bal_shape<N>();
    constexpr int L = S.left[J], R = S.right[J], sl = S.size[L];
    constexpr int kl = K < sl ? K : sl, kr = K > sl ? K - sl : 0;
    constexpr int dst = S.off[J] + K;
    v[dst] = lut4<NIB>(tab, J, bal_child<N, L, kl>(in, v), bal_child<N, R, kr>(in, v), sh);
}

template <int N, int J, bool NIB, int... Ks>
__device__ __forceinline__ void bal_node_all(const uint32_t *in, uint32_t *v, const uint8_t *tab, int sh, std::integer_sequence<int, Ks...>) {
    (bal_node_variant<N, J, Ks, NIB>(in, v, tab, sh), ...);
}
// VAR: all variants of every node; DEC (ALL = false): only the unshifted variant K = size
template <int N, bool ALL, bool NIB, int... Js>
__device__ __forceinline__ void bal_all_nodes(const uint32_t *in, uint32_t *v, const uint8_t *tab, int sh, std::integer_sequence<int, Js...>) {
    if constexpr (ALL) (bal_node_all<N, Js, NIB>(in, v, tab, sh, std::make_integer_sequence<int, make_bal_shape<N>().size[N + Js] + 1>{}), ...);
    else (bal_node_variant<N, Js, make_bal_shape<N>().size[N + Js], NIB>(in, v, tab, sh), ...);
}

// Variable-node (KIND = TT_VAR) / decision (TT_DEC) pass for degree-DV nodes with balanced trees.
// Tables: slots 0..NI-1 = internal nodes in creation order, slot NI = root.
// DV == 1 (VAR only): ROOT(CHA), the build's degree-1 extension.
template <int DV, int KIND, bool CHECK, int PACK, bool NIB>
__global__ __launch_bounds__(256) void vn_balanced_fast_kernel(
    FastParams P, uint8_t *__restrict__ msgs, const uint8_t *__restrict__ cha, uint8_t *__restrict__ hard,
    const uint32_t *__restrict__ state_w, uint32_t *__restrict__ vfail_w, const uint8_t *__restrict__ tables,
    const int32_t *__restrict__ fast_idx)
{
    constexpr int N = (KIND == TT_DEC) ? DV : DV - 1;          // message leaves
    constexpr int NI = N > 1 ? N - 1 : 0;
    constexpr int NB = N > 1 ? N : 2;                          // shape used for array sizes when N <= 1
    __shared__ __attribute__((aligned(16))) uint8_t lds_tab[(NI + 1) * kFastTableStride];
    // stage the class tables (canonical order, fixed 256-byte slots)
    // (table offsets and lengths are multiples of 4: lut_program.hpp pads every table)
    for (int t = 0; t <= NI; t++) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(tables + P.tab_off[t]);
        const int i = threadIdx.x;
        if constexpr (NIB) {
            // 256 byte entries -> 128 bytes of nibbles: thread i packs entries 8i..8i+7 into one dword
            if (i < P.tab_len[t] / 8) {
                const uint32_t lo = src[2 * i], hi = src[2 * i + 1];
                const uint32_t pl = (lo & 0x0Fu) | ((lo >> 4) & 0xF0u) | ((lo >> 8) & 0xF00u) | ((lo >> 12) & 0xF000u);
                const uint32_t ph = (hi & 0x0Fu) | ((hi >> 4) & 0xF0u) | ((hi >> 8) & 0xF00u) | ((hi >> 12) & 0xF000u);
                reinterpret_cast<uint32_t *>(lds_tab + t * kFastTableStride)[i] = pl | (ph << 16);
            }
        } else {
            if (i < P.tab_len[t] / 4) reinterpret_cast<uint32_t *>(lds_tab + t * kFastTableStride)[i] = src[i];
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));    // wave-uniform -> SGPR
    const int g = wave / P.waves_per_group;
    if (g >= P.G) return;
    const int chunk = wave - g * P.waves_per_group;
    uint32_t amask[PACK];
    if (load_active<PACK>(state_w, g, lane, amask)) return;
    const uint32_t smask = pack_masks<PACK>(amask);
    const int32_t *vtab = fast_idx + P.idx_off;                 // dense [n_nodes][2] = {node id, first edge}
    uint8_t *mbase = msgs + (size_t)g * (size_t)P.E * kRowBytes + lane * 4;
    const uint8_t *cbase = cha + (size_t)g * (size_t)P.N * kRowBytes + lane * 4;
    uint8_t *hbase = hard + (size_t)g * (size_t)P.N * kRowBytes + lane * 4;
    const int first = chunk * P.nodes_per_wave;
    int last = first + P.nodes_per_wave;
    if (last > P.n_nodes) last = P.n_nodes;
    const int sh = P.shift_msg, shr = P.tab_shift[NI];
    uint32_t fail[PACK];
#pragma unroll
    for (int h = 0; h < PACK; h++) fail[h] = 0;

    for (int i = first; i < last; i++) {
        const int v = vtab[2 * (size_t)i], e0 = vtab[2 * (size_t)i + 1];
        uint32_t raw[DV + 1];
#pragma unroll
        for (int k = 0; k < DV; k++) raw[k] = *reinterpret_cast<const uint32_t *>(mbase + (size_t)(e0 + k) * kRowBytes);
        raw[DV] = *reinterpret_cast<const uint32_t *>(cbase + (size_t)v * kRowBytes);
        uint32_t out[DV], bits[PACK];
#pragma unroll
        for (int h = 0; h < PACK; h++) {
            uint32_t in[DV + 1];
#pragma unroll
            for (int k = 0; k <= DV; k++) in[k] = unpack_half<PACK>(raw[k], h);
            const uint32_t ch = in[DV];
            constexpr BalShape<NB> SH = make_bal_shape<NB>();
            constexpr int top_off = SH.off[SH.top - NB];
            uint32_t val[(N > 1 ? SH.total : 1)];
            if constexpr (N > 1) bal_all_nodes<N, KIND == TT_VAR, NIB>(in, val, lds_tab, sh, std::make_integer_sequence<int, NI>{});
            if constexpr (KIND == TT_DEC) {
                uint32_t top;
                if constexpr (N > 1) top = val[top_off + N];
                else top = in[0];
                bits[h] = swar_lt(lut4<NIB>(lds_tab, NI, top, ch, shr), 1u);     // src/LDPC_Code_LUT.cpp:342
            } else {
                uint32_t neg_ref = 0;
#pragma unroll
                for (int o = 0; o < DV; o++) {
                    uint32_t r;
                    if constexpr (N == 0) {
                        // degree 1: the only leaf is the channel label (a one-input table: label = ch)
                        r = lut4<NIB>(lds_tab, 0, ch, 0u, 0);
                    } else {
                        uint32_t top;
                        if constexpr (N > 1) top = val[top_off + o];
                        else top = in[o == 0 ? 1 : 0];                     // N == 1: the other message
                        r = lut4<NIB>(lds_tab, NI, top, ch, shr);
                    }
                    if (PACK == 2 && h == 1) out[o] |= r << 4; else out[o] = r;
                    if (CHECK) {
                        const uint32_t ng = swar_lt(r, (uint32_t)P.nz);
                        if (o == 0) neg_ref = ng; else fail[h] |= ng ^ neg_ref;
                    }
                }
                bits[h] = neg_ref;
            }
        }
        if constexpr (KIND == TT_DEC) {
            store_row_masked<PACK>(reinterpret_cast<uint32_t *>(hbase + (size_t)v * kRowBytes), pack_halves<PACK>(bits), smask);
        } else {
#pragma unroll
            for (int o = 0; o < DV; o++) *reinterpret_cast<uint32_t *>(mbase + (size_t)(e0 + o) * kRowBytes) = bfi(smask, out[o], raw[o]);
            if (CHECK && P.write_hard)
                store_row_masked<PACK>(reinterpret_cast<uint32_t *>(hbase + (size_t)v * kRowBytes), pack_halves<PACK>(bits), smask);
        }
    }
    if (KIND == TT_VAR && CHECK) flag_frames<PACK>(vfail_w, g, lane, fail, amask);
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
    // optional nibble staging of 256-entry tables (conflict-free LDS reads, more VALU; measured SLOWER on
    // MI355X for the DVB-S2 classes, so off unless LUTLDPC_NIB_TABLES=1): needs 4-bit outputs
    bool want_nib = false, can_nib = true;
    for (size_t j = 0; j < canon.size(); j++) {
        if (fp.P.tab_len[j] > 128) want_nib = true;
        if (canon[j]->K > 16 || (fp.P.tab_len[j] & 7)) can_nib = false;
    }
    const char *env = getenv("LUTLDPC_NIB_TABLES");          // minimum degree that gets nibble tables (0 = never)
    const int nib_min_deg = env ? atoi(env) : 0;
    fp.P.nib = (nib_min_deg > 0 && d >= nib_min_deg && want_nib && can_nib) ? 1 : 0;
    fp.P.deg = d; fp.P.node_off = node_off; fp.P.n_nodes = n_nodes;
    fp.ok = true;
    return fp;
}

template <int KIND, bool CHECK, int PACK, int DV>
inline void launch_vn_fast_one(hipStream_t s, const FastParams &P, uint8_t *msgs, const uint8_t *cha, uint8_t *hard, const uint32_t *state_w,
                               uint32_t *vfail_w, const uint8_t *tables, const int32_t *fast_idx) {
    const int waves = P.waves_per_group * P.G;
    if (P.nib)
        hipLaunchKernelGGL((vn_balanced_fast_kernel<DV, KIND, CHECK, PACK, true>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, P, msgs, cha, hard, state_w, vfail_w,
                           tables, fast_idx);
    else
        hipLaunchKernelGGL((vn_balanced_fast_kernel<DV, KIND, CHECK, PACK, false>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, P, msgs, cha, hard, state_w, vfail_w,
                           tables, fast_idx);
}

template <int KIND, bool CHECK, int PACK, int... DVs>
inline bool dispatch_vn_fast(int deg, std::integer_sequence<int, DVs...>, hipStream_t s, const FastParams &P, uint8_t *msgs, const uint8_t *cha,
                             uint8_t *hard, const uint32_t *state_w, uint32_t *vfail_w, const uint8_t *tables, const int32_t *fast_idx) {
    bool done = false;
    ((deg == DVs + 1 ? (launch_vn_fast_one<KIND, CHECK, PACK, DVs + 1>(s, P, msgs, cha, hard, state_w, vfail_w, tables, fast_idx), done = true) : false), ...);
    return done;
}

constexpr int kFastMaxDeg = 20;      // variable / decision nodes
constexpr int kFastMaxCnDeg = 32;    // check nodes

// launch one class; returns false when the degree has no instantiation
template <int KIND, int PACK>
inline bool launch_vn_fast(hipStream_t s, FastParams P, int G, int nz, int check, int write_hard, int nodes_per_wave, uint8_t *msgs, const uint8_t *cha,
                           uint8_t *hard, const uint32_t *state_w, uint32_t *vfail_w, const uint8_t *tables, const int32_t *fast_idx, int E, int N) {
    P.G = G; P.E = E; P.N = N; P.nz = nz; P.check = check; P.write_hard = write_hard;
    P.nodes_per_wave = nodes_per_wave;
    P.waves_per_group = (P.n_nodes + nodes_per_wave - 1) / nodes_per_wave;
    constexpr auto seq = std::make_integer_sequence<int, kFastMaxDeg>{};
    if (KIND == TT_VAR && check) return dispatch_vn_fast<KIND, true, PACK>(P.deg, seq, s, P, msgs, cha, hard, state_w, vfail_w, tables, fast_idx);
    return dispatch_vn_fast<KIND, false, PACK>(P.deg, seq, s, P, msgs, cha, hard, state_w, vfail_w, tables, fast_idx);
}

template <int PACK, int DEG>
inline void launch_cn_fast_one(hipStream_t s, const FastParams &P, uint8_t *msgs, const uint32_t *state_w, uint32_t *vfail_w, const int32_t *fast_idx) {
    // checks in flight per wave: more = more loads outstanding per wave, fewer = fewer VGPRs = more waves
    constexpr int UNR = DEG <= 4 ? 4 : DEG <= 10 ? 2 : 1;
    static const int unr_env = getenv("LUTLDPC_CN_UNR") ? atoi(getenv("LUTLDPC_CN_UNR")) : 0;
    const int waves = P.waves_per_group * P.G;
    if (unr_env == 1 || UNR == 1)
        hipLaunchKernelGGL((cn_minsum_fast_kernel<DEG, 1, PACK>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, P, msgs, state_w, vfail_w, fast_idx);
    else
        hipLaunchKernelGGL((cn_minsum_fast_kernel<DEG, UNR, PACK>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, P, msgs, state_w, vfail_w, fast_idx);
}
template <int PACK, int... Ds>
inline bool dispatch_cn_fast(int deg, std::integer_sequence<int, Ds...>, hipStream_t s, const FastParams &P, uint8_t *msgs, const uint32_t *state_w,
                             uint32_t *vfail_w, const int32_t *fast_idx) {
    bool done = false;
    ((deg == Ds + 1 ? (launch_cn_fast_one<PACK, Ds + 1>(s, P, msgs, state_w, vfail_w, fast_idx), done = true) : false), ...);
    return done;
}

// min-sum: one launch per degree class
template <int PACK>
inline bool launch_cn_fast(hipStream_t s, int deg, int n_nodes, int idx_off, int G, int E, int nz, int check, int nodes_per_wave, uint8_t *msgs,
                           const uint32_t *state_w, uint32_t *vfail_w, const int32_t *fast_idx) {
    if (!is_pow2(nz) || nz > 64 || deg < 2 || deg > kFastMaxCnDeg) return false;
    FastParams P{};
    P.n_nodes = n_nodes; P.idx_off = idx_off; P.G = G; P.E = E; P.nz = nz; P.check = check; P.deg = deg;
    P.nodes_per_wave = nodes_per_wave;
    P.waves_per_group = (n_nodes + nodes_per_wave - 1) / nodes_per_wave;
    return dispatch_cn_fast<PACK>(deg, std::make_integer_sequence<int, kFastMaxCnDeg>{}, s, P, msgs, state_w, vfail_w, fast_idx);
}

}  // namespace lutldpc
