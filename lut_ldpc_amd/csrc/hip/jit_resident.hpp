// jit_resident.hpp -- generator of the LDS-RESIDENT decode kernel (one kernel per code, compiled with hiprtc at the first decode).
//
// Replaces, for codes whose edge messages fit the LDS of a compute unit, the whole of LDPC_Code_LUT::lut_decode
// (src/LDPC_Code_LUT.cpp:259-353) for a batch: initial syndrome test (:275-279), edge initialisation (:284-289), all
// iterations (:301-338: chk_update_minsum / chk_update_lut, var_update_lut, syndrome_check(Nq, b)), decision (:340-344) and the
// final syndrome test (:346-349) -- ONE launch, no HBM traffic between the channel labels and the decided bits.
//
// Work split: a workgroup of NT threads owns S *sets* (set = the 4*PACK frames one lane of a row holds: one dword per edge);
// the E edge dwords of every set live in LDS for the whole decode.  Variable-node items (set, node) are owned by ONE thread for
// the whole decode -- its channel dword, decided-bit dword and LDS address stay in registers -- check-node items are dealt
// round-robin every pass.  Per iteration two (fixed work) or three (exit test on) workgroup barriers.  A set whose frames have
// all left through the exit test costs nothing any more, a workgroup whose sets are all done returns: the retire grain of
// parity_check_iter = true is 8 frames, not a 512-frame group.
// The node updates are the node programs of lut_program.hpp emitted as straight-line code (like jit.hpp), the min-sum is
// res_minsum (kernels_resident.hpp, the SWAR arithmetic of cn_minsum_body).
#pragma once
#include "jit.hpp"
#include "lut_program.hpp"

#include <algorithm>
#include <sstream>
#include <string>
#include <vector>

namespace lutldpc {

static const char *const kResidentHeaderText =
#include "kernels_resident.inc"
    ;

struct ResidentArgs {              // kernel arguments (by value: 112 bytes)
    const uint8_t *cha, *msg0;     // label rows [G][N][256 B]
    uint8_t *hard;                 // decided-bit rows
    uint8_t *state;                // per-frame state bytes
    int32_t *iters;                // per-frame iteration codes
    const uint8_t *tables;         // the decoder's table blob
    const int32_t *idx;            // the decoder's dense index blob (build_fast_index)
    int32_t n_sets;                // 64 * frame groups
    int32_t max_iters, psc, pisc;
    int32_t B;                     // frames of the batch (frame-major I/O: frames beyond it read as label 0)
    // frame-major I/O (the C-ABI's own layout, [B][N] bytes): when fm_cha is set the kernel reads the labels and writes the decided
    // bits there itself -- a thread's eight frames of a node are eight byte accesses, 64 lanes of consecutive nodes one 64-byte
    // segment per frame -- and the three transposes around the decode disappear (9 % of a (6,32) N=2048 step)
    const uint8_t *fm_cha, *fm_msg0;
    uint8_t *fm_bits;
    int32_t lim_cha, lim_msg;      // labels are clamped to the alphabets like the transposes do
};
static_assert(sizeof(ResidentArgs) <= 128, "kernel arguments stay small (DESIGN.md section 7.1)");

struct ResidentClass { int deg = 0, n = 0, idx_off = 0, nidx_off = 0; };
struct ResidentSpec {
    int pack = 2, N = 0, E = 0, S = 1, NT = 1024, I = 0, nq_cha = 16, min_lut = 1;
    int flag_reduce = 1;                                     // exit-test flags ORed over the wave before the LDS atomic (0: per-lane atomics)
    int waves_eu = 0;                                        // amdgpu_waves_per_eu lower bound (0: compiler's choice)
    int cn_persistent = 0;                                   // check items keep their edge addresses in registers for the whole decode (small regular codes)
    int xcd = 1;                                             // XCD-aware set order (0: workgroup b takes sets b * S ...)
    int U = 0;                                               // frames per trip of the look-up loops (0: 2 up to degree 8, else 1)
    std::vector<int> nq_msg, iter_set;                       // per iteration
    std::vector<ResidentClass> vcls, ccls;
    // [set][class]: program (null: the set has none of this kind) and {offset, bytes} of the class blob in the table blob
    std::vector<std::vector<const Program *>> var_prog, dec_prog, chk_prog;
    std::vector<std::vector<std::pair<int, int>>> var_tab, dec_tab, chk_tab;
};

namespace resident_detail {

inline std::string S_(long long v) { return std::to_string(v); }

// label expression of one look-up of a variable / decision program: operands are named values (one frame each)
inline bool var_label(const Op &op, const std::vector<std::string> &name, std::string &label, std::string &err) {
    bool all_pow2 = true;
    for (int c = 0; c < op.nchild; c++) all_pow2 = all_pow2 && jit_pow2(op.mult[c]) && jit_pow2(op.childK[c]);
    label.clear();
    for (int c = 0; c < op.nchild; c++) {
        const std::string &x = name[(size_t)op.child[c]];
        if (x.empty()) { err = "operand read before it is written"; return false; }
        if (c == 0) label = op.mult[c] == 1 ? x : "(" + x + " * " + S_(op.mult[c]) + "u)";
        else if (all_pow2) label = "lshl_or(" + x + ", " + S_(__builtin_ctz(op.mult[c])) + ", " + label + ")";
        else label = "(" + label + " + " + x + " * " + S_(op.mult[c]) + "u)";
    }
    return true;
}

// The frame loop of a variable (TT_VAR) / decision (TT_DEC) item: inputs raw[0..deg] (raw[deg] = channel dword), tables at
// `tb` (uint8_t *, LDS); results out[0..deg-1] (VAR) / hardw (DEC).  U frames per trip (independent: ILP for the LDS latency).
inline bool emit_var_frames(std::ostringstream &o, const Program &prog, int kind, int deg, int U, const std::string &ind, std::string &err) {
    if (prog.n_in != deg + 1) { err = "unexpected input count"; return false; }
    o << ind << "#pragma unroll 1\n" << ind << "for (int fs = 0; fs < F * BITS; fs += " << U << " * BITS) {\n";
    for (int u = 0; u < U; u++) {
        const std::string sfx = "_" + S_(u), sh = u ? "(uint32_t)(fs + " + S_(u) + " * BITS)" : "(uint32_t)fs";
        std::vector<std::string> name((size_t)std::max(prog.n_slots, prog.n_in) + 1);
        for (int k = 0; k <= deg; k++) {
            o << ind << "    const uint32_t i" << k << sfx << " = __builtin_amdgcn_ubfe(raw[" << k << "], " << sh << ", (uint32_t)BITS);\n";
            name[(size_t)k] = "i" + S_(k) + sfx;
        }
        for (size_t j = 0; j < prog.ops.size(); j++) {
            const Op &op = prog.ops[j];
            if (op.kind != 0) { err = "check-type look-up in a variable program"; return false; }
            if ((size_t)op.dst >= name.size()) name.resize((size_t)op.dst + 1);
            std::string label;
            if (!var_label(op, name, label, err)) return false;
            const std::string t = "t" + S_((long long)j) + sfx;
            o << ind << "    const uint32_t " << t << " = tb[" << op.tab_off << "u + " << label << "];\n";
            name[(size_t)op.dst] = t;
            if (op.out_idx >= 0) {
                if (kind == TT_DEC) o << ind << "    hardw = lshl_or(" << t << " < 1u ? 1u : 0u, " << sh << ", hardw);\n";      // src/LDPC_Code_LUT.cpp:342
                else o << ind << "    out[" << op.out_idx << "] = lshl_or(" << t << ", " << sh << ", out[" << op.out_idx << "]);\n";
            }
        }
    }
    o << ind << "}\n";
    return true;
}

// The frame loop of a CHKTREE item (min_lut = false, src/LUT_Tree.cpp:792-807,420-445): inputs x[0..deg-1], outputs out[0..deg-1];
// par_w collects the parity of the incoming signs (one bit per element) when `chk`.
inline bool emit_chk_frames(std::ostringstream &o, const Program &prog, int deg, const std::string &ind, std::string &err) {
    if (prog.kind != TT_CHK || prog.n_in != deg || prog.n_out != deg) { err = "not a check program of this degree"; return false; }
    o << ind << "#pragma unroll 1\n" << ind << "for (int fs = 0; fs < F * BITS; fs += BITS) {\n";
    std::vector<std::string> name((size_t)std::max(prog.n_slots, prog.n_in) + 1), smname(name.size());
    std::vector<int> sm_of(name.size(), 0);
    for (int k = 0; k < deg; k++) {
        o << ind << "    const uint32_t i" << k << " = __builtin_amdgcn_ubfe(x[" << k << "], (uint32_t)fs, (uint32_t)BITS);\n";
        name[(size_t)k] = "i" + S_(k);
    }
    o << ind << "    if (chk) {\n" << ind << "        uint32_t par = 0;\n";
    for (int k = 0; k < deg; k++) o << ind << "        par ^= i" << k << " < nz ? 1u : 0u;\n";
    o << ind << "        par_w = lshl_or(par, fs, par_w);\n" << ind << "    }\n";
    for (size_t j = 0; j < prog.ops.size(); j++) {
        const Op &op = prog.ops[j];
        if (op.kind != 1 && op.kind != 2) { err = "variable-type look-up in a check program"; return false; }
        if ((size_t)op.dst >= name.size()) { name.resize((size_t)op.dst + 1); sm_of.resize(name.size(), 0); smname.resize(name.size()); }
        if (op.kind == 2) {       // table over the children's full labels (lut_program.hpp: chk_full_label_program)
            bool p2 = true;
            for (int c = 0; c < op.nchild; c++) p2 = p2 && jit_pow2(op.mult[c]) && jit_pow2(op.childK[c]);
            std::string label;
            for (int c = 0; c < op.nchild; c++) {
                const std::string &xn = name[(size_t)op.child[c]];
                if (xn.empty()) { err = "operand read before it is written"; return false; }
                if (c == 0) label = op.mult[c] == 1 ? xn : "(" + xn + " * " + S_(op.mult[c]) + "u)";
                else if (p2) label = "lshl_or(" + xn + ", " + S_(__builtin_ctz(op.mult[c])) + ", " + label + ")";
                else label = "(" + label + " + " + xn + " * " + S_(op.mult[c]) + "u)";
            }
            const std::string t = "t" + S_((long long)j);
            o << ind << "    const uint32_t " << t << " = tc[" << op.tab_off << "u + " << label << "];\n";
            name[(size_t)op.dst] = t;
            if (op.out_idx >= 0) o << ind << "    out[" << op.out_idx << "] = lshl_or(" << t << ", fs, out[" << op.out_idx << "]);\n";
            continue;
        }
        bool all_pow2 = jit_pow2(op.half_len);
        for (int c = 0; c < op.nchild; c++) all_pow2 = all_pow2 && jit_pow2(op.mult[c]) && jit_pow2((uint32_t)op.childK[c] >> 1);
        std::string label, par;
        for (int c = 0; c < op.nchild; c++) {
            const size_t sl = op.child[c];
            const std::string &xn = name[sl];
            if (xn.empty()) { err = "operand read before it is written"; return false; }
            const int hh = op.childK[c] >> 1;
            if (sm_of[sl] != hh || smname[sl] != xn) {
                o << ind << "    const uint32_t n_" << xn << " = " << xn << " < " << hh << "u ? 1u : 0u, m_" << xn << " = n_" << xn << " ? " << hh - 1 << "u - " << xn
                  << " : " << xn << " - " << hh << "u;\n";
                sm_of[sl] = hh; smname[sl] = xn;
            }
            const std::string m = "m_" + xn, n = "n_" + xn;
            if (c == 0) label = op.mult[c] == 1 ? m : "(" + m + " * " + S_(op.mult[c]) + "u)";
            else if (all_pow2) label = "lshl_or(" + m + ", " + S_(__builtin_ctz(op.mult[c])) + ", " + label + ")";
            else label = "(" + label + " + " + m + " * " + S_(op.mult[c]) + "u)";
            par = c == 0 ? n : par + " ^ " + n;
        }
        const std::string idx = all_pow2 ? "lshl_or((" + par + ") ^ 1u, " + S_(__builtin_ctz(op.half_len)) + ", " + label + ")"
                                         : "(" + label + " + ((" + par + ") ? 0u : " + S_(op.half_len) + "u))";
        const std::string t = "t" + S_((long long)j);
        o << ind << "    const uint32_t " << t << " = tc[" << op.tab_off << "u + " << idx << "];\n";
        name[(size_t)op.dst] = t;
        if (op.out_idx >= 0) o << ind << "    out[" << op.out_idx << "] = lshl_or(" << t << ", fs, out[" << op.out_idx << "]);\n";
    }
    o << ind << "}\n";
    return true;
}

inline void emit_int_array(std::ostringstream &o, const std::string &name, const std::vector<int> &v) {
    o << "    static constexpr int " << name << "[" << std::max<size_t>(v.size(), 1) << "] = {";
    for (size_t i = 0; i < v.size(); i++) o << (i ? "," : "") << v[i];
    if (v.empty()) o << "0";
    o << "};\n";
}

}  // namespace resident_detail

// bytes of LDS the tables of the resident kernel need (all variable classes of one set + all check classes, maxima over the sets)
inline void resident_table_bytes(const ResidentSpec &R, std::vector<int> &tv_off, int &tv_bytes, std::vector<int> &tc_off, int &tc_bytes) {
    auto pad16 = [](int x) { return (x + 15) / 16 * 16; };
    tv_off.assign(R.vcls.size(), 0); tc_off.assign(R.ccls.size(), 0);
    tv_bytes = 0;
    for (size_t c = 0; c < R.vcls.size(); c++) {
        int m = 0;
        for (size_t s = 0; s < R.var_tab.size(); s++) { if (c < R.var_tab[s].size()) m = std::max(m, R.var_tab[s][c].second); if (c < R.dec_tab[s].size()) m = std::max(m, R.dec_tab[s][c].second); }
        tv_off[c] = tv_bytes; tv_bytes += pad16(m);
    }
    tc_bytes = 0;
    if (!R.min_lut)
        for (size_t c = 0; c < R.ccls.size(); c++) {
            int m = 0;
            for (size_t s = 0; s < R.chk_tab.size(); s++) if (c < R.chk_tab[s].size()) m = std::max(m, R.chk_tab[s][c].second);
            tc_off[c] = tc_bytes; tc_bytes += pad16(m);
        }
}
inline int resident_lds_bytes(const ResidentSpec &R) {
    std::vector<int> a, b; int tv, tc;
    resident_table_bytes(R, a, tv, b, tc);
    return R.S * R.E * 4 + tv + tc + 16 * R.S + 64;
}

inline bool jit_resident_source(const ResidentSpec &R, std::string &src, std::string &err)
{
    using namespace resident_detail;
    const int PACK = R.pack, BITS = 8 / PACK, S = R.S, NT = R.NT, E = R.E, N = R.N, I = R.I;
    const size_t n_sets_tree = R.var_prog.size();
    if (I < 1 || (int)R.nq_msg.size() != I || (int)R.iter_set.size() != I) { err = "bad iteration tables"; return false; }
    std::vector<int> tv_off, tc_off; int tv_bytes, tc_bytes;
    resident_table_bytes(R, tv_off, tv_bytes, tc_off, tc_bytes);
    bool msg_pow2 = true;
    for (int q : R.nq_msg) msg_pow2 = msg_pow2 && jit_pow2((uint32_t)q / 2);
    if (R.min_lut && !msg_pow2) { err = "min-sum needs message alphabets whose half is a power of two"; return false; }

    std::ostringstream o;
    o << kCommonHeaderText << "\n" << kResidentHeaderText << "\nusing namespace lutldpc;\n"
      << "struct ResidentArgs { const uint8_t *cha, *msg0; uint8_t *hard; uint8_t *state; int32_t *iters; const uint8_t *tables; const int32_t *idx;\n"
      << "                      int32_t n_sets, max_iters, psc, pisc, B; const uint8_t *fm_cha, *fm_msg0; uint8_t *fm_bits; int32_t lim_cha, lim_msg; };\n"
      << "extern \"C\" __global__ __launch_bounds__(" << NT << ") " << (R.waves_eu > 0 ? "__attribute__((amdgpu_waves_per_eu(" + S_(R.waves_eu) + ", 8))) " : "")
      << "void lutldpc_jit_pass(ResidentArgs A)\n{\n"
      << "    constexpr int PACK = " << PACK << ", BITS = " << BITS << ", F = 4 * PACK, S = " << S << ", NT = " << NT << ", E = " << E << ", N = " << N << ", I = " << I << ";\n"
      << "    constexpr uint32_t ONE = PACK == 2 ? 0x11111111u : 0x01010101u;\n"
      << "    constexpr uint32_t NZC = " << R.nq_cha / 2 << "u;\n"
      // one LDS block with the tables FIRST: their addresses are small constants that fold into the 16-bit offset field of
      // ds_read_u8 (behind 120 KB of messages every look-up paid a v_add_u32 for its table base: a third of the VALU work)
      << "    constexpr int TVB = " << std::max(tv_bytes, 16) << ", TCB = " << std::max(tc_bytes, 16) << ";\n"
      << "    __shared__ __attribute__((aligned(16))) uint32_t LDS_ALL[(TVB + TCB) / 4 + 2 * S + S * E];\n"
      << "    uint8_t *const TV = reinterpret_cast<uint8_t *>(LDS_ALL);\n    uint8_t *const TC = TV + TVB;\n"
      << "    uint32_t *const L_fail = LDS_ALL + (TVB + TCB) / 4, *const L_act = L_fail + S;\n"
      << "    uint32_t *const M = L_act + S;\n"
      << "    auto flag = [&](int s_, uint32_t f_) { " << (R.flag_reduce ? "res_flag<PACK>(L_fail, s_, f_);" : "if (f_) atomicOr(&L_fail[s_], f_);") << " };\n"
      << "    (void)I; (void)N; (void)TC;\n";
    // ---- per-iteration / per-set constants
    {
        std::vector<int> nz, dec_ok((size_t)n_sets_tree, 0);
        for (int q : R.nq_msg) nz.push_back(q / 2);
        nz.push_back(R.nq_msg.back() / 2);
        emit_int_array(o, "kNz", nz);
        emit_int_array(o, "kSet", R.iter_set);
    }
    const int tid_items_note = 0; (void)tid_items_note;
    // XCD-aware set order: workgroups are dealt round-robin to the 8 XCDs (one L2 each), and a workgroup reads / writes 4 * S bytes of
    // every label row -- a 64-byte sector of a row holds 16 sets.  Giving XCD k the contiguous sets [k * Q/8, (k+1) * Q/8) makes the
    // workgroups that share a sector neighbours in time on ONE L2 (measured before: 549 MB of fabric reads for 61 MB of labels).
    o << "    const int tid = threadIdx.x;\n    const int nb = (int)gridDim.x;\n"
      << "    const int bid = (" << (R.xcd ? "(nb & 7) == 0" : "false") << ") ? ((int)blockIdx.x & 7) * (nb >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;\n    const int q0 = bid * S;\n"
      << "    const int max_iters = A.max_iters;\n    const bool psc = A.psc != 0;\n"
      << "    // ---- which frames of the sets decode (frame_state_kernel mode 0 ran before: ST_ACTIVE / ST_PAD)\n"
      << "    if (tid < S) {\n        const int q = q0 + tid;\n        uint32_t am[PACK];\n"
      << "#pragma unroll\n        for (int h = 0; h < PACK; h++) am[h] = q < A.n_sets ? swar_zero_mask(reinterpret_cast<const uint32_t *>(A.state)[(size_t)q * PACK + h]) : 0u;\n"
      << "        L_act[tid] = pack_masks<PACK>(am) & ONE;\n        L_fail[tid] = 0u;\n    }\n"
      << "    __syncthreads();\n";

    o << "    // frame-major access of the eight (four) frames of set q at node v: element n of the dword <-> frame res_frame_of_element(n)\n"
      << "    auto fm_load = [&](const uint8_t *p, int q, int v, uint32_t lim) {\n        uint32_t w = 0u;\n#pragma unroll\n        for (int n = 0; n < F; n++) {\n"
      << "            const int f = q * F + res_frame_of_element<PACK>(n);\n            uint32_t x = f < A.B ? (uint32_t)p[(size_t)f * N + v] : 0u;\n"
      << "            x = x > lim ? lim : x;\n            w |= x << (n * BITS);\n        }\n        return w;\n    };\n"
      << "    auto fm_store = [&](uint8_t *p, int q, int v, uint32_t w) {\n#pragma unroll\n        for (int n = 0; n < F; n++) {\n"
      << "            const int f = q * F + res_frame_of_element<PACK>(n);\n            if (f < A.B) p[(size_t)f * N + v] = (uint8_t)((w >> (n * BITS)) & 1u);\n        }\n    };\n";
    // ---- table staging helper (all threads): bytes `len` (multiple of 4) from the blob to an LDS region
    o << "    auto stage = [&](uint8_t *dst, int off, int len) {\n"
      << "        const uint32_t *s4 = reinterpret_cast<const uint32_t *>(A.tables + off);\n"
      << "        for (int i = tid; i < (len + 3) / 4; i += NT) reinterpret_cast<uint32_t *>(dst)[i] = s4[i];\n    };\n";
    // per set: offsets / lengths of the class blobs
    auto emit_tabs = [&](const std::string &nm, const std::vector<std::vector<std::pair<int, int>>> &tabs, size_t ncls) {
        std::vector<int> off, len;
        for (size_t s = 0; s < n_sets_tree; s++)
            for (size_t c = 0; c < ncls; c++) {
                const bool have = s < tabs.size() && c < tabs[s].size();
                off.push_back(have ? tabs[s][c].first : 0); len.push_back(have ? tabs[s][c].second : 0);
            }
        emit_int_array(o, nm + "Off", off); emit_int_array(o, nm + "Len", len);
    };
    const size_t NVC = R.vcls.size(), NCC = R.ccls.size();
    emit_tabs("kVar", R.var_tab, NVC); emit_tabs("kDec", R.dec_tab, NVC);
    if (!R.min_lut) emit_tabs("kChk", R.chk_tab, NCC);
    {
        std::vector<int> a(tv_off), b(tc_off);
        emit_int_array(o, "kTvOff", a);
        if (!R.min_lut) emit_int_array(o, "kTcOff", b);
    }
    o << "    auto stage_var = [&](int set) { for (int c = 0; c < " << NVC << "; c++) stage(TV + kTvOff[c], kVarOff[set * " << NVC << " + c], kVarLen[set * " << NVC << " + c]); };\n"
      << "    auto stage_dec = [&](int set) { for (int c = 0; c < " << NVC << "; c++) stage(TV + kTvOff[c], kDecOff[set * " << NVC << " + c], kDecLen[set * " << NVC << " + c]); };\n";
    if (!R.min_lut)
        o << "    auto stage_chk = [&](int set) { for (int c = 0; c < " << NCC << "; c++) stage(TC + kTcOff[c], kChkOff[set * " << NCC << " + c], kChkLen[set * " << NCC << " + c]); };\n";

    // ---- persistent variable-node items: class c, round r
    // Items are dealt over the threads as ONE sequence of slots, heaviest class first: slot g = base(class) + item index belongs to
    // thread g mod NT, so the partly filled last round of a class is completed by the first items of the next one (a class of 52
    // degree-17 nodes otherwise occupies one wave for a whole pass while the others wait at the barrier).
    struct Item { int c, r, base; std::string sfx; };
    std::vector<Item> items;
    {
        std::vector<size_t> order(NVC);
        for (size_t c = 0; c < NVC; c++) order[c] = c;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return R.vcls[a].deg > R.vcls[b].deg; });
        int base = 0;
        for (size_t c : order) {
            const int cnt = S * R.vcls[c].n;
            for (int r = base / NT; cnt > 0 && r <= (base + cnt - 1) / NT; r++) items.push_back({(int)c, r, base, "_" + S_((long long)c) + "_" + S_(r)});
            base += cnt;
        }
    }
    if (items.size() > 48) { err = "more than 48 variable-node items per thread"; return false; }
    // registers a thread keeps per item for the whole decode: channel dword, decided-bit dword, LDS dword index of the node's first
    // edge (set * E + edge; negative: no item).  Set and node id are recomputed where they are needed (start and end only).
    for (auto &it : items) o << "    uint32_t cha" << it.sfx << " = 0u, hard" << it.sfx << " = 0u; int ma" << it.sfx << " = -1;\n";
    for (auto &it : items) {
        const ResidentClass &C = R.vcls[(size_t)it.c];
        o << "    auto sv" << it.sfx << " = [&]() { return (tid + " << it.r << " * NT - " << it.base << ") / " << C.n << "; };\n"
          << "    auto nd" << it.sfx << " = [&]() { const int it = tid + " << it.r << " * NT - " << it.base << "; return A.idx[" << C.idx_off << " + it - (it / " << C.n << ") * " << C.n << "]; };\n";
    }
    // init A: channel dwords, decided bits of the channel labels (src/LDPC_Code_LUT.cpp:275)
    for (auto &it : items) {
        const ResidentClass &C = R.vcls[(size_t)it.c];
        o << "    {\n        const int it = tid + " << it.r << " * NT - " << it.base << ";\n        if (it >= 0 && it < S * " << C.n << ") {\n"
          << "            const int s = it / " << C.n << ", j = it - s * " << C.n << ", q = q0 + s;\n"
          << "            const int32_t *vt = A.idx + " << C.idx_off << ";\n"
          << "            ma" << it.sfx << " = s * E + vt[" << C.n << " + j];\n"
          << "            if (q < A.n_sets) cha" << it.sfx << " = A.fm_cha ? fm_load(A.fm_cha, q, vt[j], (uint32_t)A.lim_cha)\n"
          << "                                               : *reinterpret_cast<const uint32_t *>(A.cha + ((size_t)(q >> 6) * N + (size_t)vt[j]) * kRowBytes + (q & 63) * 4);\n"
          << "            hard" << it.sfx << " = res_lt<PACK>(cha" << it.sfx << ", NZC);\n"
          << "        }\n    }\n";
    }
    // (cn_persistent) a check item's LDS addresses (set * E + edge id) stay in registers too: a regular code of small degrees then
    // reads NO index from global memory inside the iteration loop ((3,6) N=10000: five items x six addresses per thread)
    struct CItem { size_t c; int r; std::string sfx; };
    std::vector<CItem> citems;
    // check items: one sequence of slots over all check classes as well (widest checks first)
    std::vector<int> cbase(NCC, 0);
    std::vector<size_t> corder(NCC);
    {
        for (size_t c = 0; c < NCC; c++) corder[c] = c;
        std::stable_sort(corder.begin(), corder.end(), [&](size_t a, size_t b) { return R.ccls[a].deg > R.ccls[b].deg; });
        int base = 0;
        for (size_t c : corder) { cbase[c] = base; base += S * R.ccls[c].n; }
    }
    // first item of class c this thread handles: slots tid, tid + NT, ... at or beyond the class base
    auto cn_loop = [&](size_t c) {
        const int b = cbase[c];
        return "for (int it = tid + ((" + S_(b) + " - tid + NT - 1) / NT) * NT - " + S_(b) + "; it < S * " + S_(R.ccls[c].n) + "; it += NT) {\n";
    };
    if (R.cn_persistent && R.min_lut) {
        for (size_t c : corder) {
            const int cnt = S * R.ccls[c].n;
            for (int r = cbase[c] / NT; cnt > 0 && r <= (cbase[c] + cnt - 1) / NT; r++) citems.push_back({c, r, "_" + S_((long long)c) + "_" + S_(r)});
        }
        for (auto &ci : citems) {
            const ResidentClass &C = R.ccls[ci.c];
            o << "    int cs" << ci.sfx << " = -1, ce" << ci.sfx << "[" << C.deg << "];\n    {\n        const int it = tid + " << ci.r << " * NT - " << cbase[ci.c] << ";\n"
              << "        if (it >= 0 && it < S * " << C.n << ") {\n            const int s = it / " << C.n << ", j = it - s * " << C.n << ";\n            cs" << ci.sfx << " = s;\n"
              << "#pragma unroll\n            for (int k = 0; k < " << C.deg << "; k++) ce" << ci.sfx << "[k] = s * E + A.idx[" << C.idx_off << " + j + k * " << C.n << "];\n        }\n    }\n";
        }
    }
    // ---- syndrome over the decided bits (src/LDPC_Code_LUT.cpp:455-469): bits exchanged through M[s * E + node]
    o << "    auto put_hard = [&]() {\n";
    for (auto &it : items) o << "        if (ma" << it.sfx << " >= 0) M[sv" << it.sfx << "() * E + nd" << it.sfx << "()] = hard" << it.sfx << ";\n";
    o << "    };\n    auto syndrome = [&]() {\n";
    for (size_t c : corder) {
        const ResidentClass &C = R.ccls[c];
        o << "        " << cn_loop(c)
          << "            const int s = it / " << C.n << ", j = it - s * " << C.n << ";\n"
          << "            const uint32_t a = L_act[s];\n            if (!a) continue;\n"
          << "            const int32_t *nd = A.idx + " << C.nidx_off << " + j;\n"
          << "            uint32_t acc = 0u;\n#pragma unroll 8\n            for (int k = 0; k < " << C.deg << "; k++) acc ^= M[s * E + nd[k * " << C.n << "]];\n"
          << "            flag(s, acc & a);\n        }\n";
    }
    o << "    };\n";
    // exit test of the sets: frames that are active and did not fail leave with iteration code `value` (state `st`);
    // returns (workgroup-uniform) whether any frame of the workgroup is still active; includes a barrier
    o << "    auto retire = [&](int value, uint32_t st) -> bool {\n"
      << "        uint32_t a = 0u;\n        if (tid < S) {\n            a = L_act[tid];\n            const uint32_t f = L_fail[tid], gone = a & ~f;\n"
      << "            if (gone) {\n                const size_t f0 = (size_t)(q0 + tid) * F;\n#pragma unroll\n                for (int n = 0; n < F; n++)\n"
      << "                    if ((gone >> (n * BITS)) & 1u) { A.iters[f0 + res_frame_of_element<PACK>(n)] = value; A.state[f0 + res_frame_of_element<PACK>(n)] = (uint8_t)st; }\n            }\n"
      << "            a &= f;\n            L_act[tid] = a;\n            L_fail[tid] = 0u;\n        }\n"
      << "        return __syncthreads_or(a != 0u) != 0;\n    };\n";
    o << "    bool alive = true;\n"
      << "    if (A.pisc) {   // src/LDPC_Code_LUT.cpp:275-279\n        put_hard();\n        __syncthreads();\n        syndrome();\n        __syncthreads();\n"
      << "        alive = retire(0, ST_DONE_PISC);\n    }\n";
    // init B: every edge starts with its node's initial message (:284-289)
    o << "    if (alive) {\n";
    for (auto &it : items) {
        const ResidentClass &C = R.vcls[(size_t)it.c];
        o << "        if (ma" << it.sfx << " >= 0) {\n            const int q = q0 + sv" << it.sfx << "();\n"
          << "            const uint32_t m0 = q >= A.n_sets ? 0u : A.fm_msg0 ? fm_load(A.fm_msg0, q, nd" << it.sfx << "(), (uint32_t)A.lim_msg)\n"
          << "                              : *reinterpret_cast<const uint32_t *>(A.msg0 + ((size_t)(q >> 6) * N + (size_t)nd" << it.sfx << "()) * kRowBytes + (q & 63) * 4);\n"
          << "#pragma unroll\n            for (int k = 0; k < " << C.deg << "; k++) M[ma" << it.sfx << " + k] = m0;\n        }\n";
    }
    if (!R.min_lut) o << "        stage_chk(kSet[0]);\n";
    o << "    }\n    __syncthreads();\n";

    // ---- the check pass of one iteration
    o << "    auto cn_pass = [&](int ii, bool chk) {\n        const uint32_t nz = (uint32_t)kNz[ii];\n        (void)nz;\n";
    if (R.min_lut) {
        o << "        const int sbit = __builtin_ctz(nz);\n        const uint32_t SB = nz * ONE, LOW = SB - ONE;\n";
        for (auto &ci : citems) {                  // (cn_persistent: every check item of this thread, addresses in registers)
            const ResidentClass &C = R.ccls[ci.c];
            o << "        if (cs" << ci.sfx << " >= 0) {\n            const int s = cs" << ci.sfx << ";\n            const uint32_t a = L_act[s];\n            if (a) {\n"
              << "                const uint32_t am = res_mask<PACK>(a);\n";
            if (C.deg <= 16) {
                o << "                uint32_t x[" << C.deg << "], r[" << C.deg << "];\n"
                  << "#pragma unroll\n                for (int k = 0; k < " << C.deg << "; k++) x[k] = M[ce" << ci.sfx << "[k]];\n"
                  << "                const uint32_t tn = res_minsum<" << C.deg << ", PACK>(x, r, sbit, SB, LOW);\n"
                  << "                if (chk) flag(s, (tn >> sbit) & a);\n"
                  << "#pragma unroll\n                for (int k = 0; k < " << C.deg << "; k++) M[ce" << ci.sfx << "[k]] = bfi(am, r[k], x[k]);\n";
            } else {      // wide checks: two sweeps over LDS, the addresses stay in registers
                o << "                uint32_t max1 = 0u, max2 = 0u, spp = 0u;      // the two LARGEST complemented magnitudes = the two smallest magnitudes\n#pragma unroll\n                for (int k = 0; k < " << C.deg << "; k++) {\n"
                  << "                    const uint32_t xh = M[ce" << ci.sfx << "[k]];\n                    const uint32_t pos = xh & SB, mc = (xh ^ (pos - (pos >> sbit))) & LOW;\n"
                  << "                    spp ^= xh;\n                    const uint32_t g1 = xad(max1, LOW, mc) & SB, k1 = g1 - (g1 >> sbit);      // mc > max1\n"
                  << "                    const uint32_t hi = bfi(k1, mc, max1), lo = xor3(mc, max1, hi);\n"
                  << "                    const uint32_t g2 = xad(max2, LOW, lo) & SB, k2 = g2 - (g2 >> sbit);\n"
                  << "                    max2 = bfi(k2, lo, max2);\n                    max1 = hi;\n                }\n"
                  << "                const uint32_t tn = (spp ^ " << ((C.deg & 1) ? "SB" : "0u") << ") & SB;\n"
                  << "                if (chk) flag(s, (tn >> sbit) & a);\n"
                  << "#pragma unroll\n                for (int k = 0; k < " << C.deg << "; k++) {\n"
                  << "                    const uint32_t xh = M[ce" << ci.sfx << "[k]];\n                    const uint32_t pos = xh & SB, mc = (xh ^ (pos - (pos >> sbit))) & LOW;\n"
                  << "                    const uint32_t eq = ~(((mc ^ max1) | SB) - ONE) & SB, ke = eq - (eq >> sbit);\n"
                  << "                    const uint32_t sel = bfi(ke, max2, max1), po = (tn ^ xh) & SB, kp = po - (po >> sbit);\n"
                  << "                    M[ce" << ci.sfx << "[k]] = bfi(am, xor_or(sel, kp, po), xh);\n                }\n";
            }
            o << "            }\n        }\n";
        }
        for (size_t c : corder) {
            if (!citems.empty()) break;
            const ResidentClass &C = R.ccls[c];
            if (C.deg < 2) { err = "check of degree 1"; return false; }
            o << "        " << cn_loop(c)
              << "            const int s = it / " << C.n << ", j = it - s * " << C.n << ";\n"
              << "            const uint32_t a = L_act[s];\n            if (!a) continue;\n            const uint32_t am = res_mask<PACK>(a);\n"
              << "            const int32_t *ed = A.idx + " << C.idx_off << " + j;\n            uint32_t *Ms = M + s * E;\n            constexpr int ES = " << C.n << ";      // [k][node] table: entry k of this check at ed[k * ES]\n";
            if (C.deg <= 16) {
                o << "            int e[" << C.deg << "]; uint32_t x[" << C.deg << "], r[" << C.deg << "];\n"
                  << "#pragma unroll\n            for (int k = 0; k < " << C.deg << "; k++) e[k] = ed[k * ES];\n"
                  << "#pragma unroll\n            for (int k = 0; k < " << C.deg << "; k++) x[k] = Ms[e[k]];\n"
                  << "            const uint32_t tn = res_minsum<" << C.deg << ", PACK>(x, r, sbit, SB, LOW);\n"
                  << "            if (chk) flag(s, (tn >> sbit) & a);\n"
                  << "#pragma unroll\n            for (int k = 0; k < " << C.deg << "; k++) Ms[e[k]] = bfi(am, r[k], x[k]);\n";
            } else {
                // wide checks: two sweeps over the check's edges (the messages are re-read from LDS instead of held in registers)
                o << "            uint32_t max1 = 0u, max2 = 0u, spp = 0u;      // the two LARGEST complemented magnitudes = the two smallest magnitudes\n#pragma unroll 4\n            for (int k = 0; k < " << C.deg << "; k++) {\n"
                  << "                const uint32_t xh = Ms[ed[k * ES]];\n                const uint32_t pos = xh & SB, mc = (xh ^ (pos - (pos >> sbit))) & LOW;\n"
                  << "                spp ^= xh;\n                const uint32_t g1 = xad(max1, LOW, mc) & SB, k1 = g1 - (g1 >> sbit);      // mc > max1\n"
                  << "                const uint32_t hi = bfi(k1, mc, max1), lo = xor3(mc, max1, hi);\n"
                  << "                const uint32_t g2 = xad(max2, LOW, lo) & SB, k2 = g2 - (g2 >> sbit);\n"
                  << "                max2 = bfi(k2, lo, max2);\n                max1 = hi;\n            }\n"
                  << "            const uint32_t tn = (spp ^ " << ((C.deg & 1) ? "SB" : "0u") << ") & SB;\n"
                  << "            if (chk) flag(s, (tn >> sbit) & a);\n"
                  << "#pragma unroll 4\n            for (int k = 0; k < " << C.deg << "; k++) {\n"
                  << "                const int ek = ed[k * ES];\n                const uint32_t xh = Ms[ek];\n                const uint32_t pos = xh & SB, mc = (xh ^ (pos - (pos >> sbit))) & LOW;\n"
                  << "                const uint32_t eq = ~(((mc ^ max1) | SB) - ONE) & SB, ke = eq - (eq >> sbit);\n"
                  << "                const uint32_t sel = bfi(ke, max2, max1), po = (tn ^ xh) & SB, kp = po - (po >> sbit);\n"
                  << "                Ms[ek] = bfi(am, xor_or(sel, kp, po), xh);\n            }\n";
            }
            o << "        }\n";
        }
    } else {
        // CHKTREE check update: one code variant per distinct program text, selected by the tree set of the iteration
        o << "        const int set = kSet[ii];\n";
        for (size_t c : corder) {
            const ResidentClass &C = R.ccls[c];
            std::vector<std::string> bodies;
            std::vector<int> variant_of(n_sets_tree, -1);
            for (size_t s = 0; s < n_sets_tree; s++) {
                if (s >= R.chk_prog.size() || c >= R.chk_prog[s].size() || !R.chk_prog[s][c]) continue;
                std::ostringstream b;
                if (!emit_chk_frames(b, *R.chk_prog[s][c], C.deg, "                ", err)) return false;
                size_t v = 0;
                while (v < bodies.size() && bodies[v] != b.str()) v++;
                if (v == bodies.size()) bodies.push_back(b.str());
                variant_of[s] = (int)v;
            }
            emit_int_array(o, "kChkVar" + S_((long long)c), variant_of);
            o << "        " << cn_loop(c)
              << "            const int s = it / " << C.n << ", j = it - s * " << C.n << ";\n"
              << "            const uint32_t a = L_act[s];\n            if (!a) continue;\n            const uint32_t am = res_mask<PACK>(a);\n"
              << "            const int32_t *ed = A.idx + " << C.idx_off << " + j;\n            uint32_t *Ms = M + s * E;\n            constexpr int ES = " << C.n << ";\n"
              << "            const uint8_t *tc = TC + " << tc_off[c] << ";\n"
              << "            int e[" << C.deg << "]; uint32_t x[" << C.deg << "], out[" << C.deg << "];\n            uint32_t par_w = 0u;\n"
              << "#pragma unroll\n            for (int k = 0; k < " << C.deg << "; k++) { e[k] = ed[k * ES]; out[k] = 0u; }\n"
              << "#pragma unroll\n            for (int k = 0; k < " << C.deg << "; k++) x[k] = Ms[e[k]];\n"
              << "            switch (kChkVar" << c << "[set]) {\n";
            for (size_t v = 0; v < bodies.size(); v++) o << "            case " << v << ": {\n" << bodies[v] << "            } break;\n";
            o << "            default: break;\n            }\n"
              << "            if (chk) flag(s, par_w & a);\n"
              << "#pragma unroll\n            for (int k = 0; k < " << C.deg << "; k++) Ms[e[k]] = bfi(am, out[k], x[k]);\n        }\n";
        }
    }
    o << "    };\n";

    // ---- the variable pass / the decision pass: every thread walks its own items
    auto emit_vn_pass = [&](const std::string &fname, int kind) -> bool {
        const auto &progs = kind == TT_VAR ? R.var_prog : R.dec_prog;
        o << "    auto " << fname << " = [&](int set, int ii, bool chk) {\n        (void)ii; (void)chk;\n";
        if (kind == TT_VAR) {
            if (msg_pow2) o << "        const int sbit = __builtin_ctz((unsigned)kNz[ii + 1] | 0x100u);\n";
            else o << "        const uint32_t nzo = (uint32_t)kNz[ii + 1];\n";
        }
        // variants of every class
        std::vector<std::vector<std::string>> bodies(NVC);
        std::vector<std::vector<int>> variant_of(NVC, std::vector<int>(n_sets_tree, -1));
        for (size_t c = 0; c < NVC; c++) {
            const int deg = R.vcls[c].deg, U = R.U > 0 ? R.U : (deg <= 8 ? 2 : 1);
            for (size_t s = 0; s < n_sets_tree; s++) {
                if (s >= progs.size() || c >= progs[s].size() || !progs[s][c]) continue;
                std::ostringstream b;
                if (!emit_var_frames(b, *progs[s][c], kind, deg, ((4 * PACK) % U) ? 1 : U, "                ", err)) return false;
                size_t v = 0;
                while (v < bodies[c].size() && bodies[c][v] != b.str()) v++;
                if (v == bodies[c].size()) bodies[c].push_back(b.str());
                variant_of[c][s] = (int)v;
            }
            emit_int_array(o, (kind == TT_VAR ? "kVarV" : "kDecV") + S_((long long)c), variant_of[c]);
        }
        for (auto &it : items) {
            const ResidentClass &C = R.vcls[(size_t)it.c];
            const int deg = C.deg;
            o << "        if (ma" << it.sfx << " >= 0) {\n            const int sv = sv" << it.sfx << "();\n            const uint32_t a = L_act[sv];\n            if (a) {\n"
              << "                const uint32_t am = res_mask<PACK>(a);\n                const uint8_t *tb = TV + " << tv_off[(size_t)it.c] << ";\n"
              << "                uint32_t raw[" << deg + 1 << "], out[" << deg << "], hardw = 0u;\n"
              << "#pragma unroll\n                for (int k = 0; k < " << deg << "; k++) { raw[k] = M[ma" << it.sfx << " + k]; out[k] = 0u; }\n"
              << "                raw[" << deg << "] = cha" << it.sfx << ";\n"
              << "                switch (" << (kind == TT_VAR ? "kVarV" : "kDecV") << it.c << "[set]) {\n";
            for (size_t v = 0; v < bodies[(size_t)it.c].size(); v++) o << "                case " << v << ": {\n" << bodies[(size_t)it.c][v] << "                } break;\n";
            o << "                default: break;\n                }\n";
            if (kind == TT_DEC) {
                o << "                hard" << it.sfx << " = bfi(am, hardw, hard" << it.sfx << ");\n                (void)out;\n";
            } else {
                // unanimity of the outgoing signs + decided bits = those signs (src/LDPC_Code_LUT.cpp:437-452), on the packed outputs
                o << "                if (chk) {\n                    uint32_t diff = 0u;\n";
                if (msg_pow2) {
                    o << "#pragma unroll\n                    for (int k = 1; k < " << deg << "; k++) diff |= out[k] ^ out[0];\n"
                      << "                    hardw = (~out[0] >> sbit) & ONE;\n                    const uint32_t f = (diff >> sbit) & a;\n";
                } else {
                    o << "                    hardw = res_lt<PACK>(out[0], nzo);\n#pragma unroll\n                    for (int k = 1; k < " << deg << "; k++) diff |= res_lt<PACK>(out[k], nzo) ^ hardw;\n"
                      << "                    const uint32_t f = diff & a;\n";
                }
                o << "                    flag(sv, f);\n"
                  << "                    hard" << it.sfx << " = bfi(am, hardw, hard" << it.sfx << ");\n                }\n"
                  << "#pragma unroll\n                for (int k = 0; k < " << deg << "; k++) M[ma" << it.sfx << " + k] = bfi(am, out[k], raw[k]);\n";
            }
            o << "            }\n        }\n";
        }
        o << "    };\n";
        return true;
    };
    if (!emit_vn_pass("vn_pass", TT_VAR)) return false;
    if (!emit_vn_pass("dec_pass", TT_DEC)) return false;

    // ---- iterations (src/LDPC_Code_LUT.cpp:301-338)
    o << "    if (alive) {\n        for (int ii = 0; ii < max_iters; ii++) {\n            const int set = kSet[ii];\n"
      << "            if (ii != max_iters - 1) stage_var(set); else stage_dec(set);      // (visible after the barrier that follows the check pass)\n"
      << "            cn_pass(ii, psc && ii > 0);\n            __syncthreads();\n"
      << "            if (psc && ii > 0) { alive = retire(ii, ST_DONE_PSC); if (!alive) break; }      // :327-329 returns (ii-1)+1\n"
      << "            if (ii == max_iters - 1) break;\n";
    if (!R.min_lut) o << "            stage_chk(kSet[ii + 1]);\n";
    o << "            vn_pass(set, ii, psc);\n            __syncthreads();\n        }\n    }\n"
      // ---- decision + final syndrome (:340-349)
      << "    if (alive) {\n        dec_pass(kSet[max_iters - 1], 0, false);\n        __syncthreads();\n        put_hard();\n        __syncthreads();\n"
      << "        syndrome();\n        __syncthreads();\n"
      << "        if (tid < S) {\n            const uint32_t a = L_act[tid], f = L_fail[tid];\n            const size_t f0 = (size_t)(q0 + tid) * F;\n"
      << "#pragma unroll\n            for (int n = 0; n < F; n++)\n                if ((a >> (n * BITS)) & 1u) A.iters[f0 + res_frame_of_element<PACK>(n)] = ((f >> (n * BITS)) & 1u) ? -max_iters : max_iters;\n        }\n    }\n";
    // ---- decided bits back to their rows
    for (auto &it : items)
        o << "    if (ma" << it.sfx << " >= 0 && q0 + sv" << it.sfx << "() < A.n_sets) { const int q = q0 + sv" << it.sfx
          << "(); if (A.fm_bits) fm_store(A.fm_bits, q, nd" << it.sfx << "(), hard" << it.sfx << "); else *reinterpret_cast<uint32_t *>(A.hard + ((size_t)(q >> 6) * N + (size_t)nd" << it.sfx << "()) * kRowBytes + (q & 63) * 4) = hard" << it.sfx << "; }\n";
    o << "}\n";
    src = o.str();
    return true;
}

}  // namespace lutldpc
