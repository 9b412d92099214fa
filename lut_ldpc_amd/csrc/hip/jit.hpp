// jit.hpp -- tree-specialised variable / decision pass kernels generated at decoder creation.
//
// kernels_fast.hpp expands ONE tree shape at compile time (the balanced binary tree ber_sim designs in
// its auto modes).  Every other shape the reference accepts -- trees read from a file
// (src/LUT_Tree.cpp:296-306, trees/6_32_wide.ini of the regular example), auto_bin_high, root_only -- used
// to run on the node-program interpreter of kernels_generic.hpp (LDS value slots, run-time operand
// indices: ~10 instructions per look-up and frame).  The node program (lut_program.hpp: the tree's
// look-ups after value numbering) is straight-line code, so it is turned into HIP source here -- one
// statement per look-up, operands as named registers, place values folded into shifts -- and compiled
// for gfx950 with hiprtc.  The kernel skeleton (row addressing, software pipeline, frame loop, early
// termination) is the same as vn_balanced_body's and uses the same helpers: kernels_common.hpp is
// embedded into the library as text (csrc/Makefile: kernels_common.inc) and prepended to the source.
// The interpreter stays as the fallback (hiprtc missing, LUTLDPC_JIT=0, tables too large for LDS and
// so on) and as the reference the generated kernels are tested against.
#pragma once
#include "lut_program.hpp"

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace lutldpc {

static const char *const kCommonHeaderText =
#include "kernels_common.inc"
    ;

struct JitKernel {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    // the code object handed to hipModuleLoadData stays alive as long as the module: the HIP runtime documents no
    // copy of the image, so it is not freed under a loaded module
    std::shared_ptr<std::vector<char>> image;
    bool ok() const { return fn != nullptr; }
    void release() { if (mod) (void)hipModuleUnload(mod); mod = nullptr; fn = nullptr; image.reset(); }
};

constexpr int kJitMaxLdsTable = 48 * 1024;     // class tables above this stay in global memory (L1 / L2 hits)

inline bool jit_pow2(uint32_t x) { return x && !(x & (x - 1)); }

// Source of the pass kernel of one degree class.  kind: TT_VAR or TT_DEC; tab_bytes: size of the class's
// table blob (Program::tables), read from `tables + P.tab_off[0]`.
inline bool jit_vn_source(const Program &prog, int kind, int deg, int pack, int tab_bytes, std::string &src, std::string &err)
{
    if (kind != TT_VAR && kind != TT_DEC) { err = "only variable / decision programs are generated"; return false; }
    if (prog.n_in != deg + 1) { err = "unexpected input count"; return false; }
    const int bits = 8 / pack;
    const bool in_lds = tab_bytes <= kJitMaxLdsTable;
    const int tab_pad = (tab_bytes + 15) / 16 * 16;
    std::ostringstream o;
    o << kCommonHeaderText << "\nusing namespace lutldpc;\n"
      << "extern \"C\" __global__ __launch_bounds__(256) void lutldpc_jit_pass(const FastParams *__restrict__ Pp, uint8_t *msgs, const uint8_t *cha, uint8_t *__restrict__ hard,\n"
      << "    const uint32_t *__restrict__ state_w, uint32_t *__restrict__ vfail_w, const uint8_t *__restrict__ tables, const int32_t *__restrict__ fast_idx)\n{\n"
      << "    const FastParams &P = *Pp;\n    constexpr int PACK = " << pack << ", DV = " << deg << ", F = 4 * PACK, BITS = " << bits << ";\n";
    if (in_lds) {
        o << "    __shared__ __attribute__((aligned(16))) uint8_t tab[" << tab_pad << "];\n"
          << "    {\n        const uint32_t *src = reinterpret_cast<const uint32_t *>(tables + P.tab_off[0]);\n"
          << "        for (int i = threadIdx.x; i < " << tab_pad / 4 << "; i += 256) reinterpret_cast<uint32_t *>(tab)[i] = src[i];\n    }\n"
          << "    __syncthreads();\n";
    } else {
        o << "    const uint8_t *tab = tables + P.tab_off[0];\n";
    }
    o << R"SRC(    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int gl = wave / P.waves_per_group;
    if (gl >= P.G) return;
    const int chunk = wave - gl * P.waves_per_group;
    const int g = gl + P.g0;
    uint32_t amask[PACK];
    if (load_active<PACK>(state_w, g, lane, amask)) return;
    const uint32_t smask = pack_masks<PACK>(amask);
    const int32_t *vtab = fast_idx + P.idx_off;                 // dense [n_nodes][2] = {node id, first edge}
    const rsrc_t mbase = make_rsrc(msgs + (size_t)g * (size_t)P.E * kRowBytes, (uint32_t)P.E * kRowBytes);
    const rsrc_t cbase = make_rsrc(cha + (size_t)g * (size_t)P.N * kRowBytes, (uint32_t)P.N * kRowBytes);
    uint8_t *hbase = hard + (size_t)g * (size_t)P.N * kRowBytes;
    const uint32_t lane4 = (uint32_t)lane * 4u;
    const int first = chunk * P.nodes_per_wave;
    int last = first + P.nodes_per_wave;
    if (last > P.n_nodes) last = P.n_nodes;
    const int sbit = __builtin_ctz((unsigned)P.nz | 0x100u);
    const bool chk = P.check != 0;
    uint32_t failw = 0;
    auto fetch = [&](int i, int &vv, int &ee, uint32_t (&r)[DV + 1]) {
        const int ic = i < last ? i : last - 1;
        vv = vtab[2 * (size_t)ic]; ee = vtab[2 * (size_t)ic + 1];
        const uint32_t off = lane4 | (i < last ? 0u : 0x80000000u);
#pragma unroll
        for (int k = 0; k < DV; k++) r[k] = ld_row(mbase, (uint32_t)(ee + k) * kRowBytes, off);
        r[DV] = ld_row(cbase, (uint32_t)vv * kRowBytes, off);
    };
    auto eval = [&](const uint32_t (&raw)[DV + 1], int v, int e0) {
        uint32_t out[DV], hardw = 0;
#pragma unroll
        for (int o = 0; o < DV; o++) out[o] = 0;
#pragma unroll 1
        for (int s = 0; s < F * BITS; s += BITS) {              // one frame per trip: labels unpacked, one op per look-up
            uint32_t r0 = 0, diff = 0;
)SRC";
    // ---- inputs
    std::vector<std::string> name((size_t)std::max(prog.n_slots, prog.n_in) + 1);
    for (int k = 0; k <= deg; k++) {
        o << "            const uint32_t i" << k << " = __builtin_amdgcn_ubfe(raw[" << k << "], (uint32_t)s, (uint32_t)BITS);\n";
        name[(size_t)k] = "i" + std::to_string(k);
    }
    // ---- one statement per look-up
    bool first_out = true;
    for (size_t j = 0; j < prog.ops.size(); j++) {
        const Op &op = prog.ops[j];
        if (op.kind != 0) { err = "check-type look-up in a variable program"; return false; }
        if ((size_t)op.dst >= name.size()) name.resize((size_t)op.dst + 1);
        bool all_pow2 = true;
        for (int c = 0; c < op.nchild; c++) all_pow2 = all_pow2 && jit_pow2(op.mult[c]) && jit_pow2(op.childK[c]);
        std::string label;
        for (int c = 0; c < op.nchild; c++) {
            const std::string &x = name[(size_t)op.child[c]];
            if (x.empty()) { err = "operand read before it is written"; return false; }
            if (c == 0) {
                label = op.mult[c] == 1 ? x : "(" + x + " * " + std::to_string(op.mult[c]) + "u)";
            } else if (all_pow2) {                // disjoint bit fields: (child << log2 mult) | rest, one v_lshl_or_b32
                label = "lshl_or(" + x + ", " + std::to_string(__builtin_ctz(op.mult[c])) + ", " + label + ")";
            } else {
                label = "(" + label + " + " + x + " * " + std::to_string(op.mult[c]) + "u)";
            }
        }
        const std::string t = "t" + std::to_string(j);
        o << "            const uint32_t " << t << " = tab[" << op.tab_off << "u + " << label << "];\n";
        name[(size_t)op.dst] = t;
        if (op.out_idx >= 0) {
            if (kind == TT_DEC) {
                o << "            hardw = lshl_or(" << t << " < 1u ? 1u : 0u, s, hardw);     // src/LDPC_Code_LUT.cpp:342\n";
            } else {
                o << "            out[" << op.out_idx << "] = lshl_or(" << t << ", s, out[" << op.out_idx << "]);\n";
                if (first_out) o << "            r0 = " << t << ";\n";
                else o << "            diff |= " << t << " ^ r0;\n";
                first_out = false;
            }
        }
    }
    if (kind == TT_VAR)
        o << "            if (chk) {   // unanimity of the outgoing signs (bit sbit), src/LDPC_Code_LUT.cpp:437-452\n"
          << "                hardw = lshl_or(((r0 >> sbit) & 1u) ^ 1u, s, hardw);\n"
          << "                failw = lshl_or((diff >> sbit) & 1u, s, failw);\n            }\n";
    o << "            (void)r0; (void)diff;\n        }\n";
    if (kind == TT_DEC) {
        o << "        store_row_masked<PACK>(reinterpret_cast<uint32_t *>(hbase + (size_t)v * kRowBytes + lane4), hardw, smask);\n        (void)e0; (void)out;\n";
    } else {
        o << "#pragma unroll\n        for (int k = 0; k < DV; k++) st_row(mbase, (uint32_t)(e0 + k) * kRowBytes, lane4, bfi(smask, out[k], raw[k]));\n"
          << "        if (chk && P.write_hard) store_row_masked<PACK>(reinterpret_cast<uint32_t *>(hbase + (size_t)v * kRowBytes + lane4), hardw, smask);\n";
    }
    o << R"SRC(    };
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
)SRC";
    if (kind == TT_VAR)
        o << "    if (chk) {\n        uint32_t fail[PACK];\n#pragma unroll\n        for (int h = 0; h < PACK; h++) fail[h] = unpack_half<PACK>(failw, h);\n"
          << "        flag_frames<PACK>(vfail_w, P.vfail_stride_w, g, lane, fail, amask);\n    }\n";
    o << "    (void)failw; (void)sbit; (void)chk;\n}\n";
    src = o.str();
    return true;
}

// Source of the CHKTREE check pass of one degree class (min_lut = false, src/LDPC_Code_LUT.cpp:416-426,
// src/LUT_Tree.cpp:792-807,420-445).  Every value is used as (sign, magnitude): label = sum of the
// children's magnitudes times their place values, the parity of the children's signs selects the half of
// the expanded table (lut_program.hpp: expand_table).  The kernel skeleton is cn_minsum_body's.
inline bool jit_cn_source(const Program &prog, int deg, int pack, int tab_bytes, std::string &src, std::string &err)
{
    if (prog.kind != TT_CHK || prog.n_in != deg || prog.n_out != deg) { err = "not a check program of this degree"; return false; }
    const int bits = 8 / pack;
    const bool in_lds = tab_bytes <= kJitMaxLdsTable;
    const int tab_pad = (tab_bytes + 15) / 16 * 16;
    const bool pipe = deg <= 16;
    std::ostringstream o;
    o << kCommonHeaderText << "\nusing namespace lutldpc;\n"
      << "extern \"C\" __global__ __launch_bounds__(256) void lutldpc_jit_pass(const FastParams *__restrict__ Pp, uint8_t *msgs, const uint8_t *cha, uint8_t *__restrict__ hard,\n"
      << "    const uint32_t *__restrict__ state_w, uint32_t *__restrict__ vfail_w, const uint8_t *__restrict__ tables, const int32_t *__restrict__ fast_idx)\n{\n"
      << "    const FastParams &P = *Pp;\n    constexpr int PACK = " << pack << ", DEG = " << deg << ", F = 4 * PACK, BITS = " << bits << ";\n    (void)cha; (void)hard;\n";
    if (in_lds) {
        o << "    __shared__ __attribute__((aligned(16))) uint8_t tab[" << tab_pad << "];\n"
          << "    {\n        const uint32_t *src = reinterpret_cast<const uint32_t *>(tables + P.tab_off[0]);\n"
          << "        for (int i = threadIdx.x; i < " << tab_pad / 4 << "; i += 256) reinterpret_cast<uint32_t *>(tab)[i] = src[i];\n    }\n"
          << "    __syncthreads();\n";
    } else {
        o << "    const uint8_t *tab = tables + P.tab_off[0];\n";
    }
    o << R"SRC(    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int gl = wave / P.waves_per_group;
    if (gl >= P.G) return;
    const int chunk = wave - gl * P.waves_per_group;
    const int g = gl + P.g0;
    uint32_t amask[PACK];
    if (load_active<PACK>(state_w, g, lane, amask)) return;
    const uint32_t smask = pack_masks<PACK>(amask);
    const rsrc_t base = make_rsrc(msgs + (size_t)g * (size_t)P.E * kRowBytes, (uint32_t)P.E * kRowBytes);
    const uint32_t lane4 = (uint32_t)lane * 4u;
    const int32_t *edges = fast_idx + P.idx_off;                // dense [n_nodes][DEG] edge ids
    const int first = chunk * P.nodes_per_wave;
    int last = first + P.nodes_per_wave;
    if (last > P.n_nodes) last = P.n_nodes;
    const uint32_t nz = (uint32_t)P.nz;
    const bool chk = P.check != 0;
    uint32_t failw = 0;
    auto fetch = [&](int i, uint32_t (&xx)[DEG], int (&ee)[DEG]) {
        const int ii = i < last ? i : last - 1;
#pragma unroll
        for (int k = 0; k < DEG; k++) ee[k] = edges[(size_t)ii * DEG + k];
        const uint32_t off = lane4 | (i < last ? 0u : 0x80000000u);
#pragma unroll
        for (int k = 0; k < DEG; k++) xx[k] = ld_row(base, (uint32_t)ee[k] * kRowBytes, off);
    };
    auto eval = [&](const uint32_t (&x)[DEG], const int (&e)[DEG]) {
        uint32_t out[DEG];
#pragma unroll
        for (int k = 0; k < DEG; k++) out[k] = 0;
#pragma unroll 1
        for (int s = 0; s < F * BITS; s += BITS) {              // one frame per trip
)SRC";
    std::vector<std::string> name((size_t)std::max(prog.n_slots, prog.n_in) + 1);
    std::vector<int> sm_of((size_t)name.size(), 0);                 // threshold the (sign, magnitude) pair of a slot was emitted for
    for (int k = 0; k < deg; k++) {
        o << "            const uint32_t i" << k << " = __builtin_amdgcn_ubfe(x[" << k << "], (uint32_t)s, (uint32_t)BITS);\n";
        name[(size_t)k] = "i" + std::to_string(k);
    }
    o << "            if (chk) {   // parity of the incoming signs: second half of syndrome_check, src/LDPC_Code_LUT.cpp:437-452\n                uint32_t par = 0;\n";
    for (int k = 0; k < deg; k++) o << "                par ^= i" << k << " < nz ? 1u : 0u;\n";
    o << "                failw = lshl_or(par, s, failw);\n            }\n";
    std::vector<std::string> smname(name.size());
    for (size_t j = 0; j < prog.ops.size(); j++) {
        const Op &op = prog.ops[j];
        if (op.kind != 1 && op.kind != 2) { err = "variable-type look-up in a check program"; return false; }
        if ((size_t)op.dst >= name.size()) { name.resize((size_t)op.dst + 1); sm_of.resize(name.size(), 0); smname.resize(name.size()); }
        if (op.kind == 2) {       // table over the children's full labels (lut_program.hpp: chk_full_label_program): one op per look-up
            bool p2 = true;
            for (int c = 0; c < op.nchild; c++) p2 = p2 && jit_pow2(op.mult[c]) && jit_pow2(op.childK[c]);
            std::string label;
            for (int c = 0; c < op.nchild; c++) {
                const std::string &x = name[(size_t)op.child[c]];
                if (x.empty()) { err = "operand read before it is written"; return false; }
                if (c == 0) label = op.mult[c] == 1 ? x : "(" + x + " * " + std::to_string(op.mult[c]) + "u)";
                else if (p2) label = "lshl_or(" + x + ", " + std::to_string(__builtin_ctz(op.mult[c])) + ", " + label + ")";
                else label = "(" + label + " + " + x + " * " + std::to_string(op.mult[c]) + "u)";
            }
            const std::string t = "t" + std::to_string(j);
            o << "            const uint32_t " << t << " = tab[" << op.tab_off << "u + " << label << "];\n";
            name[(size_t)op.dst] = t;
            if (op.out_idx >= 0) o << "            out[" << op.out_idx << "] = lshl_or(" << t << ", s, out[" << op.out_idx << "]);\n";
            continue;
        }
        bool all_pow2 = jit_pow2(op.half_len);
        for (int c = 0; c < op.nchild; c++) all_pow2 = all_pow2 && jit_pow2(op.mult[c]) && jit_pow2((uint32_t)op.childK[c] >> 1);
        std::string label, par;
        for (int c = 0; c < op.nchild; c++) {
            const size_t sl = op.child[c];
            const std::string &x = name[sl];
            if (x.empty()) { err = "operand read before it is written"; return false; }
            const int hh = op.childK[c] >> 1;
            if (sm_of[sl] != hh || smname[sl] != x) {             // (sign, magnitude) of this value, once
                o << "            const uint32_t n_" << x << " = " << x << " < " << hh << "u ? 1u : 0u, m_" << x << " = n_" << x << " ? " << hh - 1 << "u - " << x
                  << " : " << x << " - " << hh << "u;\n";
                sm_of[sl] = hh; smname[sl] = x;
            }
            const std::string m = "m_" + x, n = "n_" + x;
            if (c == 0) label = op.mult[c] == 1 ? m : "(" + m + " * " + std::to_string(op.mult[c]) + "u)";
            else if (all_pow2) label = "lshl_or(" + m + ", " + std::to_string(__builtin_ctz(op.mult[c])) + ", " + label + ")";
            else label = "(" + label + " + " + m + " * " + std::to_string(op.mult[c]) + "u)";
            par = c == 0 ? n : par + " ^ " + n;
        }
        // odd sign parity -> first half of the expanded table, even -> second half (offset half_len)
        std::string idx = all_pow2 ? "lshl_or((" + par + ") ^ 1u, " + std::to_string(__builtin_ctz(op.half_len)) + ", " + label + ")"
                                   : "(" + label + " + ((" + par + ") ? 0u : " + std::to_string(op.half_len) + "u))";
        const std::string t = "t" + std::to_string(j);
        o << "            const uint32_t " << t << " = tab[" << op.tab_off << "u + " << idx << "];\n";
        name[(size_t)op.dst] = t;
        if (op.out_idx >= 0) o << "            out[" << op.out_idx << "] = lshl_or(" << t << ", s, out[" << op.out_idx << "]);\n";
    }
    o << "        }\n#pragma unroll\n        for (int k = 0; k < DEG; k++) st_row(base, (uint32_t)e[k] * kRowBytes, lane4, bfi(smask, out[k], x[k]));\n    };\n";
    if (pipe)
        o << R"SRC(    uint32_t x[DEG], xn[DEG];
    int e[DEG], en[DEG];
    fetch(first, x, e);
    fetch(first + 1, xn, en);
    eval(x, e);
    pipeline_entry_fence();
    for (int i = first + 1; i < last; i++) {
#pragma unroll
        for (int k = 0; k < DEG; k++) { x[k] = xn[k]; e[k] = en[k]; }
        fetch(i + 1, xn, en);
        eval(x, e);
    }
)SRC";
    else
        o << R"SRC(    for (int i = first; i < last; i++) {
        uint32_t x[DEG];
        int e[DEG];
        fetch(i, x, e);
        eval(x, e);
    }
)SRC";
    o << R"SRC(    if (chk) {
        uint32_t fail[PACK];
#pragma unroll
        for (int h = 0; h < PACK; h++) fail[h] = unpack_half<PACK>(failw, h);
        flag_frames<PACK>(vfail_w, P.vfail_stride_w, g, lane, fail, amask);
    }
}
)SRC";
    src = o.str();
    return true;
}

// hiprtc: source -> gfx950 code object (no device needed); log receives the compiler output
inline bool jit_compile(const std::string &src, std::vector<char> &code, std::string &log)
{
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "lutldpc_jit_pass.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { log = "hiprtcCreateProgram failed"; return false; }
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    const hiprtcResult rc = hiprtcCompileProgram(prog, 3, opts);
    size_t n = 0;
    if (hiprtcGetProgramLogSize(prog, &n) == HIPRTC_SUCCESS && n > 1) { log.resize(n); (void)hiprtcGetProgramLog(prog, &log[0]); }
    bool ok = rc == HIPRTC_SUCCESS;
    if (ok) {
        size_t cs = 0;
        ok = hiprtcGetCodeSize(prog, &cs) == HIPRTC_SUCCESS && cs > 0;
        if (ok) { code.resize(cs); ok = hiprtcGetCode(prog, code.data()) == HIPRTC_SUCCESS; }
    }
    (void)hiprtcDestroyProgram(&prog);
    return ok;
}

inline bool jit_load(const std::vector<char> &code, JitKernel &k, std::string &log)
{
    k.image = std::make_shared<std::vector<char>>(code);
    if (hipModuleLoadData(&k.mod, k.image->data()) != hipSuccess) { log = "hipModuleLoadData failed"; (void)hipGetLastError(); k.mod = nullptr; return false; }
    if (hipModuleGetFunction(&k.fn, k.mod, "lutldpc_jit_pass") != hipSuccess) { log = "kernel symbol missing"; (void)hipGetLastError(); k.release(); return false; }
    return true;
}

}  // namespace lutldpc
