// ldpc_code_lut.cpp -- see ldpc_code_lut.hpp.  Reference: src/LDPC_Code_LUT.cpp.
#include "ldpc_code_lut.hpp"
#include "itfile.hpp"
#include "lut_ldpc_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <unistd.h>

namespace lut_ldpc {

namespace {
[[noreturn]] void hip_fail(const char *what) {
    throw std::runtime_error(std::string(what) + ": " + lutldpc_last_error());
}
}  // namespace

// ------------------------------------------------------------------ LDPC_Generator_Systematic
void LDPC_Generator_Systematic::construct(LDPC_Parity *H) {
    const int N = H->get_nvar(), M = H->get_ncheck();
    if ((double)N * (double)M > 6.4e8)
        throw std::runtime_error("LDPC_Generator_Systematic: matrix too large for dense elimination; simulate with zero_codeword = true");
    const size_t W = ((size_t)N + 63) / 64;
    std::vector<uint64_t> A((size_t)M * W, 0);
    for (int r = 0; r < M; r++) for (int v : H->get_row(r)) A[(size_t)r * W + (size_t)(v >> 6)] ^= 1ull << (v & 63);
    // reduced row echelon form; prefer pivots in the LAST columns so that the information
    // positions are the leading ones for the typical [A | T] layouts
    std::vector<int> pivot_col;
    int prow = 0;
    for (int col = N - 1; col >= 0 && prow < M; col--) {
        const size_t w = (size_t)(col >> 6);
        const uint64_t bit = 1ull << (col & 63);
        int p = -1;
        for (int r = prow; r < M; r++) if (A[(size_t)r * W + w] & bit) { p = r; break; }
        if (p < 0) continue;
        if (p != prow) std::swap_ranges(A.begin() + (long)((size_t)p * W), A.begin() + (long)((size_t)p * W + W), A.begin() + (long)((size_t)prow * W));
        for (int r = 0; r < M; r++)
            if (r != prow && (A[(size_t)r * W + w] & bit))
                for (size_t j = 0; j < W; j++) A[(size_t)r * W + j] ^= A[(size_t)prow * W + j];
        pivot_col.push_back(col);
        prow++;
    }
    R_ = prow; N_ = N; K_ = N - R_;
    std::vector<char> is_pivot((size_t)N, 0);
    for (int c : pivot_col) is_pivot[(size_t)c] = 1;
    std::vector<int> perm;            // new column j = old column perm[j]
    for (int c = 0; c < N; c++) if (!is_pivot[(size_t)c]) perm.push_back(c);
    for (int c : pivot_col) perm.push_back(c);
    // parity bit of pivot row i (new position K_ + i) = sum over information columns
    const size_t WK = ((size_t)K_ + 63) / 64;
    A_.assign((size_t)R_ * (WK ? WK : 1), 0);
    for (int i = 0; i < R_; i++)
        for (int j = 0; j < K_; j++) {
            const int oc = perm[(size_t)j];
            if (A[(size_t)i * W + (size_t)(oc >> 6)] >> (oc & 63) & 1) A_[(size_t)i * WK + (size_t)(j >> 6)] |= 1ull << (j & 63);
        }
    H->permute_cols(perm);
    init_ = true;
}

void LDPC_Generator_Systematic::encode(const bvec &input, bvec &output) const {
    if (!init_) throw std::logic_error("LDPC_Generator_Systematic::encode(): generator not initialised");
    if ((int)input.size() != K_) throw std::invalid_argument("LDPC_Generator_Systematic::encode(): wrong input length");
    const size_t WK = ((size_t)K_ + 63) / 64;
    std::vector<uint64_t> u(WK ? WK : 1, 0);
    for (int j = 0; j < K_; j++) if (input[(size_t)j]) u[(size_t)(j >> 6)] |= 1ull << (j & 63);
    output.assign((size_t)N_, 0);
    std::copy(input.begin(), input.end(), output.begin());
    for (int i = 0; i < R_; i++) {
        uint64_t acc = 0;
        for (size_t w = 0; w < WK; w++) acc ^= A_[(size_t)i * WK + w] & u[w];
        output[(size_t)(K_ + i)] = (unsigned char)(__builtin_popcountll(acc) & 1);
    }
}

void LDPC_Generator_Systematic::save(const std::string &filename) const {
    it_file_writer f(filename, /*truncate=*/false);
    f.write("G_type", std::string("systematic_lutldpc"));
    f.write("G_N", N_); f.write("G_K", K_); f.write("G_R", R_);
    std::vector<int> words;
    for (uint64_t w : A_) { words.push_back((int)(uint32_t)w); words.push_back((int)(uint32_t)(w >> 32)); }
    f.write("G_A", words);
}

void LDPC_Generator_Systematic::load(const std::string &filename) {
    it_file_reader f(filename);
    if (!f.has("G_A")) { init_ = false; return; }
    N_ = f.get_int("G_N"); K_ = f.get_int("G_K"); R_ = f.get_int("G_R");
    const std::vector<int> words = f.get_ivec("G_A");
    A_.assign(words.size() / 2, 0);
    for (size_t i = 0; i < A_.size(); i++) A_[i] = (uint64_t)(uint32_t)words[2 * i] | ((uint64_t)(uint32_t)words[2 * i + 1] << 32);
    init_ = true;
}

// ------------------------------------------------------------------ construction
LDPC_Code_LUT::LDPC_Code_LUT() { max_iters = 0; }

LDPC_Code_LUT::LDPC_Code_LUT(const LDPC_Parity *H, LDPC_Generator *G_in, bool check) { set_code(H, G_in, check); }

LDPC_Code_LUT::LDPC_Code_LUT(const LDPC_Parity *H, const LUT_Tree_Array &var_trees_, const bvec &reuse_vec_, int Nq_Cha_, const ivec &Nq_Msg_,
                             const vec &qb_Cha_, const vec &qb_Msg_, LDPC_Generator *G_in, bool check) {
    set_code(H, G_in, check);
    reuse_vec = reuse_vec_; max_iters = (int)reuse_vec.size();
    Nq_Cha = Nq_Cha_; Nq_Msg = Nq_Msg_; qb_Cha = qb_Cha_; qb_Msg = qb_Msg_;
    minLUT = true;
    set_trees(var_trees_, check);
    LUTs_defined = true;
}

LDPC_Code_LUT::LDPC_Code_LUT(const LDPC_Parity *H, const LUT_Tree_Array &var_trees_, const LUT_Tree_Array &chk_trees_, const bvec &reuse_vec_,
                             int Nq_Cha_, const ivec &Nq_Msg_, const vec &qb_Cha_, const vec &qb_Msg_, LDPC_Generator *G_in, bool check) {
    set_code(H, G_in, check);
    reuse_vec = reuse_vec_; max_iters = (int)reuse_vec.size();
    Nq_Cha = Nq_Cha_; Nq_Msg = Nq_Msg_; qb_Cha = qb_Cha_; qb_Msg = qb_Msg_;
    minLUT = false;
    set_trees(var_trees_, chk_trees_, check);
    LUTs_defined = true;
}

LDPC_Code_LUT::LDPC_Code_LUT(const std::string &filename, LDPC_Generator *G_in) { psc = true; pisc = false; load_code(filename, G_in); }

LDPC_Code_LUT::~LDPC_Code_LUT() { drop_device(); }

void LDPC_Code_LUT::drop_device() {
    if (dev) { lutldpc_decoder_destroy(dev); dev = nullptr; }
}

void LDPC_Code_LUT::set_device(int d) {
    if (d != device) drop_device();
    device = d;
}

void LDPC_Code_LUT::set_code(const LDPC_Parity *H, LDPC_Generator *G_in, bool perform_integrity_check) {
    decoder_parameterization(H);
    G = G_in;
    if (G) {
        G_defined = true;
        if (perform_integrity_check) integrity_check();
    }
    drop_device();
}

void LDPC_Code_LUT::set_code_with_rank(const LDPC_Parity *H, LDPC_Generator *G_in, int known_rank) {
    decoder_parameterization(H, known_rank);
    G = G_in;
    if (G) { G_defined = true; integrity_check(); }
    drop_device();
}

// src/LDPC_Code_LUT.cpp:488-541
void LDPC_Code_LUT::decoder_parameterization(const LDPC_Parity *Hmat, int known_rank) {
    nvar = Hmat->nvar; nchk = Hmat->ncheck;
    // the reference skips the (dense) rank computation for nvar >= 1e5 and assumes full rank
    if (known_rank > 0) nchk_lin_indep = known_rank;
    else nchk_lin_indep = nvar < 1e5 ? Hmat->row_rank() : nchk;
    dv_vec = Hmat->sumX1; dc_vec = Hmat->sumX2;
    num_edges = 0;
    for (int w : dv_vec) num_edges += w;
    // edges are numbered variable node by variable node, rows ascending within a node;
    // cn_msg_idx lists for every check the ids of its edges in that numbering order
    std::vector<ivec> per_check((size_t)nchk);
    int e = 0;
    for (int v = 0; v < nvar; v++) for (int r : Hmat->get_col(v)) per_check[(size_t)r].push_back(e++);
    cn_msg_idx.clear();
    for (auto &l : per_check) cn_msg_idx.insert(cn_msg_idx.end(), l.begin(), l.end());
    if ((int)cn_msg_idx.size() != num_edges) throw std::logic_error("LDPC_Code_LUT::decoder_parameterization(): dimension mismatch");
    chk_equ_idx.assign((size_t)nchk, {});
    for (int c = 0; c < nchk; c++) chk_equ_idx[(size_t)c] = Hmat->get_row(c);
    H_defined = true;
}

void LDPC_Code_LUT::integrity_check() {   // :547-566: every unit vector must encode to a codeword
    if (!G_defined) return;
    const int K = get_ninfo();
    for (int trial = 0; trial < std::min(K, 64); trial++) {
        bvec in((size_t)K, 0), cw;
        in[(size_t)((long long)trial * K / std::min(K, 64))] = 1;
        G->encode(in, cw);
        if (!syndrome_check(cw)) throw std::runtime_error("LDPC_Code_LUT::integrity_check(): generator and parity-check matrix mismatch");
    }
}

// src/LDPC_Code_LUT.cpp:120-169: only validation remains here -- the iteration -> tree-set map
// and the degree-class match are (re)derived by the HIP decoder from the same inputs
void LDPC_Code_LUT::set_trees(const LUT_Tree_Array &var_trees_, const LUT_Tree_Array &chk_trees_, bool) {
    if (reuse_vec.empty() || reuse_vec.front() || reuse_vec.back())
        throw std::invalid_argument("LDPC_Code_LUT::set_trees(): first and last iteration are exempt from tree reuse");
    if (var_trees_.empty()) throw std::invalid_argument("LDPC_Code_LUT::set_trees(): no variable node trees");
    for (int dv : dv_vec) {
        bool ok = false;
        for (auto &t : var_trees_[0]) if (t.get_num_leaves() == dv) { ok = true; break; }
        if (!ok) throw std::invalid_argument("LDPC_Code_LUT::set_trees(): no variable tree for degree " + std::to_string(dv));
    }
    if (!chk_trees_.empty())
        for (int dc : dc_vec) {
            bool ok = false;
            for (auto &t : chk_trees_[0]) if (t.get_num_leaves() + 1 == dc) { ok = true; break; }
            if (!ok) throw std::invalid_argument("LDPC_Code_LUT::set_trees(): no check tree for degree " + std::to_string(dc));
        }
    var_trees = var_trees_;
    chk_trees = chk_trees_;
    drop_device();
}

void LDPC_Code_LUT::set_trees(const LUT_Tree_Array &var_trees_, bool check) { set_trees(var_trees_, LUT_Tree_Array(), check); }

void LDPC_Code_LUT::set_exit_conditions(int max_iters_in, bool syndr_check_each_iter, bool syndr_check_at_start) {
    if (max_iters_in < 0) throw std::invalid_argument("LDPC_Code_LUT::set_nrof_iterations(): maximum number of iterations can not be negative");
    max_iters = max_iters_in; psc = syndr_check_each_iter; pisc = syndr_check_at_start;
    if (dev && lutldpc_decoder_set_exit_conditions(dev, max_iters, psc, pisc) != LUTLDPC_OK) hip_fail("LDPC_Code_LUT::set_exit_conditions()");
}

// ------------------------------------------------------------------ design cache
// A design is a pure function of (tree method [+ the tree file's text], degree distribution of the code, min_lut, sigma^2,
// iterations, reuse pattern, alphabets, degree-1 extension): the 50-iteration density evolution of DVB-S2 takes ~25 s of
// CPU every time ber_sim starts.  With LUTLDPC_DESIGN_CACHE=<directory> the result is kept there, keyed by a hash of all
// those inputs, as text: the quantiser boundaries as hex floats (exact) and the trees in the reference's own serialisation
// (operator<< of Array<Array<LUT_Tree>>, src/LUT_Tree.cpp:847-865) -- the same strings save_code writes into lut_codec.it.
// Off unless the variable is set (the design parity tests always design).
namespace {
// format tag + a hash of the design sources of THIS build (csrc/Makefile: design_src_hash.inc): a file written by a build whose
// design code differed is a foreign file and is ignored
const char *kDesignCacheTag = "lutldpc-design-cache-2-"
#include "design_src_hash.inc"
    ;

struct Fnv {
    uint64_t h = 1469598103934665603ull;
    void bytes(const void *p, size_t n) { const unsigned char *c = (const unsigned char *)p; for (size_t i = 0; i < n; i++) { h ^= c[i]; h *= 1099511628211ull; } }
    template <class T> void pod(const T &v) { bytes(&v, sizeof(T)); }
    template <class T> void vec_(const std::vector<T> &v) { pod((uint64_t)v.size()); if (!v.empty()) bytes(v.data(), v.size() * sizeof(T)); }
    void str(const std::string &s) { pod((uint64_t)s.size()); bytes(s.data(), s.size()); }
};

std::string design_cache_dir() {
    const char *e = std::getenv("LUTLDPC_DESIGN_CACHE");
    if (!e || !*e || std::string(e) == "0" || std::string(e) == "off") return std::string();
    return e;
}

std::string design_key(const std::string &tree_method, const LDPC_Ensemble &ens, bool min_lut, double sigma2, int max_iters, const bvec &reuse_vec,
                       int Nq_Cha, const ivec &Nq_Msg, bool allow_degree_one) {
    Fnv f;
    f.str(kDesignCacheTag); f.str(tree_method);
    if (tree_method.rfind("filename=", 0) == 0) {            // the trees come from a file: its text is part of the key
        std::ifstream in(tree_method.substr(9), std::ios::binary);
        std::stringstream ss; ss << in.rdbuf(); f.str(ss.str());
    }
    f.vec_(ens.sget_degree_lam()); f.vec_(ens.sget_lam()); f.vec_(ens.sget_degree_rho()); f.vec_(ens.sget_rho());
    f.pod((int)min_lut); f.pod(sigma2); f.pod(max_iters); f.vec_(reuse_vec); f.pod(Nq_Cha); f.vec_(Nq_Msg); f.pod((int)allow_degree_one);
    char buf[32];
    std::snprintf(buf, sizeof buf, "%016llx", (unsigned long long)f.h);
    return buf;
}

void write_dvec(std::ostream &o, const char *name, const vec &v) {
    o << name << ' ' << v.size();
    char buf[64];
    for (double x : v) { std::snprintf(buf, sizeof buf, " %a", x); o << buf; }
    o << '\n';
}
bool read_dvec(std::istream &in, const char *name, vec &v) {
    std::string tag; size_t n = 0;
    if (!(in >> tag >> n) || tag != name || n > (1u << 20)) return false;
    v.resize(n);
    for (auto &x : v) { std::string t; if (!(in >> t)) return false; x = std::strtod(t.c_str(), nullptr); }
    return true;
}
}  // namespace

bool LDPC_Code_LUT::load_design(const std::string &path, const std::string &key, LUT_Tree_Array &var_luts, LUT_Tree_Array &chk_luts) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    std::string tag, k;
    if (!(in >> tag >> k) || tag != kDesignCacheTag || k != key) return false;
    vec qc, qm;
    ivec map;
    if (!read_dvec(in, "qb_Cha", qc) || !read_dvec(in, "qb_Msg", qm)) return false;
    { std::string t; size_t n = 0; if (!(in >> t >> n) || t != "map" || n > 4096) return false; map.resize(n); for (auto &x : map) if (!(in >> x)) return false; }
    auto read_trees = [&](const char *name, LUT_Tree_Array &a) {
        std::string t; size_t n = 0;
        if (!(in >> t >> n) || t != name || n > (1u << 30)) return false;
        in.get();                                            // the newline after the length
        std::string txt(n, '\0');
        if (n && !in.read(&txt[0], (std::streamsize)n)) return false;
        std::istringstream is(txt);
        if (n) is >> a;
        return true;
    };
    try {
        if (!read_trees("var_trees", var_luts) || !read_trees("chk_trees", chk_luts)) return false;
    } catch (const std::exception &) { return false; }
    std::string end;
    if (!(in >> end) || end != "end") return false;         // a truncated file is not a design
    qb_Cha = qc; qb_Msg = qm; Nq_Cha_2_Nq_Msg_map = map;
    return true;
}

void LDPC_Code_LUT::store_design(const std::string &path, const std::string &key) const {
    const std::string tmp = path + ".tmp" + std::to_string((long long)::getpid());
    {
        std::error_code ec;                                     // a fresh checkout has no cache directory yet
        const auto dir = std::filesystem::path(path).parent_path();
        if (!dir.empty()) std::filesystem::create_directories(dir, ec);
    }
    {
        std::ofstream o(tmp, std::ios::binary);
        if (!o) return;                                      // an unwritable cache directory only costs the next start its design
        o << kDesignCacheTag << ' ' << key << '\n';
        write_dvec(o, "qb_Cha", qb_Cha); write_dvec(o, "qb_Msg", qb_Msg);
        o << "map " << Nq_Cha_2_Nq_Msg_map.size();
        for (int x : Nq_Cha_2_Nq_Msg_map) o << ' ' << x;
        o << '\n';
        // (chk_trees: with min_lut the fresh design leaves the check templates in place -- a cache hit must leave the same object behind)
        const std::string v = to_string(var_trees), c = chk_trees.empty() ? std::string() : to_string(chk_trees);
        o << "var_trees " << v.size() << '\n' << v << '\n' << "chk_trees " << c.size() << '\n' << c << '\n' << "end\n";
        if (!o) { o.close(); std::remove(tmp.c_str()); return; }
    }
    if (std::rename(tmp.c_str(), path.c_str()) != 0) std::remove(tmp.c_str());      // atomic: ranks of one node share the directory
}

// src/LDPC_Code_LUT.cpp:699-746
double LDPC_Code_LUT::design_luts(const std::string &tree_method, const LDPC_Ensemble &ens, bool min_lut, double sigma2, int max_iters_,
                                  const bvec &reuse_vec_, int Nq_Cha_, const ivec &Nq_Msg_, bool allow_degree_one) {
    minLUT = min_lut; max_iters = max_iters_; reuse_vec = reuse_vec_; Nq_Cha = Nq_Cha_; Nq_Msg = Nq_Msg_;
    if ((int)reuse_vec.size() != max_iters || (int)Nq_Msg.size() != max_iters)
        throw std::invalid_argument("LDPC_Code_LUT::design_luts(): reuse_vec / Nq_Msg must have max_iters entries");
    const double sig = std::sqrt(sigma2);
    LUT_Tree_Array var_luts, chk_luts;
    const std::string cache = design_cache_dir();
    std::string key, path;
    design_from_cache = false;
    if (!cache.empty()) {
        key = design_key(tree_method, ens, min_lut, sigma2, max_iters, reuse_vec, Nq_Cha, Nq_Msg, allow_degree_one);
        path = cache + "/" + key + ".lutdesign";
        if (load_design(path, key, var_luts, chk_luts)) {
            set_trees(var_luts, chk_luts);
            LUTs_defined = true;
            design_from_cache = true;
            return sig;
        }
    }
    get_lut_tree_templates(tree_method, ens, Nq_Msg, Nq_Cha, min_lut, var_luts, chk_luts, allow_degree_one);
    LDPC_DE_LUT de(ens, Nq_Cha, Nq_Msg, max_iters, var_luts, chk_luts, reuse_vec);
    de.get_quant_bound(sig, qb_Cha, qb_Msg);
    de.get_lut_trees(var_luts, chk_luts, sig);
    set_trees(var_luts, chk_luts);
    // Nq_Cha_2_Nq_Msg_map (:735-741): designed on a uniform grid of +-25, not on qb_Cha's cells
    const double LLR_max_mag = 25.0, delta = 2 * LLR_max_mag / Nq_Cha;
    const vec pmf_channel = get_gaussian_pmf(2 / (sig * sig), 2 / sig, Nq_Cha, delta);
    vec p_msg;
    (void)quant_mi_sym(p_msg, Nq_Cha_2_Nq_Msg_map, pmf_channel, Nq_Msg[0], true);
    LUTs_defined = true;
    if (!path.empty()) store_design(path, key);
    return sig;
}

// ------------------------------------------------------------------ device plumbing
lutldpc_decoder *LDPC_Code_LUT::device_handle() {
    if (dev) return dev;
    if (!H_defined) throw std::logic_error("LDPC_Code_LUT::lut_decode(): parity check matrix not defined");
    if (!LUTs_defined) throw std::logic_error("LDPC_Code_LUT::lut_decode(): LUTs not defined");
    const int I = (int)reuse_vec.size();
    const std::string vtxt = to_string(var_trees), ctxt = minLUT ? std::string() : to_string(chk_trees);
    if (lutldpc_decoder_create(nvar, nchk, dv_vec.data(), dc_vec.data(), cn_msg_idx.data(), Nq_Cha, Nq_Msg.data(), reuse_vec.data(), I,
                               minLUT ? 1 : 0, vtxt.c_str(), ctxt.c_str(), device, &dev) != LUTLDPC_OK)
        hip_fail("LDPC_Code_LUT: creating the HIP decoder failed");
    if (lutldpc_decoder_set_exit_conditions(dev, max_iters, psc, pisc) != LUTLDPC_OK) hip_fail("LDPC_Code_LUT::set_exit_conditions()");
    return dev;
}

void LDPC_Code_LUT::lut_decode_batch(const uint8_t *cha, const uint8_t *msg0, int B, uint8_t *bits, int32_t *iters) {
    if (output_verbosity > 1) { lut_decode_batch_dump(cha, msg0, B, bits, iters, output_verbosity > 2 ? 3 : 2, std::cout); return; }
    if (lutldpc_decoder_decode_batch(device_handle(), cha, msg0, B, bits, iters) != LUTLDPC_OK) hip_fail("LDPC_Code_LUT::lut_decode()");
}

// src/LDPC_Code_LUT.cpp:292-298, 311-317, 331-337: what lut_decode prints with output_verbosity > 1.  The device hands back every
// dump of every frame (lutldpc_decoder_decode_batch_trace); which of them the reference would have printed for a frame follows
// from its return code: 0 (passed the test on the channel decisions, :275-279) returns before the first print; a frame that
// leaves through the exit test of iteration ii (return ii + 1, :327-329) returns BEFORE the dump of that iteration's variable
// update; everything else prints all of them -- after the last iteration too, where no variable update ran (:331 is outside the
// `if`).  std::hex / std::uppercase / setfill('0') are sticky in the reference's stream, so the iteration numbers of the
// headlines come out in upper-case hex.
void LDPC_Code_LUT::lut_decode_batch_dump(const uint8_t *cha, const uint8_t *msg0, int B, uint8_t *bits, int32_t *iters, int level, std::ostream &os) {
    if (level < 2) level = 2;
    if (level > 3) level = 3;
    const int per_iter = level - 1, n_dumps = 1 + max_iters * per_iter;
    for (int f0 = 0; f0 < B; f0 += 64) {                           // a debug path: 64 frames per device call
        const int n = std::min(64, B - f0);
        std::vector<uint8_t> trace((size_t)n_dumps * (size_t)n * (size_t)num_edges);
        int32_t got = 0;
        if (lutldpc_decoder_decode_batch_trace(device_handle(), cha + (size_t)f0 * nvar, msg0 + (size_t)f0 * nvar, n, level, bits + (size_t)f0 * nvar, iters + f0,
                                               trace.data(), (int64_t)trace.size(), &got) != LUTLDPC_OK || got != n_dumps)
            hip_fail("LDPC_Code_LUT::lut_decode() with output_verbosity > 1");
        auto row = [&](int dump, int f) { return trace.data() + ((size_t)dump * (size_t)n + (size_t)f) * (size_t)num_edges; };
        auto print = [&](const uint8_t *m) {
            for (int e = 0; e < num_edges; e++) os << std::setfill('0') << std::setw(8) << std::uppercase << std::hex << (int)m[e] << "  ";
            os << std::endl;
        };
        for (int f = 0; f < n; f++) {
            const int rc = iters[f0 + f];
            if (rc == 0) continue;                                   // :275-279
            const auto flags = os.flags();
            const char fill = os.fill();
            os << "Initial VN-to-CN messages: " << std::endl;
            print(row(0, f));
            const int last = rc > 0 && rc < max_iters ? rc - 1 : max_iters - 1;       // last iteration that prints anything
            const bool early = rc > 0 && rc < max_iters;                             // (rc == max_iters: the final syndrome test, no early return)
            for (int ii = 0; ii <= last; ii++) {
                if (level > 2) { os << "CN-to-VN messages after CN update at iteration " << ii << ":" << std::endl; print(row(1 + ii * per_iter, f)); }
                if (early && ii == last) break;
                os << "VN-to-CN messages after VN update at iteration " << ii << ":" << std::endl;
                print(row(1 + ii * per_iter + (per_iter - 1), f));
            }
            os.flags(flags);
            os.fill(fill);
        }
    }
}

void LDPC_Code_LUT::decode_batch(const double *llr, int B, uint8_t *bits, int32_t *iters) {
    if (initial_message_mode != CONT && initial_message_mode != QCHA) throw std::logic_error("LDPC_Code_LUT::decode(): Initial message mode undefined!");
    if (lutldpc_decoder_decode_llr_batch(device_handle(), llr, B, qb_Cha.data(), (int)qb_Cha.size(), qb_Msg.data(), (int)qb_Msg.size(),
                                         initial_message_mode == QCHA ? 1 : 0, Nq_Cha_2_Nq_Msg_map.empty() ? nullptr : Nq_Cha_2_Nq_Msg_map.data(),
                                         bits, iters) != LUTLDPC_OK)
        hip_fail("LDPC_Code_LUT::decode()");
    if (output_verbosity > 0) {
        std::vector<uint8_t> cha((size_t)nvar);
        for (int f = 0; f < B; f++) {
            for (int v = 0; v < nvar; v++) cha[(size_t)v] = (uint8_t)quant_nonlin(llr[(size_t)f * nvar + v], qb_Cha);
            print_stimuli(cha.data(), bits + (size_t)f * nvar);
        }
    }
}

// src/LDPC_Code_LUT.cpp:228-238 -- the text the VHDL flow replays
void LDPC_Code_LUT::print_stimuli(const uint8_t *cha, const uint8_t *bits) const {
    std::ostream &o = std::cout;
    o << "Stimuli Pair (Quantized channel LLR decoder inputs in hex format and decoder output in binary format): " << std::endl;
    const auto flags = o.flags();
    for (int i = 0; i < nvar; i++) o << std::setfill('0') << std::setw(8) << std::uppercase << std::hex << (int)cha[i] << "  ";
    o << std::endl;
    for (int i = 0; i < nvar; i++) o << (int)bits[i] << "  ";
    o << std::endl << std::endl;
    o.flags(flags);
}

int LDPC_Code_LUT::lut_decode(const ivec &LLRin_cha, const ivec &LLRin_msg, bvec &LLRout) {
    if ((int)LLRin_cha.size() != nvar || (int)LLRin_msg.size() != nvar)
        throw std::invalid_argument("LDPC_Code_LUT::lut_decode(): Wrong input dimensions");
    std::vector<uint8_t> a((size_t)nvar), b((size_t)nvar);
    for (int v = 0; v < nvar; v++) { a[(size_t)v] = (uint8_t)LLRin_cha[(size_t)v]; b[(size_t)v] = (uint8_t)LLRin_msg[(size_t)v]; }
    LLRout.assign((size_t)nvar, 0);
    int32_t it = 0;
    lut_decode_batch(a.data(), b.data(), 1, LLRout.data(), &it);
    return it;
}

void LDPC_Code_LUT::decode(const vec &llr_in, bvec &syst_bits) {
    if ((int)llr_in.size() != nvar) throw std::invalid_argument("LDPC_Code_LUT::decode(): Wrong input dimensions");
    bvec all((size_t)nvar);
    int32_t it;
    decode_batch(llr_in.data(), 1, all.data(), &it);
    syst_bits.assign(all.begin(), all.begin() + get_ninfo());       // :225
}

bvec LDPC_Code_LUT::decode(const vec &llr_in) { bvec b; decode(llr_in, b); return b; }

void LDPC_Code_LUT::encode(const bvec &input, bvec &output) {
    if (!G_defined) throw std::logic_error("LDPC_Code_LUT::encode(): LDPC Generator is required for encoding");
    G->encode(input, output);
}

bvec LDPC_Code_LUT::encode(const bvec &input) { bvec o; encode(input, o); return o; }

bool LDPC_Code_LUT::syndrome_check(const bvec &b) const {   // :455-469
    for (auto &row : chk_equ_idx) {
        int synd = 0;
        for (int v : row) synd += b[(size_t)v] ? 1 : 0;
        if (synd & 1) return false;
    }
    return true;
}

// ------------------------------------------------------------------ codec files
void LDPC_Code_LUT::save_code(const std::string &filename) const {
    if (!H_defined) throw std::logic_error("LDPC_Code_LUT::save_to_file(): There is no parity check matrix");
    {
        it_file_writer f(filename);
        f.write("Fileversion", LUT_LDPC_binary_file_version);
        f.write("H_defined", H_defined); f.write("G_defined", G_defined); f.write("LUTs_defined", LUTs_defined);
        f.write("nvar", nvar); f.write("nchk", nchk); f.write("nchk_lin_indep", nchk_lin_indep);
        f.write("dv_vec", dv_vec); f.write("dc_vec", dc_vec);
        f.write("chk_equ_idx", chk_equ_idx); f.write("cn_msg_idx", cn_msg_idx);
        f.write("Nq_Cha", Nq_Cha); f.write("Nq_Msg", Nq_Msg); f.write("Nq_Cha_2_Nq_Msg_map", Nq_Cha_2_Nq_Msg_map);
        f.write("qb_Cha", qb_Cha); f.write("qb_Msg", qb_Msg); f.write("reuse_vec", reuse_vec);
        f.write("minLUT", minLUT); f.write("output_verbosity", output_verbosity); f.write("max_iters", max_iters);
        f.write("var_tree_string", to_string(var_trees));
        f.write("chk_tree_string", to_string(chk_trees));
        f.close();
    }
    if (G_defined) G->save(filename);
}

void LDPC_Code_LUT::load_code(const std::string &filename, LDPC_Generator *G_in) {
    it_file_reader f(filename);
    if (f.get_int("Fileversion") != LUT_LDPC_binary_file_version) throw std::runtime_error("LDPC_Code_LUT::load_code(): Unsupported file format");
    H_defined = f.get_bool("H_defined"); G_defined = f.get_bool("G_defined"); LUTs_defined = f.get_bool("LUTs_defined");
    nvar = f.get_int("nvar"); nchk = f.get_int("nchk"); nchk_lin_indep = f.get_int("nchk_lin_indep");
    dv_vec = f.get_ivec("dv_vec"); dc_vec = f.get_ivec("dc_vec");
    chk_equ_idx = f.get_ivec_array("chk_equ_idx"); cn_msg_idx = f.get_ivec("cn_msg_idx");
    max_iters = f.get_int("max_iters");
    Nq_Cha = f.get_int("Nq_Cha"); Nq_Msg = f.get_ivec("Nq_Msg"); Nq_Cha_2_Nq_Msg_map = f.get_ivec("Nq_Cha_2_Nq_Msg_map");
    qb_Cha = f.get_dvec("qb_Cha"); qb_Msg = f.get_dvec("qb_Msg"); reuse_vec = f.get_bvec("reuse_vec");
    minLUT = f.get_bool("minLUT"); output_verbosity = f.get_int("output_verbosity");
    LUT_Tree_Array vt, ct;
    { std::istringstream is(f.get_string("var_tree_string")); is >> vt; }
    { std::istringstream is(f.get_string("chk_tree_string")); is >> ct; }
    num_edges = 0;
    for (int w : dv_vec) num_edges += w;
    set_trees(vt, ct, true);
    if (G_defined) {
        if (!G_in) throw std::invalid_argument("LDPC_Code_LUT::load_code(): Generator object is missing");
        G = G_in;
        G->load(filename);
    } else G = nullptr;
}

std::ostream &operator<<(std::ostream &os, const LDPC_Code_LUT &C) {
    auto hist = [](const ivec &deg) {
        ivec h((size_t)*std::max_element(deg.begin(), deg.end()) + 1, 0);
        for (int d : deg) h[(size_t)d]++;
        std::ostringstream s;
        s << '[';
        for (size_t i = 0; i < h.size(); i++) s << (i ? " " : "") << h[i];
        s << ']';
        return s.str();
    };
    os << "--- LDPC codec ----------------------------------\n"
       << "Nvar : " << C.get_nvar() << "\n"
       << "Ncheck : " << C.get_nchk() << "\n"
       << "Rate : " << C.get_rate() << "\n"
       << "Column degrees (node perspective): " << hist(C.dv_vec) << "\n"
       << "Row degrees (node perspective): " << hist(C.dc_vec) << "\n"
       << "-------------------------------------------------\n"
       << "Decoder parameters:\n"
       << " - max. iterations : " << C.max_iters << "\n"
       << " - syndrome check at each iteration : " << C.psc << "\n"
       << " - syndrome check at start : " << C.pisc << "\n"
       << "-------------------------------------------------\n"
       << "Decoder back end: HIP (gfx950), frames batched on the device\n";
    return os;
}

}  // namespace lut_ldpc
