// ldpc_code_lut.hpp -- LDPC_Code_LUT: the reference's LUT / min-LUT codec object
// (src/LDPC_Code_LUT.hpp:66-366) with the decode path running on the MI355X.
//
// Same public surface (constructors, set_code, set_trees, design_luts, set_exit_conditions,
// encode, decode overloads, lut_decode, syndrome_check, getters, verbosity, initial message
// mode, save_code / load_code, operator<<), plus batched entry points -- the GPU wants many
// frames per call, so the frame loop of LDPC_BER_Sim::sim_snr_point moves below this API.
// All message passing happens in the HIP kernels behind include/lut_ldpc_hip.h; this class
// only builds the index arrays / tables and forwards.  Like the reference object it is not
// re-entrant (one instance per host thread / stream).
#pragma once
#include <ostream>
#include "ldpc_parity.hpp"
#include "lut_design.hpp"
#include "lut_tree.hpp"

#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

struct lutldpc_decoder;

namespace lut_ldpc {

// Systematic encoder (stands in for itpp::LDPC_Generator_Systematic, used at
// src/LDPC_BER_Sim.cpp:447-470 and src/LDPC_Code_LUT.cpp:192).  Construction reorders the
// columns of H so that the first nvar - rank(H) positions carry the information bits.  The
// column order IT++ would pick is not reproducible (its fork is absent); any valid
// order gives the same code up to a bit permutation.
class LDPC_Generator_Systematic {
public:
    LDPC_Generator_Systematic() = default;
    explicit LDPC_Generator_Systematic(LDPC_Parity *H) { construct(H); }
    void construct(LDPC_Parity *H);
    bool is_initialized() const { return init_; }
    void encode(const bvec &input, bvec &output) const;
    int get_ninfo() const { return K_; }
    void save(const std::string &filename) const;      // appended to an existing .it file
    void load(const std::string &filename);
private:
    bool init_ = false;
    int N_ = 0, K_ = 0, R_ = 0;
    std::vector<uint64_t> A_;   // R_ rows of ceil(K_/64) words: parity_i = <A_i, info>
};
using LDPC_Generator = LDPC_Generator_Systematic;

class LDPC_Code_LUT {
public:
    enum initial_message_mode_t { CONT, QCHA, num_initial_message_modes };   // hpp:75-79
    static const int LUT_LDPC_binary_file_version = 1;   // src/LDPC_Code_LUT.cpp:35

    LDPC_Code_LUT();
    LDPC_Code_LUT(const LDPC_Parity *H, LDPC_Generator *G = nullptr, bool perform_integrity_check = true);
    LDPC_Code_LUT(const LDPC_Parity *H, const LUT_Tree_Array &var_trees, const bvec &reuse_vec, int Nq_Cha, const ivec &Nq_Msg,
                  const vec &qb_Cha, const vec &qb_Msg, LDPC_Generator *G = nullptr, bool perform_integrity_check = true);
    LDPC_Code_LUT(const LDPC_Parity *H, const LUT_Tree_Array &var_trees, const LUT_Tree_Array &chk_trees, const bvec &reuse_vec, int Nq_Cha,
                  const ivec &Nq_Msg, const vec &qb_Cha, const vec &qb_Msg, LDPC_Generator *G = nullptr, bool perform_integrity_check = true);
    explicit LDPC_Code_LUT(const std::string &filename, LDPC_Generator *G = nullptr);
    ~LDPC_Code_LUT();
    LDPC_Code_LUT(const LDPC_Code_LUT &) = delete;
    LDPC_Code_LUT &operator=(const LDPC_Code_LUT &) = delete;

    void set_code(const LDPC_Parity *H, LDPC_Generator *G = nullptr, bool perform_integrity_check = true);
    // as set_code, with rank(H) supplied by the caller when known_rank > 0
    void set_code_with_rank(const LDPC_Parity *H, LDPC_Generator *G, int known_rank);
    void set_trees(const LUT_Tree_Array &var_trees, bool perform_integrity_check = true);
    void set_trees(const LUT_Tree_Array &var_trees, const LUT_Tree_Array &chk_trees, bool perform_integrity_check = true);
    // src/LDPC_Code_LUT.cpp:699-746; allow_degree_one: DESIGN.md "deviations"
    double design_luts(const std::string &tree_method, const LDPC_Ensemble &ens, bool min_lut, double sigma2, int max_iters,
                       const bvec &reuse_vec, int Nq_Cha, const ivec &Nq_Msg, bool allow_degree_one = false);
    // design cache (LUTLDPC_DESIGN_CACHE=<dir>, see design_luts): did the last design_luts come from it?
    bool design_came_from_cache() const { return design_from_cache; }
    void set_exit_conditions(int max_iters, bool syndr_check_each_iter = true, bool syndr_check_at_start = false);

    void encode(const bvec &input, bvec &output);
    bvec encode(const bvec &input);
    void decode(const vec &llr_in, bvec &syst_bits);              // one frame (B = 1 on the device)
    bvec decode(const vec &llr_in);
    int lut_decode(const ivec &LLRin_cha, const ivec &LLRin_msg, bvec &LLRout);
    bool syndrome_check(const bvec &b) const;                      // host-side utility (encoder check)

    // ---- batched entry points (frames are rows) ---------------------------------------------
    // llr[B*nvar] -> bits[B*nvar] (all code bits; the systematic part is the first get_ninfo())
    void decode_batch(const double *llr, int B, uint8_t *bits, int32_t *iters);
    void lut_decode_batch(const uint8_t *cha, const uint8_t *msg0, int B, uint8_t *bits, int32_t *iters);
    // the same with the message dumps of output_verbosity = level (2 or 3; src/LDPC_Code_LUT.cpp:292-298,311-317,331-337) written
    // to `os`, frame after frame, as the reference streams them to std::cout; lut_decode_batch calls it with std::cout when
    // set_output_verbosity(>= 2) is in force
    void lut_decode_batch_dump(const uint8_t *cha, const uint8_t *msg0, int B, uint8_t *bits, int32_t *iters, int level, std::ostream &os);
    lutldpc_decoder *device_handle();        // creates the HIP decoder on first use
    void set_device(int device);             // default 0; -1 = host-only (set-up without a GPU)

    double get_rate() const { return 1.0 - static_cast<double>(nchk_lin_indep) / nvar; }
    int get_nvar() const { return nvar; }
    int get_nchk() const { return nchk; }
    int get_ncheck() const { return nchk; }
    int get_nchk_lin_indep() const { return nchk_lin_indep; }
    int get_ncheck_lin_indep() const { return nchk_lin_indep; }
    int get_ninfo() const { return nvar - nchk_lin_indep; }
    int get_nrof_iterations() const { return max_iters; }
    void set_output_verbosity(int v) { output_verbosity = v; }
    int get_output_verbosity() const { return output_verbosity; }
    void print_stimuli(const uint8_t *cha, const uint8_t *bits) const;   // src/LDPC_Code_LUT.cpp:228-238
    void set_initial_message_mode(initial_message_mode_t m) { initial_message_mode = m; }

    void save_code(const std::string &filename) const;             // src/LDPC_Code_LUT.cpp:643-697
    void load_code(const std::string &filename, LDPC_Generator *G = nullptr);   // :568-641
    friend std::ostream &operator<<(std::ostream &os, const LDPC_Code_LUT &C);

    // read access for the driver / tests (friend class LDPC_BER_Sim_LUT in the reference)
    const ivec &get_dv_vec() const { return dv_vec; }
    const ivec &get_dc_vec() const { return dc_vec; }
    const ivec &get_cn_msg_idx() const { return cn_msg_idx; }
    const std::vector<ivec> &get_chk_equ_idx() const { return chk_equ_idx; }
    const vec &get_qb_Cha() const { return qb_Cha; }
    const vec &get_qb_Msg() const { return qb_Msg; }
    const ivec &get_Nq_Cha_2_Nq_Msg_map() const { return Nq_Cha_2_Nq_Msg_map; }
    const LUT_Tree_Array &get_var_trees() const { return var_trees; }
    const LUT_Tree_Array &get_chk_trees() const { return chk_trees; }
    int get_Nq_Cha() const { return Nq_Cha; }
    bool get_minLUT() const { return minLUT; }
    bool get_psc() const { return psc; }
    bool get_pisc() const { return pisc; }
    initial_message_mode_t get_initial_message_mode() const { return initial_message_mode; }
    ivec Nq_Msg;          // public in the reference through the friend driver (src/LDPC_BER_Sim.cpp:543-544)
    bvec reuse_vec;
    void set_nchk_lin_indep(int r) { nchk_lin_indep = r; }   // for codes whose rank is known offline

protected:
    void decoder_parameterization(const LDPC_Parity *H, int known_rank = 0);   // src/LDPC_Code_LUT.cpp:488-541
    void integrity_check();

private:
    void drop_device();
    bool load_design(const std::string &path, const std::string &key, LUT_Tree_Array &var_luts, LUT_Tree_Array &chk_luts);
    void store_design(const std::string &path, const std::string &key) const;
    bool design_from_cache = false;

    bool H_defined = false, G_defined = false, LUTs_defined = false, minLUT = false;
    int nvar = 0, nchk = 0, nchk_lin_indep = 0;
    LDPC_Generator *G = nullptr;
    int max_iters = 50;
    bool psc = true, pisc = false;
    ivec dv_vec, dc_vec;
    int num_edges = 0;
    ivec cn_msg_idx;
    std::vector<ivec> chk_equ_idx;
    LUT_Tree_Array var_trees, chk_trees;
    int Nq_Cha = 0;
    vec qb_Cha, qb_Msg;
    ivec Nq_Cha_2_Nq_Msg_map;
    int output_verbosity = 0;
    initial_message_mode_t initial_message_mode = CONT;
    lutldpc_decoder *dev = nullptr;
    int device = 0;
};

std::ostream &operator<<(std::ostream &os, const LDPC_Code_LUT &C);

}  // namespace lut_ldpc
