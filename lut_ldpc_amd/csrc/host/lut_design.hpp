// lut_design.hpp -- LUT design by discrete density evolution (set-up path of ber_sim).
//
// Mirrors what LDPC_Code_LUT::design_luts needs from the reference's src/LDPC_Ensemble.* and
// src/LDPC_DE.*: LDPC_Ensemble (degree distributions, empirical ensemble of a parity-check
// matrix), the tree-template factory get_lut_tree_templates and LDPC_DE_LUT with evolve(),
// get_quant_bound(), get_lut_trees() and bisec_search().  Out of scope (threshold analysis
// tooling not used by ber_sim): LDPC_DE_BP, evolve_adaptive_reuse, the lambda2-stability bounds.
#pragma once
#include "ldpc_parity.hpp"
#include "lut_tree.hpp"

#include <string>
#include <vector>

namespace lut_ldpc {

class LDPC_Ensemble {
public:
    LDPC_Ensemble() = default;
    // edge-perspective pmfs indexed by degree-1 (src/LDPC_Ensemble.cpp:46-51)
    LDPC_Ensemble(const vec &l, const vec &r);
    // explicit active degrees (src/LDPC_Ensemble.cpp:134-148)
    LDPC_Ensemble(const ivec &dl, const vec &l, const ivec &dr, const vec &r);
    double get_rate() const;                                   // :320-322
    int get_dv_act() const { return (int)degree_lam.size(); }
    int get_dc_act() const { return (int)degree_rho.size(); }
    const vec &sget_lam() const { return lam; }
    const vec &sget_rho() const { return rho; }
    const ivec &sget_degree_lam() const { return degree_lam; }
    const ivec &sget_degree_rho() const { return degree_rho; }
private:
    void normalize();
    ivec degree_lam, degree_rho;
    vec lam, rho;
};

LDPC_Ensemble get_empirical_ensemble(const LDPC_Parity &H);     // src/LDPC_Ensemble.cpp:391-423

// src/LDPC_DE.cpp:1124-1290.  tree_method: "auto_bin_balanced" | "auto_bin_high" | "root_only" |
// "filename=<ini>".  allow_degree_one: see LUT_Tree (DESIGN.md "deviations").
void get_lut_tree_templates(const std::string &tree_method, const LDPC_Ensemble &ens, const ivec &Nq_Msg, int Nq_Cha,
                            bool minLUT, LUT_Tree_Array &var_luts, LUT_Tree_Array &chk_luts, bool allow_degree_one = false);

class LDPC_DE_LUT {
public:
    enum { INDIVIDUAL, JOINT_LEVEL, JOINT_ROOT };
    // defaults as src/LDPC_DE.hpp:134-140
    LDPC_DE_LUT(const LDPC_Ensemble &ens, int Nq_Cha, const ivec &Nq_Msg_vec, int maxiter_de,
                const LUT_Tree_Array &var_tree_templates, const LUT_Tree_Array &chk_tree_templates,
                const bvec &reuse_vec = bvec(), double thr_prec = 1e-6, double Pe_max = 1e-9, int maxiter_bisec = 30,
                double LLR_max = 25, int Nq_fine = 5000, const std::string &irregular_design_strategy = "joint_root");

    // src/LDPC_DE.cpp:198-326; return value as the reference (iteration of convergence, -1, or
    // the iteration count when save_luts)
    int evolve(double thr, bool save_luts, LUT_Tree_Array &var_trees, LUT_Tree_Array &chk_trees);
    int evolve(double thr);
    void get_quant_bound(double sig, vec &qb_Cha, vec &qb_Msg) const;        // :561-601
    void get_lut_trees(LUT_Tree_Array &var_trees, LUT_Tree_Array &chk_trees, double sig);   // :607-612
    void set_bisec_window(double tmin, double tmax) { thr_min = tmin; thr_max = tmax; }
    void set_exit_conditions(int maxiter_de_, int maxiter_bisec_, int max_ni_de_iters_, double Pe_max_, double thr_prec_);
    int bisec_search(double &thr);                                           // :49-96 (arithmetic mean)

private:
    void set_channel_pmf(double sig);                                        // :400-412
    void chk_update_irr(int iter, std::vector<LUT_Tree> &prev);              // :414-489
    void var_update_irr(int iter, std::vector<LUT_Tree> &prev);              // :494-558
    void lut_update_irr(int iter, std::vector<LUT_Tree> &prev, const LUT_Tree_Array &templates, const vec &dist,
                        const vec &p_msg, int Nq_in, int Nq_out, vec &acc);

    LDPC_Ensemble ens;
    int Nq_Cha, maxiter_de, maxiter_bisec, Nq_fine, max_ni_de_iters = 1, strategy;
    ivec Nq_Msg_vec;
    bvec reuse_vec;
    double thr_prec, Pe_max, LLR_max, thr_min, thr_max;
    bool min_lut;
    LUT_Tree_Array var_tree_templates, chk_tree_templates;
    vec pmf_cha, pmf_var2chk, pmf_chk2var;
};

// joint designs over the degree classes (src/LDPC_DE.cpp:1293-1466)
void joint_root_irr_lut_design(const vec &degree_dist, std::vector<LUT_Tree> &lut_trees);
void joint_level_irr_lut_design(const vec &degree_dist, std::vector<LUT_Tree> &lut_trees);

}  // namespace lut_ldpc
