// lut_design.cpp -- see lut_design.hpp.  Reference: src/LDPC_Ensemble.cpp, src/LDPC_DE.cpp.
#include "lut_design.hpp"
#include "ini.hpp"

#include <cmath>
#include <cstdio>
#include <stdexcept>

namespace lut_ldpc {

// ------------------------------------------------------------------ LDPC_Ensemble
namespace {
void active_degrees(const vec &pmf, ivec &deg, vec &w) {     // set_*_degree_dist, :53-91
    deg.clear(); w.clear();
    for (size_t i = 0; i < pmf.size(); i++) if (pmf[i] > 0) { deg.push_back((int)i + 1); w.push_back(pmf[i]); }
}
}  // namespace

void LDPC_Ensemble::normalize() {      // check_consistency, :119-128
    const double sl = sum(lam), sr = sum(rho);
    if (!(std::abs(1.0 - sl) < 1e-2 || std::abs(1.0 - sr) < 1e-2))
        throw std::invalid_argument("LDPC_Ensemble::check_consistency(): degree distributions do not sum to one");
    for (double &x : lam) x = x / sl;
    for (double &x : rho) x = x / sr;
    if (!(get_rate() > 0)) throw std::invalid_argument("LDPC_Ensemble::check_consistency(): code rate is not positive");
}

LDPC_Ensemble::LDPC_Ensemble(const vec &l, const vec &r) {
    active_degrees(r, degree_rho, rho);
    active_degrees(l, degree_lam, lam);
    normalize();
}

LDPC_Ensemble::LDPC_Ensemble(const ivec &dl, const vec &l, const ivec &dr, const vec &r) : degree_lam(dl), degree_rho(dr), lam(l), rho(r) {
    if (dl.size() != l.size() || dr.size() != r.size()) throw std::invalid_argument("LDPC_Ensemble: input dimension mismatch");
    for (int d : dl) if (d < 1) throw std::invalid_argument("LDPC_Ensemble: degrees must be larger than 0");
    for (int d : dr) if (d < 1) throw std::invalid_argument("LDPC_Ensemble: degrees must be larger than 0");
    normalize();
}

double LDPC_Ensemble::get_rate() const {
    double a = 0, b = 0;
    for (size_t i = 0; i < rho.size(); i++) a += rho[i] / degree_rho[i];
    for (size_t i = 0; i < lam.size(); i++) b += lam[i] / degree_lam[i];
    return 1 - a / b;
}

LDPC_Ensemble get_empirical_ensemble(const LDPC_Parity &H) {
    const size_t max_degree = 200;
    vec var_edge(max_degree, 0.0), chk_edge(max_degree, 0.0);
    for (int w : H.get_colsum()) {
        if (w < 1 || (size_t)w > max_degree) throw std::invalid_argument("get_empirical_ensemble(): degree outside [1,200]");
        var_edge[(size_t)w - 1] += w;
    }
    for (int w : H.get_rowsum()) {
        if (w < 1 || (size_t)w > max_degree) throw std::invalid_argument("get_empirical_ensemble(): degree outside [1,200]");
        chk_edge[(size_t)w - 1] += w;
    }
    const double sv = sum(var_edge), sc = sum(chk_edge);
    for (double &x : var_edge) x = x / sv;
    for (double &x : chk_edge) x = x / sc;
    return LDPC_Ensemble(var_edge, chk_edge);
}

// ------------------------------------------------------------------ templates
void get_lut_tree_templates(const std::string &tree_method, const LDPC_Ensemble &ens, const ivec &Nq_Msg, int Nq_Cha,
                            bool minLUT, LUT_Tree_Array &var_luts, LUT_Tree_Array &chk_luts, bool allow_degree_one) {
    const int max_iters = (int)Nq_Msg.size();
    const size_t eq = tree_method.find('=');
    const std::string tm = tree_method.substr(0, eq);
    const std::string filename = eq == std::string::npos ? "" : tree_method.substr(eq + 1);
    const ivec &var_deg = ens.sget_degree_lam(), &chk_deg = ens.sget_degree_rho();
    const size_t dv_act = var_deg.size(), dc_act = chk_deg.size();
    var_luts.clear(); chk_luts.clear();
    auto key = [](const char *prefix, int n) { char b[32]; std::snprintf(b, sizeof b, "%s%03d", prefix, n); return std::string(b); };

    if (tm == "filename") {
        if (filename.empty()) throw std::invalid_argument("get_lut_tree_templates(): use 'filename=<name of file>'");
        if (max_iters < 2) throw std::invalid_argument("get_lut_tree_templates(): file trees need at least 2 iterations");
        Ini ini(filename);
        if (!ini.has_section("var_iter_000")) throw std::runtime_error("get_lut_tree_templates(): error reading variable node tree from file");
        if (!ini.has_section("DT")) throw std::runtime_error("get_lut_tree_templates(): error reading decision node tree from file");
        var_luts.assign((size_t)max_iters, {});
        for (int ii = 0; ii < max_iters - 1; ii++) {
            auto &row = var_luts[(size_t)ii];
            const std::string sec = key("var_iter_", ii);
            if (ii > 0 && !ini.has_section(sec)) { row = var_luts[(size_t)ii - 1]; continue; }   // inherit, :1186-1190
            row.resize(dv_act);
            for (size_t dd = 0; dd < dv_act; dd++) {
                auto s = ini.get_optional(sec, key("var_deg_", var_deg[dd]));
                if (!s) throw std::runtime_error("get_lut_tree_templates(): no tree string for variable node degree " + std::to_string(var_deg[dd]) + " at iteration " + std::to_string(ii));
                row[dd] = LUT_Tree(*s, LUT_Tree::VARTREE);
                if (row[dd].get_num_leaves() != var_deg[dd]) throw std::runtime_error("get_lut_tree_templates(): LUT tree does not match node degree");
                row[dd].set_resolution(Nq_Msg[(size_t)ii], Nq_Msg[(size_t)ii + 1], Nq_Cha);
            }
        }
        auto &dt = var_luts[(size_t)max_iters - 1];
        dt.resize(dv_act);
        for (size_t dd = 0; dd < dv_act; dd++) {
            auto s = ini.get_optional("DT", key("var_deg_", var_deg[dd]));
            if (!s) throw std::runtime_error("get_lut_tree_templates(): no decision tree string for variable node degree " + std::to_string(var_deg[dd]));
            dt[dd] = LUT_Tree(*s, LUT_Tree::DECTREE);
            if (dt[dd].get_num_leaves() != var_deg[dd] + 1) throw std::runtime_error("get_lut_tree_templates(): LUT tree does not match node degree");
            dt[dd].set_resolution(Nq_Msg[0], Nq_Msg[1], Nq_Cha);
        }
        if (!minLUT) {
            if (!ini.has_section("chk_iter_000")) throw std::runtime_error("get_lut_tree_templates(): error reading check node tree from file");
            chk_luts.assign((size_t)max_iters, {});
            for (int ii = 0; ii < max_iters; ii++) {
                auto &row = chk_luts[(size_t)ii];
                const std::string sec = key("chk_iter_", ii);
                // the reference copies dv_act entries here (:1243); the check classes are meant
                if (ii > 0 && !ini.has_section(sec)) { row = chk_luts[(size_t)ii - 1]; continue; }
                row.resize(dc_act);
                for (size_t dd = 0; dd < dc_act; dd++) {
                    auto s = ini.get_optional(sec, key("chk_deg_", chk_deg[dd]));
                    if (!s) throw std::runtime_error("get_lut_tree_templates(): no tree string for check node degree " + std::to_string(chk_deg[dd]) + " at iteration " + std::to_string(ii));
                    row[dd] = LUT_Tree(*s, LUT_Tree::CHKTREE);
                    if (row[dd].get_num_leaves() != chk_deg[dd] - 1) throw std::runtime_error("get_lut_tree_templates(): LUT tree does not match node degree");
                    row[dd].set_resolution(Nq_Msg[(size_t)ii], Nq_Msg[(size_t)std::min(ii + 1, max_iters - 1)], Nq_Cha);
                }
            }
        }
        return;
    }
    if ((tm == "auto_bin_balanced" || tm == "auto_bin_high" || tm == "root_only") && filename.empty()) {
        var_luts.assign((size_t)max_iters, std::vector<LUT_Tree>(dv_act));
        for (int ii = 0; ii < max_iters; ii++)
            for (size_t dd = 0; dd < dv_act; dd++) {
                LUT_Tree &t = var_luts[(size_t)ii][dd];
                if (ii == max_iters - 1) {
                    t = LUT_Tree(var_deg[dd] + 1, LUT_Tree::DECTREE, tm);
                    t.set_resolution(Nq_Msg[(size_t)ii], 2, Nq_Cha);
                } else {
                    t = LUT_Tree(var_deg[dd], LUT_Tree::VARTREE, tm, allow_degree_one);
                    t.set_resolution(Nq_Msg[(size_t)ii], Nq_Msg[(size_t)ii + 1], Nq_Cha);
                }
            }
        if (!minLUT) {
            chk_luts.assign((size_t)max_iters, std::vector<LUT_Tree>(dc_act));
            for (int ii = 0; ii < max_iters; ii++)
                for (size_t dd = 0; dd < dc_act; dd++) {
                    chk_luts[(size_t)ii][dd] = LUT_Tree(chk_deg[dd] - 1, LUT_Tree::CHKTREE, tm);
                    chk_luts[(size_t)ii][dd].set_resolution(Nq_Msg[(size_t)ii], Nq_Msg[(size_t)ii]);
                }
        }
        return;
    }
    throw std::invalid_argument("Could not parse tree_method " + tree_method);
}

// ------------------------------------------------------------------ joint designs
namespace {

// src/LDPC_DE.cpp:1379-1466: one quantiser for all listed nodes, weighted by degree
// distribution and by the node's share of the leaves at its level
void level_lut_tree_update(std::vector<std::deque<LUT_Tree_Node *>> &tree_nodes, const vec &degree_dist, LUT_Tree::tree_type_t t) {
    const size_t L = tree_nodes.size();
    std::vector<std::vector<vec>> prod(L);
    std::vector<vec> weight(L);
    size_t M_tot = 0;
    int n_out = -1;
    for (size_t ll = 0; ll < L; ll++) {
        for (LUT_Tree_Node *n : tree_nodes[ll]) {
            if (n_out == -1) n_out = n->K;
            weight[ll].push_back(n->get_num_leaves());
            prod[ll].push_back(LUT_Tree::get_input_product_pmf(*n, t));
            M_tot += prod[ll].back().size();
        }
        const double ws = sum(weight[ll]);
        for (double &w : weight[ll]) w = w / ws;
    }
    if (n_out < 0) return;     // no node at this level
    vec overall(M_tot, -1e9);
    size_t I = 0;
    for (size_t ll = 0; ll < L; ll++)
        for (size_t jj = 0; jj < prod[ll].size(); jj++) {
            const vec &p = prod[ll][jj];
            const size_t M = p.size();
            for (size_t mm = 0; mm < M / 2; mm++) {
                overall[I + mm] = weight[ll][jj] * degree_dist[ll] * p[mm];
                overall[M_tot - 1 - I - mm] = weight[ll][jj] * degree_dist[ll] * p[M - 1 - mm];
            }
            I += M / 2;
        }
    const double s = sum(overall);
    for (double &x : overall) x = x / s;
    vec p_out;
    const ivec Q_all = design_quantizer_skip_zero_mass(p_out, overall, n_out);
    I = 0;
    for (size_t ll = 0; ll < L; ll++)
        for (size_t jj = 0; jj < prod[ll].size(); jj++) {
            LUT_Tree_Node *n = tree_nodes[ll][jj];
            const vec &p = prod[ll][jj];
            const size_t M = p.size();
            n->Q.assign(Q_all.begin() + (long)I, Q_all.begin() + (long)(I + M / 2));
            I += M / 2;
            n->p.assign((size_t)n_out, 0.0);
            for (size_t mm = 0; mm < M; mm++) {
                if (mm < M / 2) n->p[(size_t)n->Q[mm]] += p[mm];
                else n->p[(size_t)(n_out - 1 - n->Q[M - 1 - mm])] += p[mm];
            }
        }
}

}  // namespace

void joint_root_irr_lut_design(const vec &degree_dist, std::vector<LUT_Tree> &trees) {   // :1345-1377
    if (degree_dist.size() != trees.size()) throw std::invalid_argument("joint_root_irr_lut_design(): input dimension mismatch");
    for (LUT_Tree &t : trees) (void)t.update();
    std::vector<std::deque<LUT_Tree_Node *>> roots(trees.size());
    for (size_t ll = 0; ll < trees.size(); ll++) roots[ll] = trees[ll].get_level_nodes(0);
    level_lut_tree_update(roots, degree_dist, trees[0].get_type());
}

void joint_level_irr_lut_design(const vec &degree_dist, std::vector<LUT_Tree> &trees) {  // :1293-1343
    if (degree_dist.size() != trees.size()) throw std::invalid_argument("joint_level_irr_lut_design(): input dimension mismatch");
    int deepest = 0;
    for (LUT_Tree &t : trees) deepest = std::max(deepest, t.get_height());
    for (int level = deepest - 1; level >= 0; level--) {
        std::vector<std::deque<LUT_Tree_Node *>> nodes(trees.size());
        for (size_t ll = 0; ll < trees.size(); ll++) {
            if (trees[ll].get_height() <= level) continue;
            for (LUT_Tree_Node *n : trees[ll].get_level_nodes(level))
                if (n->type == LUT_Tree_Node::IM || n->type == LUT_Tree_Node::ROOT) nodes[ll].push_back(n);
        }
        level_lut_tree_update(nodes, degree_dist, trees[0].get_type());
    }
}

// ------------------------------------------------------------------ LDPC_DE_LUT
LDPC_DE_LUT::LDPC_DE_LUT(const LDPC_Ensemble &ens_, int Nq_Cha_, const ivec &Nq_Msg_vec_, int maxiter_de_,
                         const LUT_Tree_Array &var_t, const LUT_Tree_Array &chk_t, const bvec &reuse_vec_, double thr_prec_,
                         double Pe_max_, int maxiter_bisec_, double LLR_max_, int Nq_fine_, const std::string &strategy_)
    : ens(ens_), Nq_Cha(Nq_Cha_), maxiter_de(maxiter_de_), maxiter_bisec(maxiter_bisec_), Nq_fine(Nq_fine_), Nq_Msg_vec(Nq_Msg_vec_),
      reuse_vec(reuse_vec_.empty() ? bvec((size_t)maxiter_de_, 0) : reuse_vec_), thr_prec(thr_prec_), Pe_max(Pe_max_), LLR_max(LLR_max_),
      min_lut(chk_t.empty()), var_tree_templates(var_t), chk_tree_templates(chk_t) {
    thr_max = rate_to_shannon_thr(ens.get_rate());
    thr_min = thr_max * 1e-4;
    if (strategy_ == "individual") strategy = INDIVIDUAL;
    else if (strategy_ == "joint_level") strategy = JOINT_LEVEL;
    else if (strategy_ == "joint_root") strategy = JOINT_ROOT;
    else throw std::invalid_argument("Irregular Design Strategy " + strategy_ + " unknown!");
    if ((int)Nq_Msg_vec.size() < maxiter_de || (int)reuse_vec.size() < maxiter_de)
        throw std::invalid_argument("LDPC_DE_LUT: resolution / reuse vectors shorter than maxiter_de");
}

void LDPC_DE_LUT::set_exit_conditions(int maxiter_de_, int maxiter_bisec_, int max_ni_de_iters_, double Pe_max_, double thr_prec_) {
    maxiter_de = maxiter_de_; maxiter_bisec = maxiter_bisec_; max_ni_de_iters = max_ni_de_iters_; Pe_max = Pe_max_; thr_prec = thr_prec_;
}

void LDPC_DE_LUT::set_channel_pmf(double sig) {
    const double delta = 2 * LLR_max / Nq_fine;
    const vec fine = get_gaussian_pmf(2 / (sig * sig), 2 / sig, Nq_fine, delta);
    ivec Q;
    (void)quant_mi_sym(pmf_cha, Q, fine, Nq_Cha, true);
    (void)quant_mi_sym(pmf_var2chk, Q, fine, Nq_Msg_vec[0], true);
}

void LDPC_DE_LUT::get_quant_bound(double sig, vec &qb_Cha, vec &qb_Msg) const {
    const double delta = 2 * LLR_max / Nq_fine;
    const vec fine = get_gaussian_pmf(2 / (sig * sig), 2 / sig, Nq_fine, delta);
    const int M = Nq_fine;
    auto bounds = [&](int K) {
        vec p; ivec Q;
        (void)quant_mi_sym(p, Q, fine, K, true);
        vec pos((size_t)(K / 2 - 1), 0.0);
        int label = 0;
        for (int mm = 0; mm < M / 2; mm++)
            if (Q[(size_t)(M - M / 2 + mm)] - K / 2 > label) { pos[(size_t)label] = mm * delta; label++; }
        vec qb;
        for (size_t i = pos.size(); i-- > 0;) qb.push_back(-pos[i]);
        qb.push_back(0.0);
        qb.insert(qb.end(), pos.begin(), pos.end());
        return qb;
    };
    qb_Cha = bounds(Nq_Cha);
    qb_Msg = bounds(Nq_Msg_vec[0]);
}

void LDPC_DE_LUT::lut_update_irr(int iter, std::vector<LUT_Tree> &prev, const LUT_Tree_Array &templates, const vec &dist,
                                 const vec &p_msg, int Nq_in, int Nq_out, vec &acc) {
    const size_t L = dist.size();
    auto accumulate = [&](bool reuse) {
        for (size_t dd = 0; dd < L; dd++) {
            const vec p = prev[dd].update(reuse);
            for (size_t i = 0; i < p.size(); i++) acc[i] = acc[i] + dist[dd] * p[i];
        }
    };
    if (reuse_vec[(size_t)iter]) {
        for (size_t dd = 0; dd < L; dd++) prev[dd].set_leaves(p_msg, pmf_cha);
        accumulate(true);
        return;
    }
    prev.resize(L);
    for (size_t dd = 0; dd < L; dd++) {
        LUT_Tree t = templates[(size_t)iter][dd];
        t.set_leaves(p_msg, pmf_cha);
        t.set_resolution(Nq_in, Nq_out, Nq_Cha);
        prev[dd] = std::move(t);
    }
    if (strategy == INDIVIDUAL) { accumulate(false); return; }
    if (strategy == JOINT_LEVEL) joint_level_irr_lut_design(dist, prev);
    else joint_root_irr_lut_design(dist, prev);
    std::fill(acc.begin(), acc.end(), 0.0);
    accumulate(true);
}

void LDPC_DE_LUT::chk_update_irr(int iter, std::vector<LUT_Tree> &prev) {
    pmf_chk2var.assign((size_t)Nq_Msg_vec[(size_t)iter], 0.0);
    if (min_lut) {
        const vec &rho = ens.sget_rho();
        for (size_t dd = 0; dd < rho.size(); dd++) {
            const vec p = chk_update_minsum(pmf_var2chk, ens.sget_degree_rho()[dd]);
            for (size_t i = 0; i < p.size(); i++) pmf_chk2var[i] = pmf_chk2var[i] + rho[dd] * p[i];
        }
        return;
    }
    const vec p_in = pmf_var2chk;
    lut_update_irr(iter, prev, chk_tree_templates, ens.sget_rho(), p_in, Nq_Msg_vec[(size_t)iter], Nq_Msg_vec[(size_t)iter], pmf_chk2var);
}

void LDPC_DE_LUT::var_update_irr(int iter, std::vector<LUT_Tree> &prev) {
    pmf_var2chk.assign((size_t)Nq_Msg_vec[(size_t)iter + 1], 0.0);
    lut_update_irr(iter, prev, var_tree_templates, ens.sget_lam(), pmf_chk2var, Nq_Msg_vec[(size_t)iter], Nq_Msg_vec[(size_t)iter + 1], pmf_var2chk);
}

int LDPC_DE_LUT::evolve(double thr) {
    LUT_Tree_Array a, b;
    return evolve(thr, false, a, b);
}

int LDPC_DE_LUT::evolve(double thr, bool save_luts, LUT_Tree_Array &var_trees, LUT_Tree_Array &chk_trees) {
    // the output of the last variable node update is binary (:203); restored on exit
    struct Restore { ivec &v; ~Restore() { v.pop_back(); } } restore{Nq_Msg_vec};
    Nq_Msg_vec.push_back(2);
    double Pe_old = 1.0;
    int ni_iters = 0;
    set_channel_pmf(thr);
    std::vector<LUT_Tree> var_iter, chk_iter;
    if (save_luts) { var_trees.clear(); chk_trees.clear(); }
    const int max_iter = save_luts ? maxiter_de : maxiter_de - 1;
    for (int ii = 0; ii < max_iter; ii++) {
        double Pe = 0;
        for (int k = 0; k < Nq_Msg_vec[(size_t)ii] / 2; k++) Pe += pmf_var2chk[(size_t)k];
        if (Pe < Pe_max && !save_luts) return ii;
        if (Pe <= Pe_old) Pe_old = Pe; else ni_iters++;
        if (ni_iters >= max_ni_de_iters && !save_luts) return -1;
        chk_update_irr(ii, chk_iter);
        var_update_irr(ii, var_iter);
        if (save_luts && !reuse_vec[(size_t)ii]) {
            var_trees.push_back(var_iter);
            if (!min_lut) chk_trees.push_back(chk_iter);
        }
    }
    if (!save_luts) return -1;
    for (auto &row : var_trees) for (auto &t : row) t.reset_pmfs();
    for (auto &row : chk_trees) for (auto &t : row) t.reset_pmfs();
    return max_iter;
}

void LDPC_DE_LUT::get_lut_trees(LUT_Tree_Array &var_trees, LUT_Tree_Array &chk_trees, double sig) {
    if (reuse_vec[0]) throw std::invalid_argument("LDPC_DE_LUT::get_lut_trees(): reuse not possible for the initial iteration");
    // a reused stage reads and writes the alphabets of the stage it repeats: the reference adds probability vectors of different
    // lengths otherwise (src/LDPC_DE.cpp:434-487,505-557: an IT++ size assertion in a debug build, undefined in a release build)
    for (int i = 1; i < maxiter_de; i++)
        if (reuse_vec[(size_t)i] && (Nq_Msg_vec[(size_t)i] != Nq_Msg_vec[(size_t)i - 1] || (i + 1 < maxiter_de && Nq_Msg_vec[(size_t)i + 1] != Nq_Msg_vec[(size_t)i])))
            throw std::invalid_argument("LDPC_DE_LUT::get_lut_trees(): a reused LUT stage must keep the message alphabets of the stage it repeats");
    (void)evolve(sig, true, var_trees, chk_trees);
}

int LDPC_DE_LUT::bisec_search(double &thr) {
    bool converged = false;
    int ii = 0;
    double sig = -1.0, lo = thr_min, hi = thr_max;
    while (!converged && ii < maxiter_bisec) {
        sig = (hi + lo) / 2;
        const int ach = evolve(sig);
        if ((hi - lo < thr_prec) && ach >= 0) converged = true;
        if (ach >= 0) lo = sig; else hi = sig;
        ii++;
    }
    if (converged) { thr = sig; return ii; }
    thr = 0;
    return -1;
}

}  // namespace lut_ldpc
