// lut_tree.hpp -- LUT_Tree / LUT_Tree_Node: tree-structured node-update descriptions.
//
// Mirrors the public surface of the reference's src/LUT_Tree.hpp:47-155,184-333 that the decode
// path needs: template parsing and generation, resolution / leaf-pmf set-up, the DE-side local
// quantiser design (`update`) and the text (de)serialisation that doubles as the wire format of
// the HIP decoder's C-ABI.  What is deliberately NOT here: per-node evaluation
// (var_msg_update / chk_msg_update / dec_update) -- that is the hot path and lives in the HIP
// kernels (lut_ldpc_amd/csrc/hip); TikZ drawing (plotting aid, out of scope).
#pragma once
#include "numerics.hpp"

#include <deque>
#include <iosfwd>
#include <memory>
#include <string>
#include <vector>

namespace lut_ldpc {

class LUT_Tree_Node {
public:
    enum node_type_t { IM, ROOT, MSG, CHA, num_node_types };     // src/LUT_Tree.hpp:188-193
    node_type_t type;
    int K = 0;            // number of output labels
    ivec Q;               // half of the symmetric map (src/LUT_Tree.cpp:414-417)
    vec p;                // pmf of the node's output during design
    std::vector<std::unique_ptr<LUT_Tree_Node>> children;

    explicit LUT_Tree_Node(node_type_t t) : type(t) {}
    bool is_leaf() const { return type == MSG || type == CHA; }
    std::unique_ptr<LUT_Tree_Node> deep_copy() const;
    int get_num_leaves() const;
    int get_height() const;
    void get_level_nodes(int req_level, int cur_level, std::deque<LUT_Tree_Node *> &out);
};

class LUT_Tree {
public:
    enum tree_type_t { VARTREE, CHKTREE, DECTREE, num_tree_types };   // src/LUT_Tree.hpp:50-54

    LUT_Tree() = default;
    // from a template string over r i m c / (src/LUT_Tree.cpp:579-592)
    LUT_Tree(const std::string &tree_string, tree_type_t t);
    // auto-generated: "auto_bin_balanced" | "auto_bin_high" | "root_only" (src/LUT_Tree.cpp:594-630);
    // allow_degree_one enables the ROOT(CHA) tree for degree-1 variable nodes, which the
    // reference asserts on (src/LUT_Tree.cpp:202; DESIGN.md "deviations")
    LUT_Tree(int num_leaves, tree_type_t t, const std::string &mode = "auto_bin_balanced", bool allow_degree_one = false);
    LUT_Tree(const LUT_Tree &o);
    LUT_Tree(LUT_Tree &&) = default;
    LUT_Tree &operator=(LUT_Tree o) { swap(*this, o); return *this; }
    static void swap(LUT_Tree &a, LUT_Tree &b);

    int get_num_leaves() const { return num_leaves; }
    tree_type_t get_type() const { return type; }
    int get_height() const { return root ? root->get_height() : 0; }
    bool empty() const { return !root; }
    std::string gen_template_string() const;
    void set_resolution(int Nq_in, int Nq_out, int Nq_cha = 0);      // src/LUT_Tree.cpp:296-306
    void set_leaves(const vec &p_Msg, const vec &p_Cha);             // src/LUT_Tree.cpp:92-103
    vec update(bool reuse = false);                                  // src/LUT_Tree.cpp:683-698
    void reset_pmfs();
    std::deque<LUT_Tree_Node *> get_level_nodes(int level);
    LUT_Tree_Node *get_root() { return root.get(); }
    const LUT_Tree_Node *get_root() const { return root.get(); }

    // per-node design steps (src/LUT_Tree.cpp:709-766)
    static void var_update(vec &p_out, ivec &Q_out, const std::vector<vec> &p_in, int Nq, bool reuse);
    static void chk_update(vec &p_out, ivec &Q_out, const std::vector<vec> &p_in, int Nq, bool reuse);
    // product pmf of a node's children (src/LUT_Tree.cpp:537-561)
    static vec get_input_product_pmf(const LUT_Tree_Node &n, tree_type_t t);

    friend std::ostream &operator<<(std::ostream &os, const LUT_Tree &t);
    friend std::istream &operator>>(std::istream &is, LUT_Tree &t);

private:
    tree_type_t type = VARTREE;
    int num_leaves = 0;
    std::unique_ptr<LUT_Tree_Node> root;
};

// Array<Array<LUT_Tree>> of the reference: [tree set][degree class]
using LUT_Tree_Array = std::vector<std::vector<LUT_Tree>>;
std::ostream &operator<<(std::ostream &os, const LUT_Tree &t);
std::istream &operator>>(std::istream &is, LUT_Tree &t);
std::ostream &operator<<(std::ostream &os, const LUT_Tree_Array &a);     // src/LUT_Tree.cpp:855-864
std::istream &operator>>(std::istream &is, LUT_Tree_Array &a);           // src/LUT_Tree.cpp:893-927
std::string to_string(const LUT_Tree_Array &a);

// Symmetric quantiser for a product pmf after removing label pairs of zero mass
// (shared by LUT_Tree::var_update and the joint designs, src/LUT_Tree.cpp:725-737,
// src/LDPC_DE.cpp:1428-1440).  Returns the full-length map.
ivec design_quantizer_skip_zero_mass(vec &p_out, const vec &prod, int Nq);

}  // namespace lut_ldpc
