// lut_tree.cpp -- see lut_tree.hpp.  Reference: src/LUT_Tree.cpp.
#include "lut_tree.hpp"

#include <istream>
#include <list>
#include <ostream>
#include <sstream>
#include <stdexcept>

namespace lut_ldpc {

using Node = LUT_Tree_Node;
using NodePtr = std::unique_ptr<Node>;

// ------------------------------------------------------------------ LUT_Tree_Node
NodePtr Node::deep_copy() const {
    NodePtr n(new Node(type));
    n->K = K; n->Q = Q; n->p = p;
    for (auto &c : children) n->children.push_back(c->deep_copy());
    return n;
}

int Node::get_num_leaves() const {
    if (is_leaf()) return 1;
    int n = 0;
    for (auto &c : children) n += c->get_num_leaves();
    return n;
}

int Node::get_height() const {
    int h = 0;
    for (auto &c : children) h = std::max(h, c->get_height() + 1);
    return h;
}

void Node::get_level_nodes(int req, int cur, std::deque<Node *> &out) {
    if (req == cur) { out.push_back(this); return; }
    for (auto &c : children) c->get_level_nodes(req, cur + 1, out);
}

// ------------------------------------------------------------------ construction
namespace {

NodePtr parse_template(std::istream &is) {      // src/LUT_Tree.cpp:167-198
    int ch = is.get();
    while (ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n') ch = is.get();
    Node::node_type_t t;
    switch (ch) {
    case EOF: case '/': return nullptr;
    case 'r': t = Node::ROOT; break;
    case 'i': t = Node::IM; break;
    case 'm': t = Node::MSG; break;
    case 'c': t = Node::CHA; break;
    default:
        throw std::invalid_argument("LUT_Tree_Node::parse(): allowed characters are r, i, m, c and /");
    }
    NodePtr n(new Node(t));
    while (NodePtr c = parse_template(is)) n->children.push_back(std::move(c));
    return n;
}

NodePtr leaf(Node::node_type_t t) { return NodePtr(new Node(t)); }

// src/LUT_Tree.cpp:200-237: FIFO pairing of the message leaves
NodePtr gen_bin_balanced(int num_leaves, bool var, bool allow_degree_one) {
    const int n_msg = num_leaves - (var ? 1 : 0);
    if (n_msg == 0 && var && allow_degree_one) {          // degree-1 extension: ROOT(CHA)
        NodePtr r(new Node(Node::ROOT));
        r->children.push_back(leaf(Node::CHA));
        return r;
    }
    if (num_leaves < 2) throw std::invalid_argument("LUT_Tree_Node::gen_bin_balanced_tree(): num_leaves must be >= 2");
    std::list<NodePtr> fifo;
    for (int i = 0; i < n_msg; i++) fifo.push_back(leaf(Node::MSG));
    while (fifo.size() > 1) {
        NodePtr im(new Node(Node::IM));
        im->children.push_back(std::move(fifo.front())); fifo.pop_front();
        im->children.push_back(std::move(fifo.front())); fifo.pop_front();
        fifo.push_back(std::move(im));
    }
    NodePtr top = std::move(fifo.front());
    if (!var) { top->type = Node::ROOT; return top; }
    NodePtr r(new Node(Node::ROOT));
    r->children.push_back(std::move(top));
    r->children.push_back(leaf(Node::CHA));
    return r;
}

// src/LUT_Tree.cpp:240-270: a chain of binary nodes
NodePtr gen_bin_high(int num_leaves, bool var) {
    if (num_leaves < 2) throw std::invalid_argument("LUT_Tree_Node::gen_bin_high_tree(): num_leaves must be >= 2");
    NodePtr root(new Node(Node::ROOT));
    Node *cur = root.get();
    cur->children.push_back(leaf(var ? Node::CHA : Node::MSG));
    for (int todo = num_leaves - 1; todo > 1; todo--) {
        cur->children.insert(cur->children.begin(), NodePtr(new Node(Node::IM)));
        cur = cur->children.front().get();
        cur->children.push_back(leaf(Node::MSG));
    }
    cur->children.push_back(leaf(Node::MSG));
    return root;
}

// src/LUT_Tree.cpp:272-294
NodePtr gen_root_only(int num_leaves, bool var) {
    if (num_leaves < 2) throw std::invalid_argument("LUT_Tree_Node::gen_root_only_tree(): num_leaves must be >= 2");
    NodePtr root(new Node(Node::ROOT));
    for (int i = 0; i < num_leaves - 1; i++) root->children.push_back(leaf(Node::MSG));
    root->children.push_back(leaf(var ? Node::CHA : Node::MSG));
    return root;
}

void template_string(const Node &n, std::string &s) {   // src/LUT_Tree.cpp:142-165
    s += n.type == Node::ROOT ? 'r' : n.type == Node::IM ? 'i' : n.type == Node::MSG ? 'm' : 'c';
    for (auto &c : n.children) template_string(*c, s);
    s += '/';
}

void set_resolution_rec(Node &n, int Nq_in, int Nq_out, int Nq_cha) {
    n.K = n.type == Node::ROOT ? Nq_out : n.type == Node::CHA ? Nq_cha : Nq_in;
    for (auto &c : n.children) set_resolution_rec(*c, Nq_in, Nq_out, Nq_cha);
}

void set_leaves_rec(Node &n, const vec &p_Msg, const vec &p_Cha) {
    if (n.type == Node::MSG) n.p = p_Msg;
    else if (n.type == Node::CHA) n.p = p_Cha;
    else for (auto &c : n.children) set_leaves_rec(*c, p_Msg, p_Cha);
}

void reset_rec(Node &n) {
    for (auto &c : n.children) reset_rec(*c);
    n.p.clear();
}

// src/LUT_Tree.cpp:114-130
const vec &update_rec(Node &n, bool reuse, bool chk) {
    if (n.is_leaf()) return n.p;
    std::vector<vec> pin;
    for (auto &c : n.children) pin.push_back(update_rec(*c, reuse, chk));
    if (chk) LUT_Tree::chk_update(n.p, n.Q, pin, n.K, reuse);
    else LUT_Tree::var_update(n.p, n.Q, pin, n.K, reuse);
    return n.p;
}

void requantize(vec &p_out, const ivec &Q, const vec &prod, int Nq) {     // the `reuse` branches
    const size_t M = prod.size();
    p_out.assign((size_t)Nq, 0.0);
    for (size_t mm = 0; mm < M; mm++) {
        if (mm < M / 2) p_out[(size_t)Q[mm]] += prod[mm];
        else p_out[(size_t)(Nq - 1 - Q[M - 1 - mm])] += prod[mm];
    }
}

void normalize(vec &p) {
    const double s = sum(p);
    for (double &x : p) x = x / s;
}

}  // namespace

ivec design_quantizer_skip_zero_mass(vec &p_out, const vec &prod, int Nq) {
    const size_t M = prod.size();
    std::vector<char> nz(M);
    vec pnz;
    for (size_t mm = 0; mm < M; mm++) {
        nz[mm] = (.5 * (prod[mm] + prod[M - 1 - mm]) != 0);
        if (nz[mm]) pnz.push_back(prod[mm]);
    }
    ivec Qnz;
    (void)quant_mi_sym(p_out, Qnz, pnz, Nq);
    // labels without mass get the least confident outputs, symmetrically
    ivec Q(M);
    for (size_t mm = 0; mm < M; mm++) Q[mm] = mm < M / 2 ? Nq / 2 - 1 : Nq / 2;
    size_t k = 0;
    for (size_t mm = 0; mm < M; mm++) if (nz[mm]) Q[mm] = Qnz[k++];
    return Q;
}

void LUT_Tree::var_update(vec &p_out, ivec &Q_out, const std::vector<vec> &p_in, int Nq, bool reuse) {
    const vec prod = get_var_product_pmf(p_in);
    if (reuse) requantize(p_out, Q_out, prod, Nq);
    else {
        ivec Q = design_quantizer_skip_zero_mass(p_out, prod, Nq);
        Q.resize(Q.size() / 2);
        Q_out.swap(Q);
    }
    normalize(p_out);
}

void LUT_Tree::chk_update(vec &p_out, ivec &Q_out, const std::vector<vec> &p_in, int Nq, bool reuse) {
    const vec prod = get_chk_product_pmf(p_in);
    if (reuse) requantize(p_out, Q_out, prod, Nq);
    else {
        ivec Q;
        (void)quant_mi_sym(p_out, Q, prod, Nq);
        Q.resize(Q.size() / 2);
        Q_out.swap(Q);
    }
    normalize(p_out);
}

vec LUT_Tree::get_input_product_pmf(const Node &n, tree_type_t t) {
    std::vector<vec> pin;
    for (auto &c : n.children) pin.push_back(c->p);
    return t == CHKTREE ? get_chk_product_pmf(pin) : get_var_product_pmf(pin);
}

LUT_Tree::LUT_Tree(const std::string &tree_string, tree_type_t t) : type(t) {
    if (tree_string.find('c') == std::string::npos && t != CHKTREE)
        throw std::invalid_argument("LUT_Tree::LUT_Tree(): trees other than CHKTREE need a channel leaf");
    std::istringstream is(tree_string);
    root = parse_template(is);
    if (!root) throw std::invalid_argument("LUT_Tree::LUT_Tree(): empty tree string");
    num_leaves = root->get_num_leaves();
}

LUT_Tree::LUT_Tree(int l, tree_type_t t, const std::string &m, bool allow_degree_one) : type(t), num_leaves(l) {
    const bool var = (t != CHKTREE);
    if (m == "auto_bin_balanced") root = gen_bin_balanced(l, var, allow_degree_one);
    else if (m == "auto_bin_high") root = gen_bin_high(l, var);
    else if (m == "root_only") root = gen_root_only(l, var);
    else throw std::invalid_argument("LUT_Tree::LUT_Tree(): autogeneration mode " + m + " not supported");
}

LUT_Tree::LUT_Tree(const LUT_Tree &o) : type(o.type), num_leaves(o.num_leaves), root(o.root ? o.root->deep_copy() : nullptr) {}

void LUT_Tree::swap(LUT_Tree &a, LUT_Tree &b) {
    std::swap(a.type, b.type); std::swap(a.num_leaves, b.num_leaves); a.root.swap(b.root);
}

std::string LUT_Tree::gen_template_string() const {
    std::string s;
    if (root) template_string(*root, s);
    return s;
}

void LUT_Tree::set_resolution(int Nq_in, int Nq_out, int Nq_cha) { set_resolution_rec(*root, Nq_in, Nq_out, Nq_cha); }
void LUT_Tree::set_leaves(const vec &p_Msg, const vec &p_Cha) { set_leaves_rec(*root, p_Msg, p_Cha); }
void LUT_Tree::reset_pmfs() { if (root) reset_rec(*root); }
vec LUT_Tree::update(bool reuse) { return update_rec(*root, reuse, type == CHKTREE); }

std::deque<Node *> LUT_Tree::get_level_nodes(int level) {
    std::deque<Node *> out;
    if (!root) throw std::logic_error("LUT_Tree::get_level_nodes(): tree empty");
    root->get_level_nodes(level, 0, out);
    return out;
}

// ------------------------------------------------------------------ text format
// trees/README.md:87-95: per node "num_children", "type inres outres", then the map if inres > 0
namespace {
void serialize_rec(const Node &n, std::ostream &os) {
    os << n.children.size() << '\n' << static_cast<int>(n.type) << ' ' << n.Q.size() << ' ' << n.K << '\n';
    if (!n.Q.empty()) {
        for (size_t i = 0; i + 1 < n.Q.size(); i++) os << n.Q[i] << ' ';
        os << n.Q.back() << '\n';
    }
    for (auto &c : n.children) serialize_rec(*c, os);
}

NodePtr deserialize_rec(std::istream &is) {
    long nch, t, inres, outres;
    if (!(is >> nch >> t >> inres >> outres)) throw std::runtime_error("LUT_Tree: truncated node record");
    if (t < 0 || t >= Node::num_node_types || inres < 0 || outres < 0 || nch < 0)
        throw std::runtime_error("LUT_Tree_Node: wrong node data");
    NodePtr n(new Node(static_cast<Node::node_type_t>(t)));
    n->K = (int)outres;
    n->Q.resize((size_t)inres);
    for (long i = 0; i < inres; i++) {
        if (!(is >> n->Q[(size_t)i]) || n->Q[(size_t)i] < 0 || n->Q[(size_t)i] >= n->K)
            throw std::runtime_error("LUT_Tree_Node: wrong mapping data");
    }
    for (long i = 0; i < nch; i++) n->children.push_back(deserialize_rec(is));
    return n;
}
}  // namespace

std::ostream &operator<<(std::ostream &os, const LUT_Tree &t) {
    os << static_cast<int>(t.type) << ' ' << t.num_leaves << '\n';
    if (t.root) serialize_rec(*t.root, os);
    return os;
}

std::istream &operator>>(std::istream &is, LUT_Tree &t) {
    long tt, nl;
    if (!(is >> tt >> nl) || tt < 0 || tt >= LUT_Tree::num_tree_types || nl < 0) throw std::runtime_error("LUT_Tree: wrong tree data");
    t = LUT_Tree();
    t.type = static_cast<LUT_Tree::tree_type_t>(tt);
    t.num_leaves = (int)nl;
    t.root = deserialize_rec(is);
    return is;
}

std::ostream &operator<<(std::ostream &os, const LUT_Tree_Array &a) {
    os << a.size() << '\n';
    for (auto &row : a) {
        os << row.size() << '\n';
        for (auto &t : row) os << t;
    }
    return os;
}

std::istream &operator>>(std::istream &is, LUT_Tree_Array &a) {
    long ns;
    if (!(is >> ns) || ns < 0) throw std::runtime_error("Array<Array<LUT_Tree>>: wrong tree data");
    a.assign((size_t)ns, {});
    for (auto &row : a) {
        long nc;
        if (!(is >> nc) || nc < 1) throw std::runtime_error("Array<Array<LUT_Tree>>: wrong tree data");
        row.resize((size_t)nc);
        for (auto &t : row) is >> t;
    }
    return is;
}

std::string to_string(const LUT_Tree_Array &a) {
    std::ostringstream os;
    os << a;
    return os.str();
}

}  // namespace lut_ldpc
