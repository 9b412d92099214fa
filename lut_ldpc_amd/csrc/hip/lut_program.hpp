// lut_program.hpp -- host side of the node-update evaluator.
//
// The reference evaluates a LUT_Tree by walking a pointer tree once per output message with
// a freshly copied std::deque (src/LUT_Tree.cpp:774-820, 402-445).  Here every
// (tree set, degree class) is compiled ONCE into a flat, branch-free "node program":
//   * values   : the node's inputs (its d messages, then the channel label) followed by one
//                value per distinct (LUT node, input tuple) pair -- identical sub-expressions
//                of the d output walks are shared, which turns d*(d-1) look-ups into
//                sum over nodes of (leaves+1) for the usual trees;
//   * look-ups : full tables (the symmetric half stored by the reference is expanded), so the
//                device does one indexed byte load per look-up and no sign handling;
//   * slots    : values are assigned to a small number of reusable slots by liveness.
// The device kernels either interpret this program (generic path) or, for the balanced binary
// trees ber_sim always designs, use a compile-time specialisation of the same value numbering.
#pragma once
#include <array>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace lutldpc {

enum NodeType { NT_IM = 0, NT_ROOT = 1, NT_MSG = 2, NT_CHA = 3 };   // src/LUT_Tree.hpp:188-192
enum TreeType { TT_VAR = 0, TT_CHK = 1, TT_DEC = 2 };              // src/LUT_Tree.hpp:50-53

struct TreeNode {
    int type = 0;
    int K = 0;                     // output alphabet
    std::vector<int> Q;            // half table as stored by the reference
    // composed nodes (compose_tree below) carry their FULL table in the look-up's own indexing: VAR / DEC: index = sum of child
    // label * place value; CHK: [odd-parity half | even-parity half], each indexed by the children's magnitudes
    std::vector<uint8_t> full;
    std::vector<std::unique_ptr<TreeNode>> child;
    bool is_leaf() const { return type == NT_MSG || type == NT_CHA; }
};

struct Tree {
    int type = 0;
    int num_leaves = 0;
    std::unique_ptr<TreeNode> root;
};

using TreeArray = std::vector<std::vector<Tree>>;   // [set][class]

// ---- parser for the reference's text serialisation (src/LUT_Tree.cpp:847-927) -------------
class IntReader {
public:
    explicit IntReader(const char *s) : p_(s ? s : "") {}
    bool next(long &v) {
        char *e;
        v = std::strtol(p_, &e, 10);
        if (e == p_) return false;
        p_ = e;
        return true;
    }
private:
    const char *p_;
};

inline bool parse_node(IntReader &r, std::unique_ptr<TreeNode> &out, int depth, std::string &err) {
    long nch, t, inres, outres;
    if (depth > 64) { err = "tree deeper than 64 levels"; return false; }
    if (!r.next(nch) || !r.next(t) || !r.next(inres) || !r.next(outres)) { err = "truncated node record"; return false; }
    if (t < 0 || t > 3 || inres < 0 || outres < 0 || nch < 0 || nch > 64 || inres > (1L << 26)) { err = "node record out of range"; return false; }
    out.reset(new TreeNode);
    out->type = (int)t; out->K = (int)outres;
    out->Q.resize((size_t)inres);
    for (long i = 0; i < inres; i++) {
        long q;
        if (!r.next(q)) { err = "truncated LUT"; return false; }
        if (q < 0 || q >= outres) { err = "LUT entry outside the node's output alphabet"; return false; }
        out->Q[(size_t)i] = (int)q;
    }
    for (long i = 0; i < nch; i++) {
        std::unique_ptr<TreeNode> c;
        if (!parse_node(r, c, depth + 1, err)) return false;
        out->child.push_back(std::move(c));
    }
    return true;
}

inline bool parse_tree_array(const char *txt, TreeArray &out, std::string &err) {
    IntReader r(txt);
    long ns;
    out.clear();
    if (!r.next(ns) || ns < 0 || ns > 100000) { err = "bad tree-set count"; return false; }
    out.resize((size_t)ns);
    for (long s = 0; s < ns; s++) {
        long nc;
        if (!r.next(nc) || nc < 1 || nc > 1000) { err = "bad degree-class count"; return false; }
        out[(size_t)s].resize((size_t)nc);
        for (long c = 0; c < nc; c++) {
            long tt, nl;
            if (!r.next(tt) || !r.next(nl) || tt < 0 || tt > 2 || nl < 0) { err = "bad tree header"; return false; }
            Tree &t = out[(size_t)s][(size_t)c];
            t.type = (int)tt; t.num_leaves = (int)nl;
            if (!parse_node(r, t.root, 0, err)) return false;
        }
    }
    return true;
}

// ---- compiled node program -----------------------------------------------------------------
constexpr int kMaxChildren = 8;
constexpr int kMaxChildrenCompose = 4;      // fan-in of a composed node (compose_tree)

struct Op {                 // one LUT look-up
    uint32_t tab_off;       // byte offset of its full table inside the class blob
    uint32_t tab_len;       // entries of the full table
    uint16_t dst;           // slot written
    int16_t  out_idx;       // >= 0: the result is output message #out_idx of the node
    uint8_t  nchild;
    uint8_t  kind;          // 0: VAR/DEC label = sum child*mult ; 1: CHK sign-magnitude label
    uint16_t child[kMaxChildren];   // slots read
    uint32_t mult[kMaxChildren];    // place value of each child in the label
    uint16_t childK[kMaxChildren];  // alphabet of each child (CHK: sign threshold K/2)
    uint32_t half_len;      // CHK: offset of the parity-even half inside the table
};

struct Program {
    int kind = 0;            // TreeType
    int n_in = 0;            // inputs (slots 0..n_in-1 on entry)
    int n_out = 0;
    int n_slots = 0;
    int n_ops_naive = 0;     // look-ups of the reference's per-output walk
    std::vector<Op> ops;
    std::vector<uint8_t> tables;   // concatenated full tables
    // LUT node -> {offset, length, parity-even offset} of its full table inside `tables`
    std::vector<std::pair<const TreeNode *, std::array<uint32_t, 3>>> node_tabs;
    // for inputs that are forwarded unchanged to an output (cannot happen with LUT roots, kept
    // for completeness): out_from_input[i] = input slot or -1
};

namespace detail {

struct ValKey {
    const TreeNode *node;
    std::vector<int> args;
    bool operator<(const ValKey &o) const { return node != o.node ? node < o.node : args < o.args; }
};

struct Builder {
    int kind;
    std::map<ValKey, int> memo;              // -> value id
    std::map<const TreeNode *, uint32_t> tab_of;
    struct Val { const TreeNode *node; std::vector<int> args; int out_idx; };
    std::vector<Val> vals;                    // value id = n_in + index
    int n_in = 0;
    int naive = 0;
    std::string err;

    // walk like LUT_Tree_Node::{var,chk}_msg_update: leaves consume the queue front
    int walk(const TreeNode *n, const std::vector<int> &queue, size_t &pos) {
        bool leaf = (kind == TT_CHK) ? (n->type == NT_MSG) : n->is_leaf();
        if (leaf) {
            if (pos >= queue.size()) { err = "tree has more leaves than the node has inputs"; return -1; }
            return queue[pos++];
        }
        if (n->child.empty()) { err = "internal node without children"; return -1; }
        ValKey k; k.node = n;
        for (auto &c : n->child) {
            int v = walk(c.get(), queue, pos);
            if (v < 0) return -1;
            k.args.push_back(v);
        }
        naive++;
        auto it = memo.find(k);
        if (it != memo.end()) return it->second;
        int id = n_in + (int)vals.size();
        vals.push_back({n, k.args, -1});
        memo.emplace(std::move(k), id);
        return id;
    }
};

inline bool expand_table(const TreeNode *n, int kind, std::vector<uint8_t> &blob, uint32_t &off, uint32_t &len,
                         uint32_t &half, std::string &err) {
    // label space of the node
    uint64_t space = 1;
    for (auto &c : n->child) {
        int K = c->K;
        if (K < 2 || (K & 1)) { err = "child alphabet must be even and >= 2"; return false; }
        space *= (uint64_t)(kind == TT_CHK ? K / 2 : K);
        if (space > (1ull << 26)) { err = "node table beyond 2^26 labels"; return false; }
    }
    if (n->K < 2 || n->K > 256) { err = "node output alphabet must be in [2,256]"; return false; }
    size_t L = n->Q.size();
    off = (uint32_t)blob.size();
    if (!n->full.empty()) {        // composed node: the table is there already
        const uint64_t want = kind == TT_CHK ? 2 * space : space;
        if (n->full.size() != want) { err = "composed table length does not match its inputs"; return false; }
        len = (uint32_t)want; half = kind == TT_CHK ? (uint32_t)space : 0u;
        blob.insert(blob.end(), n->full.begin(), n->full.end());
        while (blob.size() & 3) blob.push_back(0);
        return true;
    }
    if (kind == TT_CHK) {
        // src/LUT_Tree.cpp:441-444: odd sign parity -> Q[label], even -> K-1-Q[label]
        if (L != space) { err = "check LUT length does not match its inputs"; return false; }
        len = (uint32_t)(2 * L); half = (uint32_t)L;
        blob.resize(blob.size() + 2 * L);
        for (size_t i = 0; i < L; i++) { blob[off + i] = (uint8_t)n->Q[i]; blob[off + L + i] = (uint8_t)(n->K - 1 - n->Q[i]); }
    } else {
        // src/LUT_Tree.cpp:414-417: label < |Q| -> Q[label], else K-1-Q[2|Q|-1-label]
        if (2 * L != space) { err = "variable LUT length is not half of its label space"; return false; }
        len = (uint32_t)(2 * L); half = 0;
        blob.resize(blob.size() + 2 * L);
        for (size_t i = 0; i < L; i++) { blob[off + i] = (uint8_t)n->Q[i]; blob[off + 2 * L - 1 - i] = (uint8_t)(n->K - 1 - n->Q[i]); }
    }
    while (blob.size() & 3) blob.push_back(0);
    return true;
}

}  // namespace detail

// ---- exact table composition ---------------------------------------------------------------------------------------
// A LUT node X whose value feeds ONE input of its parent P can be folded into P: P'(..., children of X, ...) with the table
// T'[...] = T_P[..., T_X[...], ...] gives the same output for every input tuple (the maps Q are plain functions of the labels,
// src/LUT_Tree.cpp:402-418 / 420-445), so the decoder stays bit-exact while X's look-ups disappear -- for the balanced trees of
// a degree-8 node 34 look-ups per frame become 20, for degree 3 six become three.  The leaves of X take X's place in P's child
// list, so the depth-first leaf order (= the order in which the queue is consumed) does not change.
// CHK nodes work on (sign, magnitude): the composed output depends on the signs only through their total parity (the sign of
// X's output is parity_X xor a function of the magnitudes), so P' is again a [odd | even] pair of magnitude-indexed tables.
// Greedy, bottom-up: a node absorbs internal children while the label space stays <= max_space and the fan-in <= 8.
namespace detail {

inline uint64_t node_space(const TreeNode *n, int kind) {
    uint64_t sp = 1;
    for (auto &c : n->child) sp *= (uint64_t)(kind == TT_CHK ? c->K / 2 : c->K);
    return sp;
}
inline bool node_is_leaf(const TreeNode *n, int kind) { return kind == TT_CHK ? n->type == NT_MSG : n->is_leaf(); }

// full table of a node in the look-up's indexing (see TreeNode::full)
inline bool node_full_table(const TreeNode *n, int kind, std::vector<uint8_t> &t) {
    if (!n->full.empty()) { t = n->full; return true; }
    std::vector<uint8_t> blob;
    uint32_t off, len, half;
    std::string err;
    if (!expand_table(n, kind, blob, off, len, half, err)) return false;
    t.assign(blob.begin(), blob.begin() + len);
    return true;
}

// fold child number `ci` of P into P
inline bool absorb_child(TreeNode *P, size_t ci, int kind) {
    TreeNode *X = P->child[ci].get();
    std::vector<uint8_t> TP, TX;
    if (!node_full_table(P, kind, TP) || !node_full_table(X, kind, TX)) return false;
    const size_t nP = P->child.size(), nX = X->child.size(), nN = nP - 1 + nX;
    auto alph = [&](const TreeNode *c) { return (uint64_t)(kind == TT_CHK ? c->K / 2 : c->K); };
    std::vector<uint64_t> aP, aX, aN;
    for (auto &c : P->child) aP.push_back(alph(c.get()));
    for (auto &c : X->child) aX.push_back(alph(c.get()));
    for (size_t i = 0; i < nP; i++) { if (i == ci) aN.insert(aN.end(), aX.begin(), aX.end()); else aN.push_back(aP[i]); }
    uint64_t spN = 1, spP = 1, spX = 1;
    for (auto a : aN) spN *= a;
    for (auto a : aP) spP *= a;
    for (auto a : aX) spX *= a;
    const int hX = X->K / 2;
    std::vector<uint8_t> TN((size_t)(kind == TT_CHK ? 2 * spN : spN));
    std::vector<uint64_t> dig(nN);
    for (uint64_t idx = 0; idx < spN; idx++) {
        uint64_t r = idx;
        for (size_t i = 0; i < nN; i++) { dig[i] = r % aN[i]; r /= aN[i]; }           // child 0 least significant
        uint64_t labX = 0, mul = 1;
        for (size_t i = 0; i < nX; i++) { labX += mul * dig[ci + i]; mul *= aX[i]; }
        if (kind != TT_CHK) {
            const uint64_t xo = TX[(size_t)labX];
            uint64_t labP = 0; mul = 1;
            for (size_t i = 0, j = 0; i < nP; i++) { const uint64_t v = (i == ci) ? xo : dig[j]; labP += mul * v; mul *= aP[i]; j += (i == ci) ? nX : 1; }
            TN[(size_t)idx] = TP[(size_t)labP];
        } else {
            for (int par_all = 0; par_all < 2; par_all++) {          // 1 = odd number of negative leaves; put that sign on X's first leaf
                const uint64_t xo = TX[(size_t)(par_all ? labX : spX + labX)];
                const bool negx = (int)xo < hX;
                const uint64_t magx = negx ? (uint64_t)(hX - 1) - xo : xo - (uint64_t)hX;
                uint64_t labP = 0; mul = 1;
                for (size_t i = 0, j = 0; i < nP; i++) { const uint64_t v = (i == ci) ? magx : dig[j]; labP += mul * v; mul *= aP[i]; j += (i == ci) ? nX : 1; }
                const uint8_t out = TP[(size_t)(negx ? labP : spP + labP)];    // P's parity = sign of X's output (all other inputs positive)
                TN[(size_t)(par_all ? idx : spN + idx)] = out;
            }
        }
    }
    std::vector<std::unique_ptr<TreeNode>> nc;
    for (size_t i = 0; i < nP; i++) {
        if (i == ci) for (auto &xc : X->child) nc.push_back(std::move(xc));
        else nc.push_back(std::move(P->child[i]));
    }
    P->child = std::move(nc);
    P->full = std::move(TN);
    P->Q.clear();
    return true;
}

inline void compose_node(TreeNode *P, int kind, uint64_t max_space) {
    if (node_is_leaf(P, kind)) return;
    for (auto &c : P->child) compose_node(c.get(), kind, max_space);
    for (;;) {
        // the internal child whose absorption gives the smallest label space
        size_t best = (size_t)-1; uint64_t best_sp = 0;
        const uint64_t sp = node_space(P, kind);
        for (size_t i = 0; i < P->child.size(); i++) {
            const TreeNode *X = P->child[i].get();
            if (node_is_leaf(X, kind) || X->child.empty()) continue;
            const uint64_t a = (uint64_t)(kind == TT_CHK ? X->K / 2 : X->K);
            const uint64_t nsp = sp / a * node_space(X, kind);
            if (nsp > max_space || P->child.size() - 1 + X->child.size() > (size_t)kMaxChildrenCompose) continue;
            if (best == (size_t)-1 || nsp < best_sp) { best = i; best_sp = nsp; }
        }
        if (best == (size_t)-1 || !absorb_child(P, best, kind)) return;
    }
}

inline std::unique_ptr<TreeNode> clone_node(const TreeNode *n) {
    std::unique_ptr<TreeNode> c(new TreeNode);
    c->type = n->type; c->K = n->K; c->Q = n->Q; c->full = n->full;
    for (auto &ch : n->child) c->child.push_back(clone_node(ch.get()));
    return c;
}

}  // namespace detail

// a composed copy of the tree (max_space: largest label space of a composed node; the tables are bytes: <= 4096 keeps a class
// of degree 8 within 13 KB of LDS)
inline Tree compose_tree(const Tree &t, int kind, uint64_t max_space) {
    Tree c;
    c.type = t.type; c.num_leaves = t.num_leaves;
    if (t.root) { c.root = detail::clone_node(t.root.get()); detail::compose_node(c.root.get(), kind, max_space); }
    return c;
}

// Compile the tree for a node of degree d.  VAR: inputs = d messages + channel label, d
// outputs (output i walks the queue with element i removed, src/LUT_Tree.cpp:783-788);
// CHK: inputs = d messages, d outputs (:800-805); DEC: inputs = d messages + channel, one
// output over the whole queue (:815-816).
inline bool compile_program(const Tree &t, int kind, int d, Program &P, std::string &err) {
    detail::Builder b;
    b.kind = kind;
    P = Program();
    P.kind = kind;
    P.n_in = (kind == TT_CHK) ? d : d + 1;
    P.n_out = (kind == TT_DEC) ? 1 : d;
    b.n_in = P.n_in;
    if (!t.root) { err = "empty tree"; return false; }
    std::vector<int> roots;
    for (int i = 0; i < P.n_out; i++) {
        std::vector<int> q;
        for (int j = 0; j < P.n_in; j++) if (kind == TT_DEC || j != i) q.push_back(j);
        size_t pos = 0;
        int v = b.walk(t.root.get(), q, pos);
        if (v < 0) { err = b.err; return false; }
        if (pos != q.size()) { err = "tree has fewer leaves than the node has inputs"; return false; }
        if (v < P.n_in) { err = "tree root is a leaf"; return false; }
        if (b.vals[(size_t)(v - P.n_in)].out_idx >= 0) { err = "two outputs share one value"; return false; }
        b.vals[(size_t)(v - P.n_in)].out_idx = i;
        roots.push_back(v);
    }
    P.n_ops_naive = b.naive;
    // tables (one per distinct tree node)
    std::map<const TreeNode *, std::array<uint32_t, 3>> tabs;
    for (auto &v : b.vals) {
        if (tabs.count(v.node)) continue;
        uint32_t off, len, half;
        if ((int)v.node->child.size() > kMaxChildren) { err = "node fan-in above 8"; return false; }
        if (!detail::expand_table(v.node, kind, P.tables, off, len, half, err)) return false;
        tabs[v.node] = {off, len, half};
        P.node_tabs.push_back({v.node, {off, len, half}});
    }
    // liveness: last op index reading each value
    size_t nv = (size_t)P.n_in + b.vals.size();
    std::vector<int> last_use(nv, -1);
    for (size_t k = 0; k < b.vals.size(); k++) for (int a : b.vals[k].args) last_use[(size_t)a] = (int)k;
    // slot assignment: inputs keep slots 0..n_in-1 (the device loads them there and, for the
    // in-place update, blends stores with the original input rows, so input slots are never
    // recycled); computed values take the lowest free slot
    std::vector<int> slot_of(nv, -1);
    std::vector<int> free_slots;
    int n_slots = P.n_in;
    for (int i = 0; i < P.n_in; i++) slot_of[(size_t)i] = i;
    for (size_t k = 0; k < b.vals.size(); k++) {
        auto &v = b.vals[k];
        Op op{};
        auto tb = tabs[v.node];
        op.tab_off = tb[0]; op.tab_len = tb[1]; op.half_len = tb[2];
        op.nchild = (uint8_t)v.args.size();
        op.kind = (kind == TT_CHK) ? 1 : 0;
        op.out_idx = (int16_t)v.out_idx;
        uint32_t base = 1;
        for (size_t c = 0; c < v.args.size(); c++) {
            op.child[c] = (uint16_t)slot_of[(size_t)v.args[c]];
            op.mult[c] = base;
            int K = v.node->child[c]->K;
            op.childK[c] = (uint16_t)K;
            base *= (uint32_t)(kind == TT_CHK ? K / 2 : K);
        }
        // operands die here?  (a slot may be reused as this op's destination only if no
        // LATER operand position reads it -- all operands are read before the write, so
        // reusing a dying operand's slot is safe)
        for (int a : v.args)
            if (a >= P.n_in && last_use[(size_t)a] == (int)k && slot_of[(size_t)a] >= 0) {
                bool dup = false;
                for (int s : free_slots) if (s == slot_of[(size_t)a]) dup = true;
                if (!dup) free_slots.push_back(slot_of[(size_t)a]);
            }
        int s;
        if (v.out_idx >= 0 && last_use[(size_t)P.n_in + k] < 0) {
            // pure outputs are stored straight from the look-up; they still get a slot so the
            // interpreter has one code path
        }
        if (!free_slots.empty()) {
            size_t best = 0;
            for (size_t i = 1; i < free_slots.size(); i++) if (free_slots[i] < free_slots[best]) best = i;
            s = free_slots[best];
            free_slots.erase(free_slots.begin() + (long)best);
        } else s = n_slots++;
        slot_of[(size_t)P.n_in + k] = s;
        op.dst = (uint16_t)s;
        // an output that nobody reads frees its slot right away
        if (last_use[(size_t)P.n_in + k] < 0) free_slots.push_back(s);
        P.ops.push_back(op);
    }
    P.n_slots = n_slots;
    if (n_slots > 4096) { err = "program needs more than 4096 slots"; return false; }
    return true;
}

// Host evaluation of a program for one node (self test of the compile step).
inline bool eval_program(const Program &P, const int32_t *in, int32_t *out) {
    std::vector<int32_t> slot((size_t)P.n_slots, 0);
    for (int i = 0; i < P.n_in; i++) slot[(size_t)i] = in[i];
    for (auto &op : P.ops) {
        uint32_t label = 0, parity = 0;
        for (int c = 0; c < op.nchild; c++) {
            int32_t x = slot[op.child[c]];
            if (x < 0 || x >= op.childK[c]) return false;
            if (op.kind == 1) {
                int h = op.childK[c] / 2;
                if (x < h) { parity ^= 1; label += op.mult[c] * (uint32_t)(h - 1 - x); }
                else label += op.mult[c] * (uint32_t)(x - h);
            } else label += op.mult[c] * (uint32_t)x;
        }
        uint32_t idx = (op.kind == 1) ? (parity ? label : op.half_len + label) : label;
        if (idx >= op.tab_len) return false;
        int32_t r = P.tables[op.tab_off + idx];
        slot[op.dst] = r;
        if (op.out_idx >= 0) out[op.out_idx] = r;
    }
    return true;
}

// ---- check-node programs over FULL labels ---------------------------------------------------------------------------
// The reference's check look-up (src/LUT_Tree.cpp:420-445) works on (sign, magnitude): label = sum of the children's magnitudes
// times their place values, the parity of the children's signs picks Q[label] or K-1-Q[label].  Evaluated that way a look-up
// costs nine vector instructions (two splits, label, parity, half select).  The same function of the children's LABELS is one
// table of prod(K_c) entries -- 64 bytes for two 3-bit children, 256 for two 4-bit ones -- and then a look-up is what a
// variable-node look-up is: one v_lshl_or_b32 and one ds_read_u8.  Bit-exact by construction (the table is the map itself).
// out = a copy of `prog` with kind-2 ops (label = sum child * mult over full alphabets) and its own table blob.
inline bool chk_full_label_program(const Program &prog, Program &out, uint64_t max_entries = 4096) {
    if (prog.kind != TT_CHK) return false;
    out = prog;
    out.tables.clear();
    out.node_tabs.clear();
    std::map<std::pair<uint32_t, std::vector<uint16_t>>, uint32_t> done;       // (old table, child alphabets) -> new offset
    for (auto &op : out.ops) {
        if (op.kind != 1 || op.nchild < 1) return false;
        uint64_t space = 1;
        std::vector<uint16_t> Ks;
        for (int c = 0; c < op.nchild; c++) { const int K = op.childK[c]; if (K < 2 || (K & 1)) return false; space *= (uint64_t)K; Ks.push_back((uint16_t)K); if (space > max_entries) return false; }
        const auto key = std::make_pair(op.tab_off, Ks);
        auto it = done.find(key);
        if (it == done.end()) {
            const uint32_t noff = (uint32_t)out.tables.size();
            out.tables.resize(noff + (size_t)space);
            std::vector<int> lab((size_t)op.nchild, 0);
            for (uint64_t i = 0; i < space; i++) {
                uint64_t r = i, idx = 0; int par = 0;
                for (int c = 0; c < op.nchild; c++) {
                    const int K = op.childK[c], hh = K / 2, l = (int)(r % (uint64_t)K);
                    r /= (uint64_t)K;
                    const int n = l < hh ? 1 : 0, m = n ? hh - 1 - l : l - hh;
                    idx += (uint64_t)m * op.mult[c]; par ^= n;
                }
                const uint64_t src = (uint64_t)op.tab_off + (par ? 0u : op.half_len) + idx;      // odd sign parity -> first half
                if (src >= prog.tables.size()) return false;
                out.tables[noff + (size_t)i] = prog.tables[(size_t)src];
            }
            while (out.tables.size() & 3) out.tables.push_back(0);
            it = done.emplace(key, noff).first;
        }
        uint32_t base = 1;
        for (int c = 0; c < op.nchild; c++) { op.mult[c] = base; base *= (uint32_t)op.childK[c]; }
        op.kind = 2; op.tab_off = it->second; op.tab_len = (uint32_t)space; op.half_len = 0;
    }
    return true;
}

}  // namespace lutldpc
