// decoder.hip -- host logic and C-ABI of the MI355X LUT-LDPC decode path (include/lut_ldpc_hip.h).
//
// Replaces LDPC_Code_LUT::lut_decode and everything below it (src/LDPC_Code_LUT.cpp:259-469,
// src/LUT_Tree.cpp:402-445,774-820) for a BATCH of frames: the frame loop of
// LDPC_BER_Sim::sim_snr_point (src/LDPC_BER_Sim.cpp:260-291) becomes the innermost, coalesced
// memory dimension.  See kernels_common.hpp for the HBM layout.
#include "../../../include/lut_ldpc_hip.h"
#include "kernels_common.hpp"
#include "kernels_generic.hpp"
#include "kernels_fast.hpp"
#include "kernels_frontend.hpp"
#include "kernels_compact.hpp"
#include "lut_program.hpp"
#include "jit.hpp"
#include "jit_resident.hpp"

namespace lutldpc { LUTLDPC_FAST_LAUNCHERS(extern) }     // instantiated in fast_*.hip / fused.hip

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

using namespace lutldpc;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(LUTLDPC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        if (count <= n) return hipSuccess;
        release();
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e == hipSuccess) n = count; else p = nullptr;
        return e;
    }
    hipError_t upload(const std::vector<T> &h) {
        hipError_t e = alloc(h.size() ? h.size() : 1);
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    size_t bytes() const { return n * sizeof(T); }
};

struct NodeClass {
    int deg = 0;
    std::vector<int> nodes;     // node ids, ascending
    int tree_class = -1;        // index of the matching tree inside a tree set
};

struct PassPlan {               // one launch: all degree classes of one pass of one tree set
    PassParams P{};
    int lds_bytes = 0;
    int out_slots = 0;
    bool lds_tab = true;
    bool valid = false;
};

}  // namespace

struct lutldpc_decoder {
    // ---- code
    int nvar = 0, nchk = 0, E = 0;
    std::vector<int> dv, dc, cn_msg_idx, vn_ptr, cn_ptr, cn_vn;
    std::vector<NodeClass> vclass, cclass;
    std::vector<int> vn_list, cn_list;         // nodes sorted by class
    // ---- decoder parameters
    int Nq_Cha = 0, max_iters_created = 0, max_iters = 0, psc = 1, pisc = 0, min_lut = 1;
    std::vector<int> Nq_Msg, iter_set;         // iter_set = cumsum(reuse == 0) - 1
    TreeArray var_trees, chk_trees;
    // ---- programs: [set][class]
    std::vector<std::vector<Program>> var_prog, chk_prog, dec_prog;
    // check programs over full labels (lut_program.hpp: chk_full_label_program) for the generated check kernels, and where their
    // tables sit in the blob: [set][class], {offset, bytes}, bytes = 0: none (the generated kernel then works on sign / magnitude)
    std::vector<std::vector<Program>> chk_prog_full;
    std::vector<std::vector<std::pair<int, int>>> chk_full_tab;
    std::vector<std::vector<Program>> chk_prog_cf;                    // the same for the programs the LDS-resident decoder runs (chk_prog_c)
    std::vector<std::vector<std::pair<int, int>>> chk_tab_cf;
    int chk_full_labels = 1;    // LUTLDPC_CHK_FULL=0: generated check kernels on (sign, magnitude) tables as the reference walks them
    // the same trees after exact table composition (lut_program.hpp: compose_tree): fewer, larger look-ups; used by the generated
    // LDS-resident kernel.  *_tab_c: {offset, bytes} of the class blob inside all_tables.  LUTLDPC_COMPOSE=0: off (the originals).
    std::vector<std::vector<Program>> var_prog_c, chk_prog_c, dec_prog_c;
    std::vector<std::vector<std::pair<int, int>>> var_tab_c, chk_tab_c, dec_tab_c;
    // Measured on MI355X (tools/resident_probe.py): a 4 KB table spreads its 1024 dwords over 32 banks 32 deep -- the three-input
    // look-ups run into 3-4-way bank conflicts where a 256-byte table has at most two dwords per bank -- and the halved look-up
    // count does not pay for it: (3,6) N=10000 1.72 -> 1.24 M codewords/s with composition.  Off by default; LUTLDPC_COMPOSE=1.
    int use_compose = 0, compose_space = 4096;
    std::vector<Op> all_ops;
    std::vector<uint8_t> all_tables;
    std::vector<PassPlan> var_plan, chk_plan, dec_plan;   // per tree set
    PassPlan cn_minsum_plan;
    std::vector<std::vector<FastClassPlan>> var_fast, dec_fast;   // [set][class]
    // dense per-class index tables of the specialised kernels: variable classes {node id, first edge}
    // per node, check classes the DEG edge ids per node (no pointer chasing, scalar loads)
    std::vector<int32_t> fast_idx;
    std::vector<int> vn_idx_off, cn_idx_off;                      // per class
    std::vector<int> cn_nidx_off;                                 // per check class: the NODE of every entry of the edge table (iteration 0 reads the initial-message rows)
    std::vector<int> cn_tidx_off, cn_tnidx_off, vn_tidx_off;      // transposed tables of the LDS-resident decoder: [k][node] edges / nodes per check class, [2][node] {node id, first edge} per variable class
    int first_from_nodes = 1;                                     // LUTLDPC_FIRST_FROM_NODES=0: copy the initial messages to the edge rows first (init_edges_kernel)
    // chain fusion (build_fast_index): per check class the offset of its {back, forward} node table (-1 = no links),
    // per variable class the dense table / count of the nodes NOT updated inside the check pass
    std::vector<int> chain_idx_off, vn_red_off, vn_red_n;
    int chain_vclass = -1, n_chain_nodes = 0, use_chain = 1;
    std::vector<int> cn_npw_class;          // checks per wave of each check class (chain-rich classes of wide checks get at least 4)
    // ---- device
    int device = -1;
    hipStream_t stream = nullptr;
    DevBuf<int32_t> d_vn_ptr, d_cn_ptr, d_cn_idx, d_cn_vn, d_vn_list, d_cn_list, d_fast_idx;
    DevBuf<Op> d_ops;
    DevBuf<uint8_t> d_tables;
    // batch buffers
    int Bcap = 0;
    DevBuf<uint8_t> d_msgs, d_cha_t, d_msg0_t, d_hard, d_state, d_vfail;
    DevBuf<int32_t> d_iters;
    DevBuf<uint8_t> d_in_cha, d_in_msg, d_out_bits;   // frame-major staging for the host entry points
    DevBuf<int32_t> d_out_iters;
    DevBuf<double> d_llr, d_qb_cha, d_qb_msg;
    DevBuf<int32_t> d_map;
    DevBuf<uint8_t> d_codewords;
    DevBuf<int32_t> d_stats;
    // ---- tuning
    int nodes_per_block = 16;
    // specialised kernels: nodes handled by one wave = edges_per_wave / degree (equal work per wave for
    // every degree class); a fixed count when LUTLDPC_NODES_PER_WAVE[_CN] is set.  Measured on MI355X
    // (DVB-S2, 4096 frames, repeated runs): short waves win -- 2 degree-8 nodes / 6 degree-7 checks per wave (longer check runs also keep more chain nodes inside a wave).
    int nodes_per_wave = 0, nodes_per_wave_cn = 0;        // 0 = derive from the degree
    int vn_edges_per_wave = 16, cn_edges_per_wave = 42;
    bool cn_edges_from_env = false;
    int fused_prio = 0;
    // compaction of the surviving frames (kernels_compact.hpp): as-shipped mode, skewed pipeline
    // (off by default: measured on MI355X it does not pay -- DVB-S2 frames finish too late (41.7 of 50 iterations on
    // average), (3,6) frames finish so close together that whole groups fall idle by themselves; LUTLDPC_COMPACT=1)
    int use_compact = -1, compact_first = 8, compact_every = 0;     // use: -1 = automatic (long iterations only), every: 0 = automatic
    float compact_margin = 1.0f;                                    // LUTLDPC_COMPACT_MARGIN (0: permute whenever a group falls idle)
    float compact_min_share = 0.35f;                                // ... and at least this share of the live groups falls idle at once
    int compact_keep = 1;                                           // LUTLDPC_COMPACT_KEEP: the frames that left keep their rows, bits recovered once at the end
    DevBuf<int32_t> d_frame_of, d_perm, d_tmp3, d_ctl, d_slot_of, d_iters_tmp;
    DevBuf<int32_t> d_grp;                           // per frame group: every frame failed the probe of the test on the channel decisions
    // Parameter structures of the per-class / generic / sampler kernels live in DEVICE memory (kernel arguments stay <= 128 bytes,
    // kernels_common.hpp: launch_k): a small arena keyed by content.  The first decode of a shape runs as plain launches and
    // uploads what it needs; the captured second run finds every structure there already.
    struct ParamArena {
        std::map<std::string, size_t> off_of;        // content -> byte offset (the key's bytes are the host copy the upload reads)
        std::vector<std::unique_ptr<DevBuf<uint8_t>>> chunks;
        std::vector<size_t> chunk_base;
        size_t used = 0, cap = 0;
        void release() { for (auto &c : chunks) c->release(); chunks.clear(); chunk_base.clear(); off_of.clear(); used = cap = 0; }
    } params;
    // message dumps of output_verbosity >= 2 (src/LDPC_Code_LUT.cpp:292-298,311-317,331-337): a small-batch debug path -- per-class
    // streaming launches, the edge rows copied out after the edge initialisation, (level > 2) every check pass and every
    // variable pass.  host: [dump][B][E] bytes, dumps in the reference's print order.
    struct Trace { int level = 0; uint8_t *host = nullptr; size_t cap = 0; int n = 0; int B = 0; };
    Trace trace;
    DevBuf<uint8_t> d_trace;
    // LDS-resident decoder (jit_resident.hpp): codes whose edge messages fit the LDS of a compute unit are decoded by ONE generated
    // kernel per decode -- all iterations inside, no HBM traffic between the labels and the decided bits.  LUTLDPC_RESIDENT=0: off
    // (the streaming kernels run instead); LUTLDPC_RESIDENT_S / _NT force the sets per workgroup / threads per workgroup.
    // frame-major label / bit buffers of the current decode_device call, handed to the resident kernel (it reads and writes them
    // itself: no transposes); null = rows
    const uint8_t *fm_cha = nullptr, *fm_msg0 = nullptr;
    uint8_t *fm_bits = nullptr;
    int resident_fm = 1;           // LUTLDPC_RESIDENT_FM=0: always through the row layout
    int use_resident = 1, resident_force_S = 0, resident_force_NT = 0, resident_U = 0, resident_xcd = 1, resident_flag_reduce = -1, resident_waves_eu = 0, resident_cn_persistent = -1;
    bool resident_ok = false;
    struct ResidentPlan { int S = 0, NT = 0, lds = 0; const JitKernel *k = nullptr; };
    std::map<int, ResidentPlan> resident_plans;       // by frame groups
    std::string resident_log;
    int use_jit = 1;            // tree-specialised kernels for shapes the compile-time path does not cover (jit.hpp)
    // (the loaded kernels live in a process-wide registry keyed by device + source text, see jit_registry(): decoders share
    // them and they are never unloaded)
    std::vector<std::vector<const JitKernel *>> var_jit, dec_jit, chk_jit;     // [set][class], null = none
    std::string jit_log;                                               // last hiprtc diagnostic (describe())
    // LUTLDPC_VALIDATE=1 (debug): every role of a fused launch is checked against the allocation sizes before the launch and
    // the stream is synchronised after it, so that a device fault is attributed to ONE launch (no graph replay then)
    int validate = 0;
    // as-shipped mode: the decided bits of early-terminated frames are recovered once, at the end, from their frozen messages
    // (hard_from_frozen_kernel) instead of being stored by every variable pass.  Needs the min-sum check update and one
    // message alphabet; LUTLDPC_LATE_HARD=0 restores the stores.
    int late_hard = 1;
    DevBuf<uint8_t> d_chain_internal;      // 1 = variable node updated inside the check pass (build_fast_index)
    std::vector<uint8_t> chain_internal;
    DevBuf<int32_t> d_edge_vn;              // variable node of every edge (chain_hard_kernel)
    std::vector<int32_t> edge_vn;
    int sweep_reverse = 0;      // LUTLDPC_REVERSE: alternate the sweep direction over the frame groups between launches
    int use_graph = 1;          // replay repeated decodes as one hipGraph launch (decode_tiles)
    struct GraphSlot { int seen = 0; hipGraphExec_t exec = nullptr; };
    std::map<std::array<int, 4>, GraphSlot> graphs;       // key {B, psc, pisc, max_iters}
    void drop_graphs() { for (auto &kv : graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec); graphs.clear(); }
    double tail_front = 0.25;   // fused launches: fraction of the item list that the slowest role stays clear of at the end
    // (halving the per-wave work for short codes so that a pass has more waves was measured slower: -13 % on N=500)
    static constexpr int work_shift = 0;
    int npw_vn(int deg) const { return nodes_per_wave > 0 ? nodes_per_wave : std::max(1, (vn_edges_per_wave >> work_shift) / std::max(deg, 1)); }
    int npw_cn(int deg) const { return nodes_per_wave_cn > 0 ? nodes_per_wave_cn : std::max(1, (cn_edges_per_wave >> work_shift) / std::max(deg, 1)); }
    int npw_cn_class(size_t ci) const { return ci < cn_npw_class.size() && cn_npw_class[ci] > 0 ? cn_npw_class[ci] : npw_cn(cclass[ci].deg); }
    // Placement search (place_rows): where the row buffers of a large batch land in HBM decides 6 % of the decode rate (one
    // process, fresh allocations of the same sizes: 241.8 ... 262.9 k codewords/s on DVB-S2, each level steady to 0.1 %;
    // profiles/r03_level_probe_*.txt), and nothing visible from here predicts it -- so the first decode of a batch size tries
    // up to `place_candidates` allocations, times three iterations of the fused pipeline on each and keeps the fastest (it stops early
    // once a candidate stands clear of the slowest seen).  LUTLDPC_PLACE=0 off, =n at most n candidates.
    int place_candidates = 16;
    std::string place_info = "null";
    int use_fast = 1;
    int pack = 1;               // 2: nibble rows (all alphabets <= 16 labels), 1: byte rows
    int skew = 1;               // two-half skewed pipeline through pass_fused_kernel (one frame group: second half empty)
    bool skew_ok = false;       // every class of every set has a case in the fused kernel
    int fused_bucket_id = 0;    // degree bucket of the fused kernel (kernels_fast.hpp: kFusedVnDeg / kFusedCnDeg)
    // launch plan of the skewed pipeline for one (frame groups, psc, max_iters): the roles of every launch in DEVICE memory
    // (the kernel reads them through a pointer), the interleaved item tables, what follows each launch.  Built once, at
    // the first decode of that shape; dropped with the batch buffers (the roles hold strides of the flag buffers).
    struct SkewSlot { int n_roles = 0; size_t role_off = 0; const int32_t *items = nullptr; int nb = 0; int state_half = -1, state_ii = 0; };
    struct SkewPlan { std::vector<SkewSlot> slots; std::vector<RoleParams> h_roles; DevBuf<RoleParams> d_roles; };
    std::map<std::array<int, 3>, std::unique_ptr<SkewPlan>> skew_plans;
    // interleaved item tables, keyed by the role block counts AND the (quantised) share of the timeline each role keeps clear
    std::map<std::pair<std::vector<int>, std::vector<int>>, std::unique_ptr<DevBuf<int32_t>>> item_tabs;
    void drop_plans() {
        for (auto &kv : skew_plans) kv.second->d_roles.release();
        skew_plans.clear();
        for (auto &kv : item_tabs) kv.second->release();
        item_tabs.clear();
    }
    int tile() const { return kRowBytes * pack; }       // frames per group
    int bpad(int B) const { return (B + tile() - 1) / tile() * tile(); }
    // ---- profiling
    bool profiling = false;
    struct Ev { hipEvent_t a, b; int kind; };
    std::vector<Ev> ev_live;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[LUTLDPC_K_COUNT] = {0};
    int64_t prof_n[LUTLDPC_K_COUNT] = {0};
    std::string describe;
};

namespace {

// ----------------------------------------------------------------------------- profiling
hipEvent_t ev_get(lutldpc_decoder *d) {
    if (!d->ev_pool.empty()) { hipEvent_t e = d->ev_pool.back(); d->ev_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
void prof_fold(lutldpc_decoder *d) {
    if (d->ev_live.empty()) return;
    (void)hipStreamSynchronize(d->stream);
    for (auto &e : d->ev_live) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { d->prof_ms[e.kind] += ms; d->prof_n[e.kind]++; }
        d->ev_pool.push_back(e.a); d->ev_pool.push_back(e.b);
    }
    d->ev_live.clear();
}
struct Timed {
    lutldpc_decoder *d; int kind; hipEvent_t a{}, b{};
    Timed(lutldpc_decoder *d_, int k) : d(d_), kind(k) {
        if (d->profiling) { a = ev_get(d); b = ev_get(d); (void)hipEventRecord(a, d->stream); }
    }
    ~Timed() {
        if (d->profiling) { (void)hipEventRecord(b, d->stream); d->ev_live.push_back({a, b, kind}); }
    }
};

// ----------------------------------------------------------------------------- set-up helpers
void build_classes(const std::vector<int> &deg, std::vector<NodeClass> &cls, std::vector<int> &list) {
    std::map<int, std::vector<int>> by;
    for (size_t i = 0; i < deg.size(); i++) by[deg[i]].push_back((int)i);
    cls.clear(); list.clear();
    for (auto &kv : by) { NodeClass c; c.deg = kv.first; c.nodes = kv.second; cls.push_back(std::move(c)); }
    for (auto &c : cls) list.insert(list.end(), c.nodes.begin(), c.nodes.end());
}

// Build the launch plan of one pass from per-class programs (generic) -- progs may be empty
// for the min-sum pass.
int build_plan(lutldpc_decoder *d, const std::vector<NodeClass> &cls, const std::vector<Program> *progs,
               const std::vector<size_t> *op_off, const std::vector<size_t> *tab_off, PassPlan &plan) {
    if ((int)cls.size() > kMaxSeg) return fail(LUTLDPC_ERR_UNSUPPORTED, "more than 32 distinct node degrees in one pass");
    PassParams &P = plan.P;
    std::memset(&P, 0, sizeof(P));
    P.n_seg = (int)cls.size();
    P.nodes_per_block = d->nodes_per_block;
    P.E = d->E; P.N = d->nvar;
    int blk = 0, node_off = 0, max_slots = 0, max_tab = 0, max_out = 1;
    for (size_t i = 0; i < cls.size(); i++) {
        PassSeg &S = P.seg[i];
        S.block_begin = blk;
        S.n_nodes = (int)cls[i].nodes.size();
        S.node_off = node_off;
        S.deg = cls[i].deg;
        if (progs) {
            const Program &pr = (*progs)[i];
            S.op_off = (int)(*op_off)[i]; S.n_ops = (int)pr.ops.size();
            S.tab_off = (int)(*tab_off)[i]; S.tab_bytes = (int)pr.tables.size();
            S.n_in = pr.n_in; S.n_out = pr.n_out; S.n_slots = pr.n_slots;
            max_slots = std::max(max_slots, pr.n_slots);
            max_tab = std::max(max_tab, (int)pr.tables.size());
            max_out = std::max(max_out, pr.n_out);
        }
        blk += (S.n_nodes + P.nodes_per_block - 1) / P.nodes_per_block;
        node_off += S.n_nodes;
    }
    P.blocks_per_group = blk;
    P.slots_lds = max_slots;
    plan.out_slots = max_out;
    int slots_bytes = (max_slots + max_out) * kWave * 4;
    if (slots_bytes > 60 * 1024) return fail(LUTLDPC_ERR_UNSUPPORTED, "node program needs more than 60 KiB of LDS slots");
    plan.lds_tab = (slots_bytes + max_tab) <= 64 * 1024;
    plan.lds_bytes = slots_bytes + (plan.lds_tab ? max_tab : 0);
    plan.valid = true;
    return LUTLDPC_OK;
}

void build_fast_index(lutldpc_decoder *d) {
    d->fast_idx.clear(); d->vn_idx_off.clear(); d->cn_idx_off.clear();
    d->chain_idx_off.assign(d->cclass.size(), -1); d->vn_red_off.assign(d->vclass.size(), -1); d->vn_red_n.assign(d->vclass.size(), 0);
    for (auto &c : d->vclass) {
        d->vn_idx_off.push_back((int)d->fast_idx.size());
        for (int v : c.nodes) { d->fast_idx.push_back(v); d->fast_idx.push_back(d->vn_ptr[(size_t)v]); }
    }
    // ---- chain links (kernels_fast.hpp: cn_minsum_body<..., CHAIN>).  A degree-2 variable node whose two checks
    // are neighbours in their degree class AND fall into the same wave (the same run of npw checks) is updated by
    // that wave inside the check pass: both of its incoming messages are in registers there.  back[c] / fwd[c] =
    // the node check c shares with its predecessor / successor in the class list (+1, 0 = none).
    std::vector<int> edge_chk((size_t)d->E, -1), cls_of((size_t)d->nchk, -1), pos_of((size_t)d->nchk, -1);
    for (int c = 0; c < d->nchk; c++)
        for (int k = d->cn_ptr[(size_t)c]; k < d->cn_ptr[(size_t)c + 1]; k++) edge_chk[(size_t)d->cn_msg_idx[(size_t)k]] = c;
    for (size_t ci = 0; ci < d->cclass.size(); ci++)
        for (size_t j = 0; j < d->cclass[ci].nodes.size(); j++) { cls_of[(size_t)d->cclass[ci].nodes[j]] = (int)ci; pos_of[(size_t)d->cclass[ci].nodes[j]] = (int)j; }
    std::vector<int> back((size_t)d->nchk, 0), fwd((size_t)d->nchk, 0);
    std::vector<char> internal((size_t)d->nvar, 0);
    // checks per wave: a class of wide checks whose members are mostly linked by degree-2 nodes (the zigzag of a dual-diagonal
    // code) gets at least four checks per wave, so that three of four links fall inside a wave -- and twelve where the class is
    // large enough to keep 2048 runs per frame group (DVB-S2: 11 of 12 links inside a wave, +0.9 % over six checks per wave;
    // 18 per wave is slower again, tools/env_sweep.sh)
    d->cn_npw_class.assign(d->cclass.size(), 0);
    if (d->use_chain && d->min_lut) {
        std::vector<int> cand(d->cclass.size(), 0);
        for (int v = 0; v < d->nvar; v++) {
            if (d->dv[(size_t)v] != 2) continue;
            const int e0 = d->vn_ptr[(size_t)v], c1 = edge_chk[(size_t)e0], c2 = edge_chk[(size_t)e0 + 1];
            if (c1 < 0 || c2 < 0 || c1 == c2 || cls_of[(size_t)c1] != cls_of[(size_t)c2]) continue;
            if (std::abs(pos_of[(size_t)c1] - pos_of[(size_t)c2]) == 1) cand[(size_t)cls_of[(size_t)c1]]++;
        }
        for (size_t ci = 0; ci < d->cclass.size(); ci++)
            if (2 * cand[ci] >= (int)d->cclass[ci].nodes.size() && d->nodes_per_wave_cn <= 0) {
                const int n = (int)d->cclass[ci].nodes.size();
                d->cn_npw_class[ci] = std::max(4, d->npw_cn(d->cclass[ci].deg));
                if (!d->cn_edges_from_env) d->cn_npw_class[ci] = std::max(d->cn_npw_class[ci], std::min(12, n / 2048));
            }
    }
    if (d->use_chain && d->min_lut)
        for (int v = 0; v < d->nvar; v++) {
            if (d->dv[(size_t)v] != 2) continue;
            const int e0 = d->vn_ptr[(size_t)v];
            int c1 = edge_chk[(size_t)e0], c2 = edge_chk[(size_t)e0 + 1];
            if (c1 < 0 || c2 < 0 || c1 == c2 || cls_of[(size_t)c1] != cls_of[(size_t)c2]) continue;
            if (pos_of[(size_t)c1] > pos_of[(size_t)c2]) std::swap(c1, c2);
            const int deg = d->cclass[(size_t)cls_of[(size_t)c1]].deg, npw = d->npw_cn_class((size_t)cls_of[(size_t)c1]);
            if (deg < 2 || deg > fused_max_cn_deg() || pos_of[(size_t)c2] != pos_of[(size_t)c1] + 1 || pos_of[(size_t)c1] / npw != pos_of[(size_t)c2] / npw) continue;
            if (fwd[(size_t)c1] || back[(size_t)c2]) continue;
            fwd[(size_t)c1] = v + 1; back[(size_t)c2] = v + 1; internal[(size_t)v] = 1;
        }
    for (size_t ci = 0; ci < d->cclass.size(); ci++) {
        auto &c = d->cclass[ci];
        d->cn_idx_off.push_back((int)d->fast_idx.size());
        bool any = false;
        for (int cn : c.nodes) {
            std::vector<int> es;
            for (int k = 0; k < c.deg; k++) es.push_back(d->cn_msg_idx[(size_t)(d->cn_ptr[(size_t)cn] + k)]);
            // the order of a check's edges is free (min-sum is symmetric): chain edges go to fixed slots, back = 0, forward = 1
            auto to_slot = [&](int v1, size_t slot) {
                if (!v1) return;
                for (size_t k = 0; k < es.size(); k++)
                    if (es[k] == d->vn_ptr[(size_t)(v1 - 1)] || es[k] == d->vn_ptr[(size_t)(v1 - 1)] + 1) { std::swap(es[k], es[slot]); return; }
            };
            to_slot(back[(size_t)cn], 0); to_slot(fwd[(size_t)cn], 1);
            if (back[(size_t)cn] && fwd[(size_t)cn] && c.deg >= 2) {     // the second swap may have moved the back edge: restore slot 0
                const int vb = back[(size_t)cn] - 1;
                if (es[0] != d->vn_ptr[(size_t)vb] && es[0] != d->vn_ptr[(size_t)vb] + 1) to_slot(back[(size_t)cn], 0);
            }
            for (int e : es) d->fast_idx.push_back(e);
            any = any || back[(size_t)cn] || fwd[(size_t)cn];
        }
        if (any) {
            d->chain_idx_off[ci] = (int)d->fast_idx.size();
            for (int cn : c.nodes) { d->fast_idx.push_back(back[(size_t)cn]); d->fast_idx.push_back(fwd[(size_t)cn]); }
        }
    }
    // the node behind every entry of the check classes' edge tables, same order
    {
        std::vector<int> edge_node((size_t)d->E, 0);
        for (int v = 0; v < d->nvar; v++)
            for (int e = d->vn_ptr[(size_t)v]; e < d->vn_ptr[(size_t)v + 1]; e++) edge_node[(size_t)e] = v;
        d->cn_nidx_off.assign(d->cclass.size(), 0);
        for (size_t ci = 0; ci < d->cclass.size(); ci++) {
            const size_t off = (size_t)d->cn_idx_off[ci], cnt = d->cclass[ci].nodes.size() * (size_t)d->cclass[ci].deg;
            d->cn_nidx_off[ci] = (int)d->fast_idx.size();
            for (size_t j = 0; j < cnt; j++) d->fast_idx.push_back(edge_node[(size_t)d->fast_idx[off + j]]);
        }
    }
    // LDS-resident decoder (jit_resident.hpp): there a LANE owns a node, so the tables are transposed -- [k][node of the class] --
    // and 64 lanes reading entry k of 64 consecutive nodes touch 256 contiguous bytes.  Canonical edge order of the check
    // (ascending variable node, as cn_msg_idx: a CHKTREE consumes its inputs in that order).
    {
        std::vector<int> edge_node((size_t)d->E, 0);
        for (int v = 0; v < d->nvar; v++)
            for (int e = d->vn_ptr[(size_t)v]; e < d->vn_ptr[(size_t)v + 1]; e++) edge_node[(size_t)e] = v;
        d->cn_tidx_off.assign(d->cclass.size(), 0); d->cn_tnidx_off.assign(d->cclass.size(), 0); d->vn_tidx_off.assign(d->vclass.size(), 0);
        for (size_t ci = 0; ci < d->cclass.size(); ci++) {
            const auto &c = d->cclass[ci];
            const size_t n = c.nodes.size();
            d->cn_tidx_off[ci] = (int)d->fast_idx.size();
            for (int k = 0; k < c.deg; k++) for (size_t j = 0; j < n; j++) d->fast_idx.push_back(d->cn_msg_idx[(size_t)(d->cn_ptr[(size_t)c.nodes[j]] + k)]);
            d->cn_tnidx_off[ci] = (int)d->fast_idx.size();
            for (int k = 0; k < c.deg; k++) for (size_t j = 0; j < n; j++) d->fast_idx.push_back(edge_node[(size_t)d->cn_msg_idx[(size_t)(d->cn_ptr[(size_t)c.nodes[j]] + k)]]);
        }
        for (size_t vi = 0; vi < d->vclass.size(); vi++) {
            const auto &c = d->vclass[vi];
            d->vn_tidx_off[vi] = (int)d->fast_idx.size();
            for (int v : c.nodes) d->fast_idx.push_back(v);
            for (int v : c.nodes) d->fast_idx.push_back(d->vn_ptr[(size_t)v]);
        }
    }
    // variable passes that follow a chained check pass skip the nodes it already updated
    for (size_t vi = 0; vi < d->vclass.size(); vi++) {
        if (d->vclass[vi].deg != 2) continue;
        d->vn_red_off[vi] = (int)d->fast_idx.size();
        for (int v : d->vclass[vi].nodes)
            if (!internal[(size_t)v]) { d->fast_idx.push_back(v); d->fast_idx.push_back(d->vn_ptr[(size_t)v]); d->vn_red_n[vi]++; }
        d->chain_vclass = (int)vi;
    }
    d->n_chain_nodes = 0;
    for (char x : internal) d->n_chain_nodes += x;
    d->chain_internal.assign(internal.begin(), internal.end());
}

// Every entry of the dense index tables the specialised kernels read with scalar loads must address a row that exists:
// variable classes {node < N, first edge + degree <= E}, check classes edge < E, chain links node <= N (0 = none).
// Always on (O(E) at creation); an inconsistency here would be an out-of-range row in every launch.
int validate_fast_index(const lutldpc_decoder *d) {
    const size_t n = d->fast_idx.size();
    auto bad = [&](const std::string &what) { return fail(LUTLDPC_ERR_STATE, "index table check failed: " + what); };
    for (size_t i = 0; i < d->vclass.size(); i++) {
        const auto &c = d->vclass[i];
        const size_t off = (size_t)d->vn_idx_off[i];
        if (off + 2 * c.nodes.size() > n) return bad("variable class table outside the blob");
        for (size_t j = 0; j < c.nodes.size(); j++) {
            const int v = d->fast_idx[off + 2 * j], e = d->fast_idx[off + 2 * j + 1];
            if (v < 0 || v >= d->nvar || e < 0 || e + c.deg > d->E) return bad("variable node / first edge out of range");
        }
        if (d->vn_red_off[i] >= 0) {
            const size_t ro = (size_t)d->vn_red_off[i];
            if (ro + 2 * (size_t)d->vn_red_n[i] > n) return bad("reduced variable class table outside the blob");
            for (int j = 0; j < d->vn_red_n[i]; j++) {
                const int v = d->fast_idx[ro + 2 * (size_t)j], e = d->fast_idx[ro + 2 * (size_t)j + 1];
                if (v < 0 || v >= d->nvar || e < 0 || e + c.deg > d->E) return bad("reduced variable class entry out of range");
            }
        }
    }
    for (size_t i = 0; i < d->vclass.size() && i < d->vn_tidx_off.size(); i++) {
        const auto &c = d->vclass[i];
        const size_t off = (size_t)d->vn_tidx_off[i], m = c.nodes.size();
        if (off + 2 * m > n) return bad("transposed variable class table outside the blob");
        for (size_t j = 0; j < m; j++) {
            const int v = d->fast_idx[off + j], e = d->fast_idx[off + m + j];
            if (v < 0 || v >= d->nvar || e < 0 || e + c.deg > d->E) return bad("transposed variable class entry out of range");
        }
    }
    for (size_t i = 0; i < d->cclass.size() && i < d->cn_tidx_off.size(); i++) {
        const auto &c = d->cclass[i];
        const size_t cnt = c.nodes.size() * (size_t)c.deg, eo = (size_t)d->cn_tidx_off[i], no = (size_t)d->cn_tnidx_off[i];
        if (eo + cnt > n || no + cnt > n) return bad("transposed check class table outside the blob");
        for (size_t j = 0; j < cnt; j++) if (d->fast_idx[eo + j] < 0 || d->fast_idx[eo + j] >= d->E || d->fast_idx[no + j] < 0 || d->fast_idx[no + j] >= d->nvar) return bad("transposed check class entry out of range");
    }
    for (size_t i = 0; i < d->cclass.size(); i++) {
        const auto &c = d->cclass[i];
        const size_t off = (size_t)d->cn_idx_off[i], cnt = c.nodes.size() * (size_t)c.deg;
        if (off + cnt > n) return bad("check class table outside the blob");
        for (size_t j = 0; j < cnt; j++) if (d->fast_idx[off + j] < 0 || d->fast_idx[off + j] >= d->E) return bad("check edge out of range");
        if (i < d->cn_nidx_off.size()) {
            const size_t no = (size_t)d->cn_nidx_off[i];
            if (no + cnt > n) return bad("check class node table outside the blob");
            for (size_t j = 0; j < cnt; j++) if (d->fast_idx[no + j] < 0 || d->fast_idx[no + j] >= d->nvar) return bad("check node out of range");
        }
        if (d->chain_idx_off[i] >= 0) {
            const size_t co = (size_t)d->chain_idx_off[i];
            if (co + 2 * c.nodes.size() > n) return bad("chain link table outside the blob");
            for (size_t j = 0; j < 2 * c.nodes.size(); j++) if (d->fast_idx[co + j] < 0 || d->fast_idx[co + j] > d->nvar) return bad("chain link out of range");
        }
    }
    return LUTLDPC_OK;
}

int compile_all(lutldpc_decoder *d) {
    std::string err;
    build_fast_index(d);
    if (int rc = validate_fast_index(d)) return rc;
    // match trees to degree classes like set_trees (src/LDPC_Code_LUT.cpp:133-139,152-158):
    // VARTREE leaves == dv, CHKTREE leaves + 1 == dc, matched on tree set 0
    if (d->var_trees.empty()) return fail(LUTLDPC_ERR_ARG, "no variable-node trees");
    int n_sets = 0;
    for (int i = 0; i < d->max_iters_created; i++) n_sets = std::max(n_sets, d->iter_set[(size_t)i] + 1);
    if ((int)d->var_trees.size() < n_sets) return fail(LUTLDPC_ERR_ARG, "fewer variable tree sets than reuse_vec requires");
    for (auto &c : d->vclass) {
        c.tree_class = -1;
        for (size_t k = 0; k < d->var_trees[0].size(); k++) if (d->var_trees[0][k].num_leaves == c.deg) { c.tree_class = (int)k; break; }
        if (c.tree_class < 0) return fail(LUTLDPC_ERR_ARG, "no variable tree for degree " + std::to_string(c.deg));
    }
    if (!d->min_lut) {
        if ((int)d->chk_trees.size() < n_sets) return fail(LUTLDPC_ERR_ARG, "fewer check tree sets than reuse_vec requires");
        for (auto &c : d->cclass) {
            c.tree_class = -1;
            for (size_t k = 0; k < d->chk_trees[0].size(); k++) if (d->chk_trees[0][k].num_leaves + 1 == c.deg) { c.tree_class = (int)k; break; }
            if (c.tree_class < 0) return fail(LUTLDPC_ERR_ARG, "no check tree for degree " + std::to_string(c.deg));
        }
    }
    d->all_ops.clear(); d->all_tables.clear();
    auto add_set = [&](const std::vector<Tree> &trees, const std::vector<NodeClass> &cls, int kind,
                       std::vector<Program> &progs, PassPlan &plan, std::vector<FastClassPlan> *fast) -> int {
        progs.resize(cls.size());
        if (fast) fast->assign(cls.size(), FastClassPlan());
        std::vector<size_t> op_off(cls.size()), tab_off(cls.size());
        int node_off = 0;
        for (size_t i = 0; i < cls.size(); i++) {
            if (cls[i].tree_class >= (int)trees.size()) return fail(LUTLDPC_ERR_ARG, "tree set is missing a degree class");
            const Tree &t = trees[(size_t)cls[i].tree_class];
            std::string e;
            if (!compile_program(t, kind, cls[i].deg, progs[i], e))
                return fail(LUTLDPC_ERR_UNSUPPORTED, "degree " + std::to_string(cls[i].deg) + ": " + e);
            op_off[i] = d->all_ops.size(); tab_off[i] = d->all_tables.size();
            d->all_ops.insert(d->all_ops.end(), progs[i].ops.begin(), progs[i].ops.end());
            d->all_tables.insert(d->all_tables.end(), progs[i].tables.begin(), progs[i].tables.end());
            if (fast) {
                std::map<const TreeNode *, std::pair<uint32_t, uint32_t>> tab_of;
                for (auto &nt : progs[i].node_tabs) tab_of[nt.first] = {(uint32_t)tab_off[i] + nt.second[0], nt.second[1]};
                (*fast)[i] = plan_fast_vn(t, kind, cls[i].deg, tab_of, node_off, (int)cls[i].nodes.size());
                (*fast)[i].P.idx_off = d->vn_idx_off[i];
            }
            node_off += (int)cls[i].nodes.size();
        }
        return build_plan(d, cls, &progs, &op_off, &tab_off, plan);
    };
    // composed variants of the programs of one set (tables appended to the same blob)
    auto add_composed = [&](const std::vector<Tree> &trees, const std::vector<NodeClass> &cls, int kind, std::vector<Program> &progs,
                            std::vector<std::pair<int, int>> &tabs) -> int {
        progs.assign(cls.size(), Program()); tabs.assign(cls.size(), {0, 0});
        for (size_t i = 0; i < cls.size(); i++) {
            const Tree &t = trees[(size_t)cls[i].tree_class];
            const Tree tc = d->use_compose ? compose_tree(t, kind, (uint64_t)d->compose_space) : compose_tree(t, kind, 0);
            std::string e;
            if (!compile_program(tc, kind, cls[i].deg, progs[i], e)) return fail(LUTLDPC_ERR_UNSUPPORTED, "composed tree, degree " + std::to_string(cls[i].deg) + ": " + e);
            progs[i].node_tabs.clear();                         // (they point into the temporary tree)
            tabs[i] = {(int)d->all_tables.size(), (int)progs[i].tables.size()};
            d->all_tables.insert(d->all_tables.end(), progs[i].tables.begin(), progs[i].tables.end());
        }
        return LUTLDPC_OK;
    };
    size_t ns = (size_t)n_sets;
    d->var_prog.assign(ns, {}); d->dec_prog.assign(ns, {}); d->chk_prog.assign(ns, {});
    d->var_plan.assign(ns, {}); d->dec_plan.assign(ns, {}); d->chk_plan.assign(ns, {});
    d->var_fast.assign(ns, {}); d->dec_fast.assign(ns, {});
    d->chk_prog_full.assign(ns, {}); d->chk_full_tab.assign(ns, {});
    d->chk_prog_cf.assign(ns, {}); d->chk_tab_cf.assign(ns, {});
    d->var_prog_c.assign(ns, {}); d->dec_prog_c.assign(ns, {}); d->chk_prog_c.assign(ns, {});
    d->var_tab_c.assign(ns, {}); d->dec_tab_c.assign(ns, {}); d->chk_tab_c.assign(ns, {});
    for (size_t s = 0; s < ns; s++) {
        // a set is either message-update trees or (the last one) decision trees
        int type = d->var_trees[s].empty() ? TT_VAR : d->var_trees[s][0].type;
        int rc;
        if (type == TT_DEC) rc = add_set(d->var_trees[s], d->vclass, TT_DEC, d->dec_prog[s], d->dec_plan[s], &d->dec_fast[s]);
        else rc = add_set(d->var_trees[s], d->vclass, TT_VAR, d->var_prog[s], d->var_plan[s], &d->var_fast[s]);
        if (rc) return rc;
        if (!d->min_lut) {
            rc = add_set(d->chk_trees[s], d->cclass, TT_CHK, d->chk_prog[s], d->chk_plan[s], nullptr);
            if (rc) return rc;
            d->chk_prog_full[s].assign(d->cclass.size(), Program());
            d->chk_full_tab[s].assign(d->cclass.size(), {0, 0});
            for (size_t i = 0; i < d->cclass.size() && d->chk_full_labels; i++) {
                Program f;
                if (!chk_full_label_program(d->chk_prog[s][i], f) || f.tables.empty()) continue;
                while (d->all_tables.size() & 15) d->all_tables.push_back(0);
                d->chk_full_tab[s][i] = {(int)d->all_tables.size(), (int)f.tables.size()};
                d->all_tables.insert(d->all_tables.end(), f.tables.begin(), f.tables.end());
                d->chk_prog_full[s][i] = std::move(f);
            }
        }
        if (type == TT_DEC) rc = add_composed(d->var_trees[s], d->vclass, TT_DEC, d->dec_prog_c[s], d->dec_tab_c[s]);
        else rc = add_composed(d->var_trees[s], d->vclass, TT_VAR, d->var_prog_c[s], d->var_tab_c[s]);
        if (rc) return rc;
        if (!d->min_lut && (rc = add_composed(d->chk_trees[s], d->cclass, TT_CHK, d->chk_prog_c[s], d->chk_tab_c[s]))) return rc;
        if (!d->min_lut) {
            d->chk_prog_cf[s].assign(d->cclass.size(), Program());
            d->chk_tab_cf[s].assign(d->cclass.size(), {0, 0});
            for (size_t i = 0; i < d->cclass.size() && d->chk_full_labels; i++) {
                Program f;
                if (!chk_full_label_program(d->chk_prog_c[s][i], f) || f.tables.empty()) continue;
                while (d->all_tables.size() & 15) d->all_tables.push_back(0);
                d->chk_tab_cf[s][i] = {(int)d->all_tables.size(), (int)f.tables.size()};
                d->all_tables.insert(d->all_tables.end(), f.tables.begin(), f.tables.end());
                d->chk_prog_cf[s][i] = std::move(f);
            }
        }
    }
    if (d->min_lut) { int rc = build_plan(d, d->cclass, nullptr, nullptr, nullptr, d->cn_minsum_plan); if (rc) return rc; }
    return LUTLDPC_OK;
}

// is class i of a pass handled by a compile-time specialised kernel?
bool fast_covers(const lutldpc_decoder *d, const std::vector<FastClassPlan> &fast, size_t i) {
    return d->use_fast && i < fast.size() && fast[i].ok && fast[i].P.deg <= kFastMaxDeg;
}

// Process-wide registry of the run-time generated kernels, keyed by device + source text.  Decoders share the loaded
// modules (equal tree shapes give equal sources: no second hiprtc run), and a module is NEVER unloaded while the process
// lives: unloading frees executable device memory that the runtime hands to the next code object it loads, and the one
// device fault this library has shown ("The round-1 abort") was the first launch of a lazily loaded code object
// right after the modules of the previous decoder had been unloaded.  Bounded: beyond kJitRegistryMax distinct sources the
// generated kernels are simply not used (the interpreter runs instead).
struct JitRegistry { std::mutex mu; std::map<std::string, JitKernel> by_src; };
constexpr size_t kJitRegistryMax = 4096;
JitRegistry &jit_registry() { static JitRegistry *r = new JitRegistry; return *r; }     // never destroyed (see above)

// HIP loads the code object of a translation unit lazily, at the first launch of one of its kernels -- possibly in the
// middle of a decode and long after other modules came and went.  Load all of them at the first decoder creation on a
// device instead, while nothing of ours is in flight.
int preload_code_objects(int device) {
    static std::mutex mu;
    static std::vector<int> done;
    std::lock_guard<std::mutex> lock(mu);
    if (std::find(done.begin(), done.end(), device) != done.end()) return LUTLDPC_OK;
    hipFuncAttributes a;
    HIP_TRY(hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&frame_state_kernel)));             // this translation unit
    HIP_TRY((preload_fused<2, 0>())); HIP_TRY((preload_fused<2, 1>())); HIP_TRY((preload_fused<2, 2>())); HIP_TRY((preload_fused<2, 3>()));
    HIP_TRY((preload_vn_fast<TT_VAR, 1>())); HIP_TRY((preload_vn_fast<TT_VAR, 2>())); HIP_TRY((preload_vn_fast<TT_DEC, 2>()));
    HIP_TRY((preload_cn_fast<2>()));
    // the row permutation of the compaction uses 66 KB of dynamic LDS (kernels_compact.hpp)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&permute_rows_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kPermuteLdsBytes));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&permute_rows_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, kPermuteLdsBytes));
    HIP_TRY(hipDeviceSynchronize());
    done.push_back(device);
    return LUTLDPC_OK;
}

// jit.hpp: generate + compile + load a kernel for every variable / decision / CHKTREE class without a
// compile-time specialised one
void build_jit(lutldpc_decoder *d) {
    const size_t ns = d->var_plan.size();
    d->var_jit.assign(ns, {}); d->dec_jit.assign(ns, {}); d->chk_jit.assign(ns, {});
    if (!d->use_jit || !d->use_fast) return;
    for (size_t s = 0; s < ns; s++)
        for (int kind : {TT_VAR, TT_DEC, TT_CHK}) {
            if (kind == TT_CHK && d->min_lut) continue;
            const PassPlan &plan = kind == TT_VAR ? d->var_plan[s] : kind == TT_DEC ? d->dec_plan[s] : d->chk_plan[s];
            if (!plan.valid) continue;
            const auto &progs = kind == TT_VAR ? d->var_prog[s] : kind == TT_DEC ? d->dec_prog[s] : d->chk_prog[s];
            const auto &cls = kind == TT_CHK ? d->cclass : d->vclass;
            auto &out = kind == TT_VAR ? d->var_jit[s] : kind == TT_DEC ? d->dec_jit[s] : d->chk_jit[s];
            out.assign(cls.size(), nullptr);
            for (size_t i = 0; i < cls.size(); i++) {
                if (kind != TT_CHK && fast_covers(d, kind == TT_VAR ? d->var_fast[s] : d->dec_fast[s], i)) continue;
                std::string src, err;
                const bool full = kind == TT_CHK && s < d->chk_full_tab.size() && i < d->chk_full_tab[s].size() && d->chk_full_tab[s][i].second > 0;
                const bool gen = kind == TT_CHK ? (full ? jit_cn_source(d->chk_prog_full[s][i], cls[i].deg, d->pack, d->chk_full_tab[s][i].second, src, err)
                                                        : jit_cn_source(progs[i], cls[i].deg, d->pack, plan.P.seg[i].tab_bytes, src, err))
                                                : jit_vn_source(progs[i], kind, cls[i].deg, d->pack, plan.P.seg[i].tab_bytes, src, err);
                if (!gen) { d->jit_log = err; continue; }
                JitRegistry &reg = jit_registry();
                std::lock_guard<std::mutex> lock(reg.mu);
                const std::string key = std::to_string(d->device) + "\n" + src;
                auto it = reg.by_src.find(key);
                if (it == reg.by_src.end()) {
                    if (reg.by_src.size() >= kJitRegistryMax) { d->jit_log = "generated-kernel registry full"; continue; }
                    std::vector<char> code;
                    JitKernel k;
                    std::string log;
                    if (!jit_compile(src, code, log) || !jit_load(code, k, log)) { d->jit_log = log; reg.by_src[key] = JitKernel(); continue; }
                    it = reg.by_src.emplace(key, k).first;
                }
                if (it->second.ok()) out[i] = &it->second;        // (std::map nodes are stable: the pointer outlives the lock)
            }
        }
}

int upload_static(lutldpc_decoder *d) {
    HIP_TRY(hipSetDevice(d->device));
    if (int rc = preload_code_objects(d->device)) return rc;
    HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    HIP_TRY(d->d_vn_ptr.upload(d->vn_ptr));
    HIP_TRY(d->d_cn_ptr.upload(d->cn_ptr));
    HIP_TRY(d->d_cn_idx.upload(d->cn_msg_idx));
    {   // syndrome kernel: node of every check-edge, bit 31 = last edge of its check, 8 entries of padding
        std::vector<int32_t> f((size_t)d->E + 8, 0);
        for (int c = 0; c < d->nchk; c++)
            for (int k = d->cn_ptr[(size_t)c]; k < d->cn_ptr[(size_t)c + 1]; k++)
                f[(size_t)k] = (int32_t)((uint32_t)d->cn_vn[(size_t)k] | (k + 1 == d->cn_ptr[(size_t)c + 1] ? 0x80000000u : 0u));
        HIP_TRY(d->d_cn_vn.upload(f));
    }
    HIP_TRY(d->d_vn_list.upload(d->vn_list));
    HIP_TRY(d->d_cn_list.upload(d->cn_list));
    HIP_TRY(d->d_fast_idx.upload(d->fast_idx));
    HIP_TRY(d->d_chain_internal.upload(d->chain_internal));
    d->edge_vn.resize((size_t)d->E);
    for (int v = 0; v < d->nvar; v++) for (int e = d->vn_ptr[(size_t)v]; e < d->vn_ptr[(size_t)v + 1]; e++) d->edge_vn[(size_t)e] = v;
    HIP_TRY(d->d_edge_vn.upload(d->edge_vn));
    HIP_TRY(d->d_ops.upload(d->all_ops));
    {   // pad the table blob so that dword staging never reads past the end
        std::vector<uint8_t> t = d->all_tables;
        t.resize((t.size() + 3) / 4 * 4 + 16, 0);
        HIP_TRY(d->d_tables.upload(t));
    }
    build_jit(d);
    return LUTLDPC_OK;
}

int place_rows(lutldpc_decoder *d, int Bpad);
void make_describe(lutldpc_decoder *d);
int ensure_batch(lutldpc_decoder *d, int B) {
    int Bpad = d->bpad(B);
    if (Bpad <= d->Bcap) return LUTLDPC_OK;
    size_t G = (size_t)(Bpad / d->tile());
    d->drop_graphs();                                // the captured launches hold the old buffer addresses,
    d->drop_plans();                                 // the launch plans the strides of the flag buffers
    HIP_TRY(d->d_msgs.alloc(G * (size_t)d->E * kRowBytes));
    HIP_TRY(d->d_cha_t.alloc(G * (size_t)d->nvar * kRowBytes));
    HIP_TRY(d->d_msg0_t.alloc(G * (size_t)d->nvar * kRowBytes));
    HIP_TRY(d->d_hard.alloc(G * (size_t)d->nvar * kRowBytes));
    HIP_TRY(d->d_state.alloc((size_t)Bpad));
    HIP_TRY(d->d_vfail.alloc((size_t)Bpad * kVfailSlots * 2));  // two buffers (skewed pipeline: this / next exit test) of kVfailSlots copies, Bcap bytes apart
    HIP_TRY(d->d_iters.alloc((size_t)Bpad));
    HIP_TRY(d->d_frame_of.alloc((size_t)Bpad)); HIP_TRY(d->d_perm.alloc((size_t)Bpad)); HIP_TRY(d->d_tmp3.alloc((size_t)Bpad * 3));
    HIP_TRY(d->d_ctl.alloc(8)); HIP_TRY(d->d_slot_of.alloc((size_t)Bpad)); HIP_TRY(d->d_iters_tmp.alloc((size_t)Bpad));
    HIP_TRY(d->d_grp.alloc(G));
    // Every row exists with a defined content from the start: with the first check pass reading the initial-message rows
    // (first_from_nodes) the edge rows of PAD frames and of frames that passed the test on the channel decisions are never
    // written while their group still has active frames, and the variable passes compute on all lanes (results masked).
    HIP_TRY(hipMemsetAsync(d->d_msgs.p, 0, d->d_msgs.bytes(), d->stream));
    HIP_TRY(hipMemsetAsync(d->d_hard.p, 0, d->d_hard.bytes(), d->stream));
    HIP_TRY(hipMemsetAsync(d->d_cha_t.p, 0, d->d_cha_t.bytes(), d->stream));
    HIP_TRY(hipMemsetAsync(d->d_msg0_t.p, 0, d->d_msg0_t.bytes(), d->stream));
    d->Bcap = Bpad;
    if (getenv("LUTLDPC_DEBUG_ADDR"))        // (tools/level_probe_realloc.py: where did the row buffers of this handle land?)
        fprintf(stderr, "lutldpc rows: msgs %p cha %p msg0 %p hard %p (%zu MB of messages)\n", (void *)d->d_msgs.p, (void *)d->d_cha_t.p, (void *)d->d_msg0_t.p, (void *)d->d_hard.p, d->d_msgs.bytes() >> 20);
    return place_rows(d, Bpad);
}

// Always on, O(1), before every decode: the batch buffers every kernel addresses rows in exist and hold Bpad frames.  (The one
// device fault this library has shown was a valid edge row off a NULL message base)
int check_batch_buffers(const lutldpc_decoder *d, int Bpad) {
    const size_t G = (size_t)(Bpad / d->tile());
    auto bad = [&](const char *what) { return fail(LUTLDPC_ERR_STATE, std::string("batch buffer check failed before the decode: ") + what); };
    if (Bpad <= 0 || Bpad > d->Bcap || Bpad % d->tile()) return bad("batch larger than the allocation");
    if (!d->d_msgs.p || d->d_msgs.n < G * (size_t)d->E * kRowBytes) return bad("message rows");
    if (!d->d_cha_t.p || d->d_cha_t.n < G * (size_t)d->nvar * kRowBytes) return bad("channel rows");
    if (!d->d_msg0_t.p || d->d_msg0_t.n < G * (size_t)d->nvar * kRowBytes) return bad("initial-message rows");
    if (!d->d_hard.p || d->d_hard.n < G * (size_t)d->nvar * kRowBytes) return bad("decided-bit rows");
    if (!d->d_state.p || d->d_state.n < (size_t)Bpad || !d->d_iters.p || d->d_iters.n < (size_t)Bpad) return bad("frame state");
    if (!d->d_vfail.p || d->d_vfail.n < (size_t)d->Bcap * kVfailSlots * 2) return bad("flag buffers");
    if (!d->d_fast_idx.p || !d->d_tables.p || !d->stream) return bad("static tables / stream");
    return LUTLDPC_OK;
}

// LUTLDPC_VALIDATE=1: additionally wait for the launch(es) just issued, so that a device fault is reported by the launch site
// that caused it (function and line), whatever the kernel -- not only the fused ones (no graph capture in that mode)
#define LAUNCH_CHECK()                                                                              \
    do {                                                                                            \
        hipError_t e_ = hipGetLastError();                                                          \
        if (e_ != hipSuccess) return fail(LUTLDPC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e_)); \
        if (d->validate) {                                                                          \
            e_ = hipStreamSynchronize(d->stream);                                                   \
            if (e_ == hipSuccess) e_ = hipGetLastError();                                           \
            if (e_ != hipSuccess) return fail(LUTLDPC_ERR_HIP, std::string("launch failed on the device (") + __func__ + ":" + std::to_string(__LINE__) + "): " + hipGetErrorString(e_)); \
        }                                                                                           \
    } while (0)

// device copy of a parameter structure (see lutldpc_decoder::ParamArena); nullptr + last error on failure
constexpr size_t kParamChunk = 1u << 20;
const void *dev_param_bytes(lutldpc_decoder *d, const void *src, size_t n) {
    auto &A = d->params;
    std::string key((const char *)src, n);
    auto it = A.off_of.find(key);
    size_t off;
    if (it == A.off_of.end()) {
        const size_t need = (n + 63) / 64 * 64;
        if (A.chunks.empty() || A.used + need > A.cap) {
            if (A.chunks.size() >= 64) {          // 64 MB of distinct parameter blocks: a caller with ever-changing shapes -- start over
                (void)hipStreamSynchronize(d->stream);
                d->drop_graphs();
                A.release();
            }
            std::unique_ptr<DevBuf<uint8_t>> c(new DevBuf<uint8_t>());
            if (c->alloc(std::max(kParamChunk, need)) != hipSuccess) { fail(LUTLDPC_ERR_HIP, "parameter arena: hipMalloc failed"); return nullptr; }
            A.chunk_base.push_back(A.cap);
            A.used = A.cap;
            A.cap += c->n;
            A.chunks.push_back(std::move(c));
        }
        off = A.used;
        A.used += need;
        it = A.off_of.emplace(std::move(key), off).first;
        const size_t ci = A.chunks.size() - 1;
        if (hipMemcpyAsync(A.chunks[ci]->p + (off - A.chunk_base[ci]), it->first.data(), n, hipMemcpyHostToDevice, d->stream) != hipSuccess) {
            fail(LUTLDPC_ERR_HIP, "parameter arena: upload failed");
            return nullptr;
        }
    } else off = it->second;
    size_t ci = A.chunks.size() - 1;
    while (ci > 0 && A.chunk_base[ci] > off) ci--;
    return A.chunks[ci]->p + (off - A.chunk_base[ci]);
}
template <class T> const T *dev_param(lutldpc_decoder *d, const T &v) { return static_cast<const T *>(dev_param_bytes(d, &v, sizeof(T))); }
#define DEV_PARAM(var, d, v)                      \
    const auto *var = dev_param((d), (v));        \
    if (!var) return LUTLDPC_ERR_HIP

// instantiate a launch for the decoder's packing
#define PACK_DISPATCH(d, ...)                        \
    do {                                             \
        if ((d)->pack == 2) { constexpr int PK = 2; __VA_ARGS__; } \
        else { constexpr int PK = 1; __VA_ARGS__; }  \
    } while (0)

// frames f0 .. f1-1 (both multiples of 256); default: the whole padded batch
// `sel`: which of the two flag buffers the exit test reads and clears (always 0 outside the skewed pipeline)
int launch_state(lutldpc_decoder *d, int B, int Bpad, int mode, int value, int f0 = 0, int f1 = -1, int sel = 0) {
    Timed t(d, LUTLDPC_K_LAYOUT);
    if (f1 < 0) f1 = Bpad;
    if (f1 <= f0) return LUTLDPC_OK;
    if (mode == 0) HIP_TRY(hipMemsetAsync(d->d_vfail.p + (size_t)kVfailSlots * d->Bcap, 0, (size_t)kVfailSlots * d->Bcap, d->stream));
    launch_k(frame_state_kernel, dim3((unsigned)((f1 - f0) / 256)), dim3(256), 0, d->stream,
                       d->d_state.p, d->d_vfail.p + (size_t)sel * kVfailSlots * d->Bcap, d->d_iters.p, B, f0, f1, mode, value, d->Bcap);
    LAUNCH_CHECK();
    return LUTLDPC_OK;
}

int launch_syndrome(lutldpc_decoder *d, int G, int sel = 0) {
    Timed t(d, LUTLDPC_K_SYNDROME);
    const int cpw = 8;
    unsigned bx = (unsigned)((d->nchk + 4 * cpw - 1) / (4 * cpw));
    PACK_DISPATCH(d, launch_k(syndrome_bits_kernel<PK>, dim3(bx, (unsigned)G), dim3(256), 0, d->stream, d->d_hard.p,
                       reinterpret_cast<const uint32_t *>(d->d_state.p), reinterpret_cast<uint32_t *>(d->d_vfail.p + (size_t)sel * kVfailSlots * d->Bcap),
                       d->d_cn_ptr.p, reinterpret_cast<const uint32_t *>(d->d_cn_vn.p), d->nchk, d->nvar, cpw, d->Bcap / 4, -1, (const int32_t *)nullptr, (int32_t *)nullptr));
    LAUNCH_CHECK();
    return LUTLDPC_OK;
}
// The test on the channel decisions (src/LDPC_Code_LUT.cpp:275-279) straight off the channel-label rows: a one-wave probe over
// the first 64 checks of every group, then the full pass, which skips the groups whose frames have all failed in the probe.
// The decided bits of the frames that pass are written at the end of the decode (hard_from_labels_masked_kernel).
int launch_syndrome_of_labels(lutldpc_decoder *d, int G) {
    Timed t(d, LUTLDPC_K_SYNDROME);
    const int cpw = 8, sbit = __builtin_ctz((unsigned)(d->Nq_Cha / 2));
    HIP_TRY(hipMemsetAsync(d->d_grp.p, 0, sizeof(int32_t) * (size_t)G, d->stream));
#define SYN_ARGS(CPW, SKIP, OUT) d->d_cha_t.p, reinterpret_cast<const uint32_t *>(d->d_state.p), reinterpret_cast<uint32_t *>(d->d_vfail.p), d->d_cn_ptr.p, \
                     reinterpret_cast<const uint32_t *>(d->d_cn_vn.p), d->nchk, d->nvar, CPW, d->Bcap / 4, sbit, SKIP, OUT
    PACK_DISPATCH(d, launch_k(syndrome_bits_kernel<PK>, dim3(1u, (unsigned)G), dim3(64), 0, d->stream, SYN_ARGS(64, (const int32_t *)nullptr, d->d_grp.p)));
    unsigned bx = (unsigned)((d->nchk + 4 * cpw - 1) / (4 * cpw));
    PACK_DISPATCH(d, launch_k(syndrome_bits_kernel<PK>, dim3(bx, (unsigned)G), dim3(256), 0, d->stream, SYN_ARGS(cpw, (const int32_t *)d->d_grp.p, (int32_t *)nullptr)));
#undef SYN_ARGS
    LAUNCH_CHECK();
    return LUTLDPC_OK;
}

// frame-major [B][N] <-> rows; the dword-vectorised kernels need N % 4 == 0 and a 4-byte aligned buffer
int launch_transpose_in(lutldpc_decoder *d, const uint8_t *src, uint8_t *dst_rows, int B, int G, int limit) {
    const int N = d->nvar;
    if (N % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 3u) == 0)
        PACK_DISPATCH(d, launch_k(transpose_in_vec_kernel<PK>, dim3((unsigned)((N + 127) / 128), (unsigned)G), dim3(256), 0, d->stream, src, dst_rows, B, N, limit));
    else
        PACK_DISPATCH(d, launch_k(transpose_in_kernel<PK>, dim3((unsigned)((N + 31) / 32), (unsigned)G), dim3(256), 0, d->stream, src, dst_rows, B, N, limit));
    LAUNCH_CHECK();
    return LUTLDPC_OK;
}
int launch_transpose_out(lutldpc_decoder *d, const uint8_t *src_rows, uint8_t *dst, int B, int G, int rows = 0) {
    const int N = rows > 0 ? rows : d->nvar;
    if (N % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) & 3u) == 0)
        PACK_DISPATCH(d, launch_k(transpose_out_vec_kernel<PK>, dim3((unsigned)((N + 127) / 128), (unsigned)G), dim3(256), 0, d->stream, src_rows, dst, B, N));
    else
        PACK_DISPATCH(d, launch_k(transpose_out_kernel<PK>, dim3((unsigned)((N + 31) / 32), (unsigned)G), dim3(256), 0, d->stream, src_rows, dst, B, N));
    LAUNCH_CHECK();
    return LUTLDPC_OK;
}

// keep only the classes flagged in `keep` (the others were handled by specialised kernels)
PassParams filter_params(const PassParams &P, const std::vector<char> &keep) {
    PassParams Q = P;
    Q.n_seg = 0;
    int blk = 0;
    for (int i = 0; i < P.n_seg; i++) {
        if (!keep[(size_t)i]) continue;
        PassSeg S = P.seg[i];
        S.block_begin = blk;
        blk += (S.n_nodes + P.nodes_per_block - 1) / P.nodes_per_block;
        Q.seg[Q.n_seg++] = S;
    }
    Q.blocks_per_group = blk;
    return Q;
}

template <int KIND>
int launch_tree_pass(lutldpc_decoder *d, PassPlan &plan, std::vector<FastClassPlan> *fast, const std::vector<const JitKernel *> *jit, int G, int nz, int check, int write_hard, int kind_id,
                     const std::vector<std::pair<int, int>> *jit_tabs = nullptr) {
    if (!plan.valid) return fail(LUTLDPC_ERR_STATE, "pass plan missing for this tree set");
    Timed t(d, kind_id);
    PassParams P = plan.P;
    P.G = G; P.nz = nz; P.check = check; P.write_hard = write_hard; P.vfail_stride_w = d->Bcap / 4;
    std::vector<char> keep((size_t)P.n_seg, 1);
    bool any = false;
    // specialised kernels take the classes they know, one launch per degree class
    if (d->use_fast && fast && KIND != TT_CHK)
        for (int i = 0; i < P.n_seg; i++) {
            if (!(*fast)[(size_t)i].ok) continue;
            bool ok = false;
            FastParams FP = (*fast)[(size_t)i].P;
            fill_vn_fast(FP, G, nz, check, write_hard, d->npw_vn(FP.deg), d->E, d->nvar, d->Bcap / 4);
            DEV_PARAM(dFP, d, FP);
            PACK_DISPATCH(d, ok = launch_vn_fast<KIND, PK>(d->stream, FP, dFP, d->d_msgs.p, d->d_cha_t.p, d->d_hard.p,
                                     reinterpret_cast<const uint32_t *>(d->d_state.p), reinterpret_cast<uint32_t *>(d->d_vfail.p), d->d_tables.p, d->d_fast_idx.p));
            if (ok) keep[(size_t)i] = 0;
        }
    // run-time generated kernels (jit.hpp) for the classes without a compile-time one
    if (jit)
        for (int i = 0; i < P.n_seg && (size_t)i < jit->size(); i++) {
            const JitKernel *k = (*jit)[(size_t)i];
            if (!keep[(size_t)i] || !k) continue;
            FastParams F{};
            F.n_nodes = P.seg[i].n_nodes; F.deg = P.seg[i].deg;
            F.idx_off = KIND == TT_CHK ? d->cn_idx_off[(size_t)i] : d->vn_idx_off[(size_t)i];
            F.nodes_per_wave = KIND == TT_CHK ? d->npw_cn(F.deg) : d->npw_vn(F.deg); F.waves_per_group = (F.n_nodes + F.nodes_per_wave - 1) / F.nodes_per_wave;
            F.G = G; F.E = d->E; F.N = d->nvar; F.g0 = 0; F.nz = nz; F.check = check; F.write_hard = write_hard; F.vfail_stride_w = d->Bcap / 4;
            F.tab_off[0] = P.seg[i].tab_off; F.tab_len[0] = P.seg[i].tab_bytes;
            if (jit_tabs && (size_t)i < jit_tabs->size() && (*jit_tabs)[(size_t)i].second > 0) { F.tab_off[0] = (*jit_tabs)[(size_t)i].first; F.tab_len[0] = (*jit_tabs)[(size_t)i].second; }   // the kernel was generated for these tables
            uint8_t *msgs = d->d_msgs.p, *hard = d->d_hard.p;
            const uint8_t *cha = d->d_cha_t.p, *tables = d->d_tables.p;
            const uint32_t *state_w = reinterpret_cast<const uint32_t *>(d->d_state.p);
            uint32_t *vfail_w = reinterpret_cast<uint32_t *>(d->d_vfail.p);
            const int32_t *fidx = d->d_fast_idx.p;
            DEV_PARAM(dF, d, F);
            void *args[] = {&dF, &msgs, &cha, &hard, &state_w, &vfail_w, &tables, &fidx};
            const unsigned blocks = (unsigned)((F.waves_per_group * G + 3) / 4);
            HIP_TRY(hipModuleLaunchKernel(k->fn, blocks, 1, 1, 256, 1, 1, 0, d->stream, args, nullptr));
            keep[(size_t)i] = 0;
        }
    for (char k : keep) any = any || k;
    if (any) {
        P = filter_params(P, keep);
        DEV_PARAM(dP, d, P);
        dim3 grid((unsigned)(P.blocks_per_group * G)), block(64);
        const int32_t *list = KIND == TT_CHK ? d->d_cn_list.p : d->d_vn_list.p;
        const int32_t *ptr = KIND == TT_CHK ? d->d_cn_ptr.p : d->d_vn_ptr.p;
        if (plan.lds_tab)
            PACK_DISPATCH(d, launch_k(tree_pass_kernel<KIND, true, PK>, grid, block, (size_t)plan.lds_bytes, d->stream, dP, d->d_msgs.p, d->d_cha_t.p,
                               d->d_hard.p, reinterpret_cast<const uint32_t *>(d->d_state.p), reinterpret_cast<uint32_t *>(d->d_vfail.p),
                               d->d_ops.p, d->d_tables.p, list, ptr, d->d_cn_idx.p, plan.out_slots));
        else
            PACK_DISPATCH(d, launch_k(tree_pass_kernel<KIND, false, PK>, grid, block, (size_t)plan.lds_bytes, d->stream, dP, d->d_msgs.p, d->d_cha_t.p,
                               d->d_hard.p, reinterpret_cast<const uint32_t *>(d->d_state.p), reinterpret_cast<uint32_t *>(d->d_vfail.p),
                               d->d_ops.p, d->d_tables.p, list, ptr, d->d_cn_idx.p, plan.out_slots));
    }
    LAUNCH_CHECK();
    return LUTLDPC_OK;
}

int launch_cn_minsum(lutldpc_decoder *d, int G, int nz, int check) {
    Timed t(d, LUTLDPC_K_CN_PASS);
    PassParams P = d->cn_minsum_plan.P;
    P.G = G; P.nz = nz; P.check = check; P.vfail_stride_w = d->Bcap / 4;
    std::vector<char> keep((size_t)P.n_seg, 1);
    bool any = false;
    if (d->use_fast)
        for (int i = 0; i < P.n_seg; i++) {
            bool ok = false;
            FastParams FP;
            if (!fill_cn_fast(FP, P.seg[i].deg, P.seg[i].n_nodes, d->cn_idx_off[(size_t)i], G, d->E, nz, check, d->npw_cn(P.seg[i].deg), d->Bcap / 4)) continue;
            DEV_PARAM(dFP, d, FP);
            PACK_DISPATCH(d, ok = launch_cn_fast<PK>(d->stream, FP, dFP, d->d_msgs.p,
                               reinterpret_cast<const uint32_t *>(d->d_state.p), reinterpret_cast<uint32_t *>(d->d_vfail.p), d->d_fast_idx.p));
            if (ok) keep[(size_t)i] = 0;
        }
    for (char k : keep) any = any || k;
    if (any) {
        P = filter_params(P, keep);
        DEV_PARAM(dP, d, P);
        PACK_DISPATCH(d, launch_k(cn_minsum_generic_kernel<PK>, dim3((unsigned)(P.blocks_per_group * G)), dim3(64), 0, d->stream, dP, d->d_msgs.p,
                           reinterpret_cast<const uint32_t *>(d->d_state.p), reinterpret_cast<uint32_t *>(d->d_vfail.p),
                           d->d_cn_list.p, d->d_cn_ptr.p, d->d_cn_idx.p));
    }
    LAUNCH_CHECK();
    return LUTLDPC_OK;
}

// ----------------------------------------------------------------------------- skewed two-half pipeline
// (kernels_fast.hpp: pass_fused_kernel).  Half A = groups [0, GA), half B = [GA, G).  Each half runs
// the reference's sequence  CN(0) VN(0) CN(1) ... CN(I-1)  (src/LDPC_Code_LUT.cpp:301-338); B lags A by
// one pass, so every launch pairs a check pass of one half with a variable pass of the other.
struct HalfRange { int g0, G; };

bool skew_eligible(const lutldpc_decoder *d) {
    if (!d->min_lut || !d->use_fast) return false;
    if ((int)(d->cclass.size() + d->vclass.size()) > kFusedMaxRoles) return false;
    int max_cn = 0, max_vn = 0;
    for (auto &c : d->cclass) { if (c.deg < 2) return false; max_cn = std::max(max_cn, c.deg); }
    for (auto &c : d->vclass) max_vn = std::max(max_vn, c.deg);
    if (fused_bucket(max_vn, max_cn) < 0) return false;
    for (int nq : d->Nq_Msg) if (!is_pow2(nq / 2) || nq / 2 > 64) return false;
    for (size_t s = 0; s < d->var_fast.size(); s++) {
        if (d->var_plan[s].valid == false) continue;          // decision-only set
        for (auto &f : d->var_fast[s])
            if (!f.ok || f.P.n_tables > kFusedMaxTables) return false;
    }
    return true;
}

// Are the decided bits of early-terminated frames recovered at the end (hard_from_frozen_kernel) instead of being stored by
// every variable pass?  Min-sum checks, one message alphabet, and -- in the skewed pipeline -- chain fusion on in every
// iteration or in none (the nodes it updates get their bits from the check pass).  `chain_skip`: those nodes are skipped by
// the recovery.  (Compaction drops the messages of finished frames: its check points run the recovery first.)
bool chain_active(const lutldpc_decoder *d, int set);
bool late_hard_active(const lutldpc_decoder *d, bool skewed, bool *chain_skip) {
    if (chain_skip) *chain_skip = false;
    if (!d->late_hard || !d->psc || !d->min_lut) return false;
    for (int i = 1; i < d->max_iters; i++) if (d->Nq_Msg[(size_t)i] != d->Nq_Msg[0]) return false;
    if (!skewed) return true;
    int on = 0, off = 0;
    for (int ii = 0; ii + 1 < d->max_iters; ii++) (chain_active(d, d->iter_set[(size_t)ii]) ? on : off)++;
    if (on && off) return false;
    if (chain_skip) *chain_skip = on > 0;
    return true;
}

// chain fusion applies to a check pass that is followed by a variable pass (not the last iteration) when the degree-2
// class has the compile-time kernel (its root table is staged)
bool chain_active(const lutldpc_decoder *d, int set) {
    if (!d->use_chain || d->chain_vclass < 0 || d->n_chain_nodes == 0) return false;
    const FastClassPlan &f = d->var_fast[(size_t)set][(size_t)d->chain_vclass];
    return f.ok && f.P.n_tables == 1 && f.P.tab_len[0] <= 1024;
}

void add_cn_roles(const lutldpc_decoder *d, FusedParams &FP, std::vector<int> &blocks, HalfRange h, int ii, int check) {
    const int I = d->max_iters, nz = d->Nq_Msg[(size_t)ii] / 2;
    const int buf_w = kVfailSlots * d->Bcap / 4;                      // words per flag buffer
    const bool on = ii != I - 1 && chain_active(d, d->iter_set[(size_t)ii]);
    // decided bits of the nodes updated here: stored by the check pass that reads their messages, unless they are recovered at
    // the end with everything else (late_hard_active + chain_hard_kernel)
    const bool hard = d->psc && ii >= 1 && chain_active(d, d->iter_set[(size_t)(ii - 1)]) && !late_hard_active(d, true, nullptr);
    for (size_t i = 0; i < d->cclass.size(); i++) {
        RoleParams R{};
        R.vfail_off_w = (ii & 1) * buf_w;                             // parity flags: this iteration's exit test
        if ((on || hard) && d->chain_idx_off[i] >= 0) {
            R.chain.idx_off = d->chain_idx_off[i];
            R.chain.hard = hard ? 1 : 0;
            if (on) {
                const FastParams &F2 = d->var_fast[(size_t)d->iter_set[(size_t)ii]][(size_t)d->chain_vclass].P;
                R.chain.on = 1;
                R.chain.tab_off = F2.tab_off[0]; R.chain.tab_len = F2.tab_len[0]; R.chain.tab_shift = F2.tab_shift[0];
                R.chain.check = d->psc ? 1 : 0;
                R.chain.vfail_off_w = ((ii + 1) & 1) * buf_w;         // unanimity of the nodes updated here: the next exit test
                R.chain.sbit_out = __builtin_ctz((unsigned)(d->Nq_Msg[(size_t)(ii + 1)] / 2) | 0x100u);
            }
        }
        const int npw = d->npw_cn_class(i);
        R.kind = 0; R.deg = d->cclass[i].deg; R.g0 = h.g0; R.G = h.G;
        R.n_nodes = (int)d->cclass[i].nodes.size(); R.nodes_per_wave = npw;
        R.waves_per_group = (R.n_nodes + npw - 1) / npw;
        R.idx_off = d->cn_idx_off[i]; R.E = d->E; R.N = d->nvar; R.nz = nz; R.check = check; R.vfail_stride_w = d->Bcap / 4;
        if (ii == 0 && d->first_from_nodes) { R.first = 1; R.nidx_off = d->cn_nidx_off[i]; }
        FP.role[FP.n_roles++] = R;
        blocks.push_back((R.waves_per_group * h.G + 3) / 4);
    }
}
void add_vn_roles(const lutldpc_decoder *d, FusedParams &FP, std::vector<int> &blocks, HalfRange h, int ii, int check, int write_hard) {
    const int set = d->iter_set[(size_t)ii], nz = d->Nq_Msg[(size_t)(ii + 1)] / 2;
    const bool chained = chain_active(d, set);
    const int buf_w = kVfailSlots * d->Bcap / 4;
    for (size_t i = 0; i < d->vclass.size(); i++) {
        const FastParams &F = d->var_fast[(size_t)set][i].P;
        const int npw = d->npw_vn(F.deg);
        RoleParams R{};
        R.kind = 1; R.deg = F.deg; R.g0 = h.g0; R.G = h.G;
        R.n_nodes = F.n_nodes; R.nodes_per_wave = npw;
        R.idx_off = F.idx_off;
        if (chained && (int)i == d->chain_vclass) { R.n_nodes = d->vn_red_n[i]; R.idx_off = d->vn_red_off[i]; }   // the others were updated by the check pass
        R.waves_per_group = (R.n_nodes + npw - 1) / npw; R.E = d->E; R.N = d->nvar; R.nz = nz; R.shift_msg = F.shift_msg; R.check = check; R.write_hard = write_hard; R.vfail_stride_w = d->Bcap / 4;
        R.vfail_off_w = ((ii + 1) & 1) * buf_w;                       // unanimity flags: the exit test after the NEXT check pass
        for (int t = 0; t < F.n_tables; t++) { R.tab_off[t] = F.tab_off[t]; R.tab_len[t] = F.tab_len[t]; R.tab_shift[t] = F.tab_shift[t]; }
        FP.role[FP.n_roles++] = R;
        blocks.push_back((R.waves_per_group * h.G + 3) / 4);
    }
}

// interleave the blocks of all roles evenly over the launch: block j of a role with n blocks sits at
// position (j + 1/2) / n of the timeline
// `reverse`: the blocks of every role in descending order (the sweep over the frame groups runs backwards: see
// build_skew_plan)
int item_table(lutldpc_decoder *d, const std::vector<int> &blocks, const std::vector<double> &front, bool reverse, const int32_t **out, int *total) {
    int nb = 0;
    for (int b : blocks) nb += b;
    *total = nb;
    std::vector<int> fq(front.size());
    for (size_t r = 0; r < front.size(); r++) fq[r] = (int)(front[r] * 4096.0);
    fq.push_back(reverse ? 1 : 0);
    const auto key = std::make_pair(blocks, fq);
    auto it = d->item_tabs.find(key);
    if (it == d->item_tabs.end()) {
        std::vector<std::pair<double, std::pair<int, int>>> pos;
        pos.reserve((size_t)nb);
        // `front[r]` in [0,1): roles with long-running blocks are issued over [0, 1 - front) only, so that the
        // launch does not end on a tail of a few slow blocks (the next launch needs this one complete)
        for (size_t r = 0; r < blocks.size(); r++)
            for (int j = 0; j < blocks[r]; j++) pos.push_back({((double)j + 0.5) / (double)blocks[r] * (1.0 - front[r]), {(int)r, reverse ? blocks[r] - 1 - j : j}});
        std::stable_sort(pos.begin(), pos.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        std::vector<int32_t> h;
        h.reserve(2 * (size_t)nb);
        for (auto &q : pos) { h.push_back(q.second.first); h.push_back(q.second.second); }
        std::unique_ptr<DevBuf<int32_t>> buf(new DevBuf<int32_t>());
        HIP_TRY(buf->upload(h));
        it = d->item_tabs.emplace(key, std::move(buf)).first;
    }
    *out = it->second->p;
    return LUTLDPC_OK;
}

// LUTLDPC_VALIDATE: the roles of one fused launch against the sizes of everything they address
int validate_fused(const lutldpc_decoder *d, const FusedParams &FP, const std::vector<int> &blocks) {
    auto bad = [&](int r, const std::string &what) { return fail(LUTLDPC_ERR_STATE, "fused launch check failed, role " + std::to_string(r) + ": " + what); };
    if (FP.n_roles < 0 || FP.n_roles > kFusedMaxRoles || (size_t)FP.n_roles != blocks.size()) return bad(-1, "role count");
    const int groups = d->Bcap / d->tile();
    const size_t idx_n = d->fast_idx.size(), tab_n = d->d_tables.n, vfail_w = d->d_vfail.n / 4;
    for (int r = 0; r < FP.n_roles; r++) {
        const RoleParams &R = FP.role[r];
        if (R.G < 1 || R.g0 < 0 || R.g0 + R.G > groups) return bad(r, "frame groups outside the batch buffers");
        if (R.E != d->E || R.N != d->nvar) return bad(r, "E / N");
        if (R.n_nodes < 1 || R.nodes_per_wave < 1 || R.waves_per_group != (R.n_nodes + R.nodes_per_wave - 1) / R.nodes_per_wave) return bad(r, "waves per group");
        if (blocks[(size_t)r] != (R.waves_per_group * R.G + 3) / 4) return bad(r, "block count");
        if (R.vfail_stride_w != d->Bcap / 4 || R.vfail_off_w < 0 || (size_t)R.vfail_off_w + (size_t)kVfailSlots * (size_t)R.vfail_stride_w > vfail_w) return bad(r, "flag buffer");
        if (R.kind == 0) {
            if (R.deg < 2 || R.deg > kFusedCnDeg[d->fused_bucket_id]) return bad(r, "check degree outside the bucket");
            if (R.idx_off < 0 || (size_t)R.idx_off + (size_t)R.n_nodes * (size_t)R.deg > idx_n) return bad(r, "edge table");
            if (!is_pow2(R.nz) || R.nz > 64) return bad(r, "nz");
            if (R.first && (R.nidx_off < 0 || (size_t)R.nidx_off + (size_t)R.n_nodes * (size_t)R.deg > idx_n || R.check || R.chain.hard)) return bad(r, "node table of the first check pass");
            if (R.chain.on || R.chain.hard) {
                if (R.chain.idx_off < 0 || (size_t)R.chain.idx_off + 2 * (size_t)R.n_nodes > idx_n) return bad(r, "chain link table");
                if (R.chain.on && (R.chain.tab_off < 0 || R.chain.tab_len < 4 || R.chain.tab_len > 1024 || (size_t)R.chain.tab_off + (size_t)R.chain.tab_len > tab_n)) return bad(r, "chain table");
                if (R.chain.on && R.chain.check && (R.chain.vfail_off_w < 0 || (size_t)R.chain.vfail_off_w + (size_t)kVfailSlots * (size_t)R.vfail_stride_w > vfail_w)) return bad(r, "chain flag buffer");
            }
        } else {
            if (R.deg < 1 || R.deg > kFusedVnDeg[d->fused_bucket_id]) return bad(r, "variable degree outside the bucket");
            if (R.idx_off < 0 || (size_t)R.idx_off + 2 * (size_t)R.n_nodes > idx_n) return bad(r, "node table");
            const int nt = R.deg >= 3 ? R.deg - 1 : 1;
            for (int t = 0; t < nt; t++)
                if (R.tab_off[t] < 0 || R.tab_len[t] < 1 || R.tab_len[t] > kFastTableStride || (R.tab_off[t] & 3) || (size_t)R.tab_off[t] + (size_t)R.tab_len[t] > tab_n) return bad(r, "table " + std::to_string(t));
        }
    }
    return LUTLDPC_OK;
}

// the item table of one launch: per-wave work of a role ~ edges per wave, a variable-node edge costing about 3x a check
// edge (LUT look-ups); the slow roles keep clear of the end of the launch (item_table)
int plan_items(lutldpc_decoder *d, const FusedParams &FP, const std::vector<int> &blocks, bool reverse, const int32_t **items, int *nb) {
    std::vector<double> cost(blocks.size()), front(blocks.size());
    double cmax = 0;
    for (size_t r = 0; r < blocks.size(); r++) {
        const RoleParams &R = FP.role[r];
        cost[r] = (double)R.deg * R.nodes_per_wave * (R.kind ? 3.0 * R.deg / 4.0 : 1.0);
        cmax = std::max(cmax, cost[r]);
    }
    for (size_t r = 0; r < blocks.size(); r++) front[r] = d->tail_front * cost[r] / (cmax > 0 ? cmax : 1.0);
    return item_table(d, blocks, front, reverse, items, nb);
}

int launch_fused_slot(lutldpc_decoder *d, const lutldpc_decoder::SkewPlan &plan, const lutldpc_decoder::SkewSlot &sl, bool vn_check) {
    if (sl.nb == 0) return LUTLDPC_OK;
    Timed t(d, LUTLDPC_K_FUSED_PASS);
#define FUSED_ARGS d->stream, plan.d_roles.p + sl.role_off, sl.items, sl.nb, d->fused_prio, vn_check, d->d_msgs.p, d->d_cha_t.p, d->d_hard.p, \
                   reinterpret_cast<const uint32_t *>(d->d_state.p), reinterpret_cast<uint32_t *>(d->d_vfail.p), d->d_tables.p, d->d_fast_idx.p, d->d_msg0_t.p
    if (d->fused_bucket_id == 0) PACK_DISPATCH(d, (lutldpc::launch_fused<PK, 0>(FUSED_ARGS)));
    else if (d->fused_bucket_id == 1) PACK_DISPATCH(d, (lutldpc::launch_fused<PK, 1>(FUSED_ARGS)));
    else if (d->fused_bucket_id == 2) PACK_DISPATCH(d, (lutldpc::launch_fused<PK, 2>(FUSED_ARGS)));
    else PACK_DISPATCH(d, (lutldpc::launch_fused<PK, 3>(FUSED_ARGS)));
#undef FUSED_ARGS
    LAUNCH_CHECK();
    if (d->validate) {                               // attribute a device fault to this launch
        hipError_t e = hipStreamSynchronize(d->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) return fail(LUTLDPC_ERR_HIP, std::string("fused launch failed on the device: ") + hipGetErrorString(e));
    }
    return LUTLDPC_OK;
}

// Decided bits of the frames that left through the exit test, read off their frozen messages (hard_from_frozen_kernel) and, for
// the nodes updated inside the check pass, off the parity equations (chain_hard_kernel): groups g0 .. g0+G-1; ctl: a compaction
// check point's control words (the kernels return when it does not permute) or NULL at the end of the decode.
int launch_late_hard(lutldpc_decoder *d, bool skewed, int g0, int G, const int32_t *ctl) {
    bool chain_skip = false;
    if (!late_hard_active(d, skewed, &chain_skip) || G <= 0) return LUTLDPC_OK;
    const unsigned gx = ctl ? 1024u : 2048u;
    PACK_DISPATCH(d, launch_k(hard_from_frozen_kernel<PK>, dim3(std::min<unsigned>(gx, (unsigned)((d->nvar + 3) / 4)), (unsigned)G), dim3(256), 0, d->stream, d->d_msgs.p, d->d_hard.p,
                                        reinterpret_cast<const uint32_t *>(d->d_state.p), d->d_vn_ptr.p, chain_skip ? d->d_chain_internal.p : nullptr, d->nvar, d->E,
                                        d->Nq_Msg[0] / 2, g0, ctl));
    if (chain_skip)
        for (size_t i = 0; i < d->cclass.size(); i++) {
            if (d->chain_idx_off[i] < 0) continue;
            const int n = (int)d->cclass[i].nodes.size(), npw = d->npw_cn_class(i), runs = (n + npw - 1) / npw;
            PACK_DISPATCH(d, launch_k(chain_hard_kernel<PK>, dim3(std::min<unsigned>(512u, (unsigned)((runs + 3) / 4)), (unsigned)G), dim3(256), 0, d->stream, d->d_hard.p,
                                                reinterpret_cast<const uint32_t *>(d->d_state.p), d->d_fast_idx.p + d->cn_idx_off[i], d->d_fast_idx.p + d->chain_idx_off[i],
                                                d->d_edge_vn.p, n, d->cclass[i].deg, npw, d->nvar, g0, ctl));
        }
    LAUNCH_CHECK();
    return LUTLDPC_OK;
}

// kernels_compact.hpp: a check point of one half right after its exit test of iteration ii -- the plan kernel decides on the
// device whether permuting the slots (active frames first) pays; if not, the row kernels return at once
constexpr unsigned kPermuteBlocks = 512;      // two 16-wave blocks per CU
bool compaction_fits(const lutldpc_decoder *, int GH) { return GH <= kPermuteMaxGroups; }
// A check point costs three short launches per half (~15 us) whether it permutes or not: automatic mode switches compaction on
// only where one iteration of the batch lasts long enough to make that noise (estimated from its row traffic at 5.5 TB/s);
// LUTLDPC_COMPACT=1 / 0 forces it on / off.
bool compaction_on(const lutldpc_decoder *d, int G) {
    if (!compaction_fits(d, (G + 1) / 2) || G < 4) return false;
    if (d->use_compact >= 0) return d->use_compact != 0;
    const double est_iter_us = (4.0 * d->E + 3.0 * d->nvar) * kRowBytes * G / 5.5e6;
    return est_iter_us >= 400.0;
}
int launch_compaction(lutldpc_decoder *d, HalfRange h, int hf, int ii) {
    Timed t(d, LUTLDPC_K_LAYOUT);
    const int T = d->tile(), s0 = h.g0 * T, n = h.G * T;
    if (n <= 0) return LUTLDPC_OK;
    uint8_t *pending = d->d_vfail.p + (size_t)((ii + 1) & 1) * kVfailSlots * d->Bcap;      // flags already raised for the next test
    int32_t *ctl = d->d_ctl.p + 4 * hf;
    const bool late = late_hard_active(d, true, nullptr);
    // keep: the frames that left keep their frozen rows (moved behind the active ones), their decided bits are recovered once
    // at the end of the decode like without compaction; otherwise (LUTLDPC_COMPACT_KEEP=0) they are recovered at the check point
    // and the rows dropped
    const bool keep = late && d->compact_keep;
    launch_k(compact_decide_kernel, dim3(1), dim3(1024), 0, d->stream, d->d_state.p, s0, n, T, ctl, d->max_iters - 1 - ii, d->compact_margin,
                       d->compact_margin > 0 ? d->compact_min_share : 0.0f, keep ? 1 : 0);
    // (not keep) the decided bits of the frames that left since the last permutation, before their messages are dropped
    if (!keep) if (int rc = launch_late_hard(d, true, h.g0, h.G, ctl)) return rc;
    launch_k(compact_apply_kernel, dim3(1), dim3(1024), 0, d->stream, d->d_state.p, d->d_iters.p, d->d_frame_of.p, pending, d->Bcap, s0, n,
                       d->d_perm.p, d->d_tmp3.p + (size_t)3 * s0, ctl, (late && !keep) ? 1 : 0);
    // (the grid is fixed and small: an empty check point must cost microseconds)
    auto rows = [&](uint8_t *a, int na, uint8_t *b, int nb, int gather) {
        const unsigned blocks = std::min<unsigned>(kPermuteBlocks, (unsigned)((na + nb + kPermuteRows - 1) / kPermuteRows));
        PACK_DISPATCH(d, launch_k(permute_rows_kernel<PK>, dim3(blocks), dim3(1024), kPermuteLdsBytes, d->stream, a, na, b, nb, h.g0, h.G,
                                            d->d_perm.p, d->d_ctl.p + 4 * hf, gather));
    };
    rows(d->d_msgs.p, d->E, d->d_cha_t.p, d->nvar, keep ? 2 : 1);
    if (!keep) rows(d->d_hard.p, d->nvar, nullptr, 0, 0);        // (keep: no decided bit exists before the end of the decode)
    LAUNCH_CHECK();
    return LUTLDPC_OK;
}
// end of the decode: decided bits and iteration codes back into the caller's frame order
int launch_uncompaction(lutldpc_decoder *d, const HalfRange (&half)[2], int Bpad) {
    Timed t(d, LUTLDPC_K_LAYOUT);
    launch_k(invert_map_kernel, dim3((unsigned)((Bpad + 255) / 256)), dim3(256), 0, d->stream, d->d_frame_of.p, d->d_slot_of.p, 0, Bpad);
    for (int hf = 0; hf < 2; hf++) {
        if (half[hf].G <= 0) continue;
        PACK_DISPATCH(d, launch_k(permute_rows_kernel<PK>, dim3(std::min<unsigned>(kPermuteBlocks, (unsigned)((d->nvar + kPermuteRows - 1) / kPermuteRows))), dim3(1024),
                                            kPermuteLdsBytes, d->stream, d->d_hard.p, d->nvar, (uint8_t *)nullptr, 0,
                                            half[hf].g0, half[hf].G, d->d_slot_of.p, (const int32_t *)nullptr, 0));
    }
    launch_k(gather_i32_kernel, dim3((unsigned)((Bpad + 255) / 256)), dim3(256), 0, d->stream, d->d_iters.p, d->d_slot_of.p, d->d_iters_tmp.p, 0, Bpad);
    HIP_TRY(hipMemcpyAsync(d->d_iters.p, d->d_iters_tmp.p, sizeof(int32_t) * (size_t)Bpad, hipMemcpyDeviceToDevice, d->stream));
    LAUNCH_CHECK();
    return LUTLDPC_OK;
}

// Build (once per shape) the launch plan of the message-passing iterations of both halves: slot s pairs pass s of half A
// with pass s-1 of half B, a pass being CN(ii) for even and VN(ii) for odd numbers.  Every role is checked against the
// sizes of what it addresses before the plan is accepted (validate_fused), the roles then move to device memory once.
int build_skew_plan(lutldpc_decoder *d, int G, lutldpc_decoder::SkewPlan &plan) {
    const int I = d->max_iters, n_ops = 2 * I - 1;
    const HalfRange half[2] = {{0, (G + 1) / 2}, {(G + 1) / 2, G - (G + 1) / 2}};
    const int psc = d->psc ? 1 : 0;
    int rc;
    for (int slot = 0; slot <= n_ops; slot++) {
        FusedParams FP{};
        std::vector<int> blocks;
        lutldpc_decoder::SkewSlot sl;
        for (int hf = 0; hf < 2; hf++) {
            const int op = slot - hf;                 // B lags by one pass
            if (op < 0 || op >= n_ops) continue;
            const int ii = op / 2;
            if ((op & 1) == 0) {                      // CN(ii)
                const int check = (psc && ii > 0) ? 1 : 0;
                add_cn_roles(d, FP, blocks, half[hf], ii, check);
                if (check) { sl.state_half = hf; sl.state_ii = ii; }
            } else {                                  // VN(ii)
                add_vn_roles(d, FP, blocks, half[hf], ii, psc, (psc && !late_hard_active(d, true, nullptr)) ? 1 : 0);
            }
        }
        // roles without work (an empty half when G == 1 never gets here; a degree class emptied by chain fusion does)
        FusedParams FQ{};
        std::vector<int> bq;
        for (int r = 0; r < FP.n_roles; r++) if (blocks[(size_t)r] > 0) { FQ.role[FQ.n_roles++] = FP.role[r]; bq.push_back(blocks[(size_t)r]); }
        if ((rc = validate_fused(d, FQ, bq))) return rc;
        // every other launch sweeps the frame groups backwards: what a half wrote last in one launch (still in the 256 MB
        // Infinity Cache) is what the next launch reads first
        if ((rc = plan_items(d, FQ, bq, d->sweep_reverse && (slot & 1), &sl.items, &sl.nb))) return rc;
        sl.n_roles = FQ.n_roles; sl.role_off = plan.h_roles.size();
        plan.h_roles.insert(plan.h_roles.end(), FQ.role, FQ.role + FQ.n_roles);
        plan.slots.push_back(sl);
    }
    HIP_TRY(plan.d_roles.upload(plan.h_roles));
    return LUTLDPC_OK;
}

// the message-passing iterations of decode_tiles for both halves
int iterate_skewed(lutldpc_decoder *d, int B, int Bpad, int G) {
    const int I = d->max_iters;
    const HalfRange half[2] = {{0, (G + 1) / 2}, {(G + 1) / 2, G - (G + 1) / 2}};
    const int psc = d->psc ? 1 : 0;
    int rc;
    auto &pp = d->skew_plans[{G, psc, I}];
    if (!pp) {
        std::unique_ptr<lutldpc_decoder::SkewPlan> np(new lutldpc_decoder::SkewPlan());
        if ((rc = build_skew_plan(d, G, *np))) { np->d_roles.release(); d->skew_plans.erase({G, psc, I}); return rc; }
        pp = std::move(np);
    }
    const lutldpc_decoder::SkewPlan &plan = *pp;
    // compaction of the surviving frames: check points every `every` iterations (a check point costs three short launches
    // per half: keep that below a few per cent of an iteration, whose duration is estimated from its row traffic)
    const bool compact = psc && compaction_on(d, G);
    const int every = d->compact_every > 0 ? d->compact_every : 2;
    if (compact) {
        Timed t(d, LUTLDPC_K_LAYOUT);
        launch_k(compact_init_kernel, dim3((unsigned)((Bpad + 255) / 256)), dim3(256), 0, d->stream, d->d_frame_of.p, Bpad, d->d_ctl.p, half[0].G, half[1].G);
        LAUNCH_CHECK();
    }
    for (const auto &sl : plan.slots) {
        if ((rc = launch_fused_slot(d, plan, sl, psc != 0))) return rc;
        if (sl.state_half >= 0) {                     // :327-329 returns (ii-1)+1
            const int f0 = half[sl.state_half].g0 * d->tile(), f1 = f0 + half[sl.state_half].G * d->tile();
            if ((rc = launch_state(d, B, Bpad, 2, sl.state_ii, f0, f1, sl.state_ii & 1))) return rc;
            if (compact && sl.state_ii >= d->compact_first && sl.state_ii < I - 2 && (sl.state_ii - d->compact_first) % every == 0 &&
                (rc = launch_compaction(d, half[sl.state_half], sl.state_half, sl.state_ii))) return rc;
        }
    }
    return LUTLDPC_OK;
}

// ----------------------------------------------------------------------------- LDS-resident decoder (jit_resident.hpp)
constexpr int kLdsPerCu = 160 * 1024, kResidentCus = 256;

ResidentSpec resident_spec(const lutldpc_decoder *d, int S, int NT) {
    ResidentSpec R;
    R.pack = d->pack; R.N = d->nvar; R.E = d->E; R.S = S; R.NT = NT; R.I = d->max_iters_created; R.nq_cha = d->Nq_Cha; R.min_lut = d->min_lut;
    R.nq_msg = d->Nq_Msg; R.iter_set = d->iter_set; R.U = d->resident_U; R.xcd = d->resident_xcd; R.waves_eu = d->resident_waves_eu;
    {   // wave-reduced exit-test flags cost registers in the item bodies: +7 % on (3,6) N=10000 as shipped, but with the wide trees of
        // N=500 (degree 17: 168 registers, one wave per SIMD less) 13.5 -> 10.2 M codewords/s -- only where the trees are small
        int max_vn = 0;
        for (auto &c : d->vclass) max_vn = std::max(max_vn, c.deg);
        R.flag_reduce = d->resident_flag_reduce >= 0 ? d->resident_flag_reduce : (max_vn <= 8 ? 1 : 0);
        // check items keep their LDS addresses in registers when that is few registers and the trees leave room for them
        int cn_regs = 0, max_cn = 0;
        for (auto &c : d->cclass) { cn_regs += (int)((S * (long long)c.nodes.size() + NT - 1) / NT + 1) * c.deg; max_cn = std::max(max_cn, c.deg); }
        // (... or the wide trees have cost the occupancy already: N=500 runs two waves per SIMD with 159 registers, +3 % with them;
        // (6,32) N=2048 would fall from three workgroups per compute unit to two: 25.6 -> 23.5 M, off)
        R.cn_persistent = d->resident_cn_persistent >= 0 ? d->resident_cn_persistent
                          : (d->min_lut && max_cn <= 16 && ((max_vn <= 4 && cn_regs <= 40) || (max_vn > 12 && cn_regs <= 96)) ? 1 : 0);      // (measured: (3,6) N=10000 1.82 -> 1.92 M codewords/s fixed work, 3.65 -> 3.93 M as shipped)
    }
    for (size_t i = 0; i < d->vclass.size(); i++) R.vcls.push_back({d->vclass[i].deg, (int)d->vclass[i].nodes.size(), d->vn_tidx_off[i], 0});
    for (size_t i = 0; i < d->cclass.size(); i++) R.ccls.push_back({d->cclass[i].deg, (int)d->cclass[i].nodes.size(), d->cn_tidx_off[i], d->cn_tnidx_off[i]});
    const size_t ns = d->var_plan.size();
    R.var_prog.assign(ns, {}); R.dec_prog.assign(ns, {}); R.chk_prog.assign(ns, {});
    R.var_tab.assign(ns, {}); R.dec_tab.assign(ns, {}); R.chk_tab.assign(ns, {});
    for (size_t s = 0; s < ns; s++) {
        auto fill = [&](const PassPlan &plan, const std::vector<Program> &progs, const std::vector<std::pair<int, int>> &tabs, std::vector<const Program *> &pp,
                        std::vector<std::pair<int, int>> &tt) {
            if (!plan.valid) return;
            for (size_t c = 0; c < progs.size(); c++) { pp.push_back(&progs[c]); tt.push_back(tabs[c]); }
        };
        fill(d->var_plan[s], d->var_prog_c[s], d->var_tab_c[s], R.var_prog[s], R.var_tab[s]);
        fill(d->dec_plan[s], d->dec_prog_c[s], d->dec_tab_c[s], R.dec_prog[s], R.dec_tab[s]);
        if (!d->min_lut && d->chk_plan[s].valid)
            for (size_t c = 0; c < d->chk_prog_c[s].size(); c++) {      // over full labels where that form exists (one instruction per look-up)
                const bool full = c < d->chk_tab_cf[s].size() && d->chk_tab_cf[s][c].second > 0;
                R.chk_prog[s].push_back(full ? &d->chk_prog_cf[s][c] : &d->chk_prog_c[s][c]);
                R.chk_tab[s].push_back(full ? d->chk_tab_cf[s][c] : d->chk_tab_c[s][c]);
            }
    }
    return R;
}

// can this code be decoded out of LDS at all (one set per workgroup)?
bool resident_eligible(const lutldpc_decoder *d) {
    if (!d->use_resident || !d->use_jit || !d->use_fast || d->device < 0) return false;
    if (d->min_lut) for (int nq : d->Nq_Msg) if (!is_pow2(nq / 2)) return false;
    for (auto &c : d->cclass) if (c.deg < 2 || c.deg > 64) return false;
    // wide CHKTREE checks (a 31-leaf tree: 184 look-ups per frame and check, inputs / outputs / edge ids of 32 edges in registers) run
    // faster through the streaming pass kernels: (6,32) N=2048 with min_lut = false 7.4 M codewords/s against 5.6 M out of LDS
    if (!d->min_lut && d->use_resident < 2) for (auto &c : d->cclass) if (c.deg > 16) return false;      // (LUTLDPC_RESIDENT=2 forces it)
    for (auto &c : d->vclass) if (c.deg > 24) return false;
    if (d->vclass.size() > 12 || d->cclass.size() > 12) return false;
    const ResidentSpec R = resident_spec(d, 1, 1024);
    return resident_lds_bytes(R) <= kLdsPerCu - 2048;
}

// Sets per workgroup (S) and workgroup size (NT) for a batch of G frame groups.  Rules read off tools/resident_probe.py sweeps on
// MI355X (profiles/r03_resident_geometry_sweep.txt):
//   * two or three workgroups per compute unit beat one large one -- their check and variable phases interleave on the vector
//     ALUs and the LDS: (6,32) N=2048 23.6 M codewords/s at S = 1 / 512 threads (three workgroups of 51 KB) against 19.3 M at
//     S = 2 / 1024 -- so S is the largest value that leaves room for two workgroups (<= 78 KB of LDS each) ...
//   * ... and gives a thread about eight variable-node items: every pass of a workgroup costs two barriers and a table staging
//     whatever its size, and the items of a heavy degree class spread evenly only when there are many (N=500: 13.3 M at
//     S = 8 / 512, 9.4 M at S = 2 / 256, 6.2 M at S = 2 / 1024);
//   * a code that fills the LDS with a single set ((3,6) N=10000: 120 KB) runs one workgroup of 1024 threads;
//   * a batch too small to give every compute unit its workgroups takes a smaller S.
// LUTLDPC_RESIDENT_S / LUTLDPC_RESIDENT_NT override.
bool resident_pick(const lutldpc_decoder *d, int G, int &S_out, int &NT_out, int &lds_out) {
    const long long sets = 64ll * G;
    auto lds_of = [&](int S, int NT) { return resident_lds_bytes(resident_spec(d, S, NT)); };
    auto items_of = [&](int S, int NT) { return (int)((S * (long long)d->nvar + NT - 1) / NT) + (int)d->vclass.size(); };
    const int budget = kLdsPerCu - 2048;
    if (lds_of(1, 512) > budget) return false;
    int S = 1, NT = 512;
    if (d->resident_force_S || d->resident_force_NT) {
        S = d->resident_force_S ? d->resident_force_S : 1;
        NT = d->resident_force_NT ? d->resident_force_NT : 512;
    } else if (lds_of(1, 512) > 78 * 1024) {
        NT = 1024;                                              // one workgroup per compute unit: all the waves it can hold
    } else {
        while (S < 64 && lds_of(S + 1, NT) <= 78 * 1024 && (S + 1) * (long long)d->nvar <= 8ll * NT + NT / 2) S++;
        while (S > 1 && (sets + S - 1) / S < (long long)kResidentCus) S--;        // small batch: at least one workgroup per compute unit
    }
    while (S > 1 && (lds_of(S, NT) > budget || items_of(S, NT) > 46)) S--;
    if (lds_of(S, NT) > budget || items_of(S, NT) > 46) {
        if (NT < 1024 && items_of(S, 1024) <= 46 && lds_of(S, 1024) <= budget) NT = 1024; else return false;
    }
    S_out = S; NT_out = NT; lds_out = lds_of(S, NT);
    return true;
}

bool resident_active(const lutldpc_decoder *d) { return d->resident_ok && d->use_resident; }

// Placement search for the row buffers of a large batch (see lutldpc_decoder::place_candidates).  Candidate 0 is what ensure_batch
// has just allocated; every further candidate is a fresh set of the same sizes, ALL kept alive until the choice is made (a freed
// set would be handed out again).  The probe is the real thing on zeroed rows: frame states, then three iterations of the fused
// pipeline (six launches), timed with events on the decoder's stream; the second run counts.  Only for the skewed streaming path
// and batches whose rows exceed 1 GiB -- below that the launches are not bound by HBM.  Allocation failures end the search quietly.
int place_rows(lutldpc_decoder *d, int Bpad) {
    const int G = Bpad / d->tile();
    const size_t total = d->d_msgs.bytes() + d->d_cha_t.bytes() + d->d_msg0_t.bytes() + d->d_hard.bytes();
    d->place_info = "null";
    if (d->place_candidates < 2 || d->device < 0 || !d->skew || !d->skew_ok || resident_active(d) || d->trace.level > 1 || total < ((size_t)1 << 30) || d->max_iters_created < 2) return LUTLDPC_OK;
    struct RowSet { DevBuf<uint8_t> msgs, cha, msg0, hard; float ms = 0.f; int id = 0; void release() { msgs.release(); cha.release(); msg0.release(); hard.release(); } };
    std::vector<std::unique_ptr<RowSet>> parked;      // the candidates tried so far, except the one the decoder holds right now
    const size_t n_msgs = d->d_msgs.n, n_node = d->d_cha_t.n;
    auto swap_in = [&](RowSet &r) { std::swap(d->d_msgs, r.msgs); std::swap(d->d_cha_t, r.cha); std::swap(d->d_msg0_t, r.msg0); std::swap(d->d_hard, r.hard); };
    const int I0 = d->max_iters; const bool psc0 = d->psc, pisc0 = d->pisc; const int prof0 = d->profiling;
    d->max_iters = std::min(3, d->max_iters_created); d->psc = d->pisc = false; d->profiling = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = LUTLDPC_OK;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); rc = -1; }
    std::vector<float> times;                          // probe time of every candidate, in the order tried
    size_t parked_bytes = 0, mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) { (void)hipGetLastError(); mem_free = 0; }
    const size_t mem_budget = mem_free / 3;          // (a third: two lanes of a device may search at the same time)
    float cur_ms = 0.f; int cur_id = 0;                // the candidate the decoder holds
    for (int k = 0; k < d->place_candidates && rc == LUTLDPC_OK; k++) {
        if (k > 0) {
            std::unique_ptr<RowSet> r(new RowSet());
            if (parked_bytes + 2 * total > mem_budget) break;      // keep the search within a third of what was free when it started
            if (r->msgs.alloc(n_msgs) != hipSuccess || r->cha.alloc(n_node) != hipSuccess || r->msg0.alloc(n_node) != hipSuccess || r->hard.alloc(n_node) != hipSuccess ||
                hipMemsetAsync(r->msgs.p, 0, r->msgs.bytes(), d->stream) != hipSuccess || hipMemsetAsync(r->cha.p, 0, r->cha.bytes(), d->stream) != hipSuccess ||
                hipMemsetAsync(r->msg0.p, 0, r->msg0.bytes(), d->stream) != hipSuccess || hipMemsetAsync(r->hard.p, 0, r->hard.bytes(), d->stream) != hipSuccess) {
                (void)hipGetLastError(); r->release(); break;        // out of memory: choose among what was tried
            }
            swap_in(*r);                               // the decoder works on candidate k, r holds candidate cur_id
            r->ms = cur_ms; r->id = cur_id;
            parked.push_back(std::move(r));
            parked_bytes += total;
            cur_id = k;
        }
        for (int rep = 0; rep < 2 && rc == LUTLDPC_OK; rep++) {
            if ((rc = launch_state(d, Bpad, Bpad, 0, 0))) break;
            if (hipEventRecord(e0, d->stream) != hipSuccess) { rc = -1; break; }
            if ((rc = iterate_skewed(d, Bpad, Bpad, G))) break;
            if (hipEventRecord(e1, d->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) { rc = -1; break; }
            (void)hipEventElapsedTime(&cur_ms, e0, e1);
        }
        times.push_back(cur_ms);
        // the levels are discrete (on DVB-S2: 5.97 / 6.25 / 6.45-6.6 ms for the probe, the top one in one allocation out of eight):
        // stop as soon as one candidate stands 6.5 % clear of the slowest seen
        if (times.size() >= 4) {
            const float lo = *std::min_element(times.begin(), times.end()), hi = *std::max_element(times.begin(), times.end());
            if (lo <= 0.935f * hi) break;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    d->max_iters = I0; d->psc = psc0; d->pisc = pisc0; d->profiling = prof0;
    if (rc != LUTLDPC_OK) {        // a probe failed: keep what the decoder holds, report nothing (the decode that follows surfaces a real error)
        (void)hipGetLastError();
        for (auto &c : parked) c->release();
        return LUTLDPC_OK;
    }
    for (auto &c : parked)
        if (c->ms < cur_ms) { swap_in(*c); std::swap(c->ms, cur_ms); std::swap(c->id, cur_id); }      // the decoder ends up with the fastest set
    std::ostringstream o;
    o << "{\"candidates\":" << times.size() << ",\"chosen\":" << cur_id << ",\"probe_ms\":[";
    for (size_t k = 0; k < times.size(); k++) o << (k ? "," : "") << times[k];
    o << "]}";
    d->place_info = o.str();
    auto &cand = parked;
    for (auto &c : cand) { c->msgs.release(); c->cha.release(); c->msg0.release(); c->hard.release(); }
    // the probes ran on zeroed rows and left their messages behind: defined content again (see ensure_batch)
    HIP_TRY(hipMemsetAsync(d->d_msgs.p, 0, d->d_msgs.bytes(), d->stream));
    HIP_TRY(hipMemsetAsync(d->d_hard.p, 0, d->d_hard.bytes(), d->stream));
    d->drop_graphs();
    make_describe(d);
    if (getenv("LUTLDPC_DEBUG_ADDR")) fprintf(stderr, "lutldpc placement: %s -> msgs %p\n", d->place_info.c_str(), (void *)d->d_msgs.p);
    return LUTLDPC_OK;
}

int resident_plan_for(lutldpc_decoder *d, int G, lutldpc_decoder::ResidentPlan **out) {
    auto it = d->resident_plans.find(G);
    if (it == d->resident_plans.end()) {
        lutldpc_decoder::ResidentPlan pl;
        if (!resident_pick(d, G, pl.S, pl.NT, pl.lds)) return fail(LUTLDPC_ERR_STATE, "resident decoder: no configuration fits");
        // equal (S, NT) of another batch size: the same kernel
        for (auto &kv : d->resident_plans) if (kv.second.S == pl.S && kv.second.NT == pl.NT) pl.k = kv.second.k;
        if (!pl.k) {
            std::string src, err;
            if (!jit_resident_source(resident_spec(d, pl.S, pl.NT), src, err)) return fail(LUTLDPC_ERR_UNSUPPORTED, "resident decoder: " + err);
            JitRegistry &reg = jit_registry();
            std::lock_guard<std::mutex> lock(reg.mu);
            const std::string key = std::to_string(d->device) + "\n" + src;
            auto kt = reg.by_src.find(key);
            if (kt == reg.by_src.end()) {
                if (reg.by_src.size() >= kJitRegistryMax) return fail(LUTLDPC_ERR_STATE, "generated-kernel registry full");
                std::vector<char> code;
                JitKernel k;
                std::string log;
                if (!jit_compile(src, code, log) || !jit_load(code, k, log)) { d->resident_log = log; reg.by_src[key] = JitKernel(); return fail(LUTLDPC_ERR_HIP, "resident decoder: hiprtc / module load failed: " + log.substr(0, 2000)); }
                kt = reg.by_src.emplace(key, k).first;
            }
            if (!kt->second.ok()) return fail(LUTLDPC_ERR_HIP, "resident decoder: kernel unavailable (earlier compile failure)");
            pl.k = &kt->second;
        }
        it = d->resident_plans.emplace(G, pl).first;
    }
    *out = &it->second;
    return LUTLDPC_OK;
}

int launch_resident(lutldpc_decoder *d, int G, int B) {
    lutldpc_decoder::ResidentPlan *pl = nullptr;
    if (int rc = resident_plan_for(d, G, &pl)) return rc;
    Timed t(d, LUTLDPC_K_RESIDENT);
    ResidentArgs A{};
    A.cha = d->d_cha_t.p; A.msg0 = d->d_msg0_t.p; A.hard = d->d_hard.p; A.state = d->d_state.p; A.iters = d->d_iters.p;
    A.tables = d->d_tables.p; A.idx = d->d_fast_idx.p; A.n_sets = 64 * G; A.max_iters = d->max_iters; A.psc = d->psc; A.pisc = d->pisc;
    A.B = B; A.fm_cha = d->fm_cha; A.fm_msg0 = d->fm_msg0; A.fm_bits = d->fm_bits; A.lim_cha = d->Nq_Cha - 1; A.lim_msg = d->Nq_Msg[0] - 1;
    void *args[] = {&A};
    const unsigned blocks = (unsigned)((64 * G + pl->S - 1) / pl->S);
    HIP_TRY(hipModuleLaunchKernel(pl->k->fn, blocks, 1, 1, (unsigned)pl->NT, 1, 1, 0, d->stream, args, nullptr));
    LAUNCH_CHECK();
    return LUTLDPC_OK;
}

// one message dump of the trace: the E edge rows of all frames, frame-major, to the next slot of the host buffer (synchronous)
int trace_dump(lutldpc_decoder *d) {
    lutldpc_decoder::Trace &T = d->trace;
    const size_t one = (size_t)T.B * (size_t)d->E;
    if ((size_t)(T.n + 1) * one > T.cap) return fail(LUTLDPC_ERR_ARG, "trace buffer too small");
    const int G = d->bpad(T.B) / d->tile();
    HIP_TRY(d->d_trace.alloc(one));
    if (int rc = launch_transpose_out(d, d->d_msgs.p, d->d_trace.p, T.B, G, d->E)) return rc;
    LAUNCH_CHECK();
    HIP_TRY(hipMemcpyAsync(T.host + (size_t)T.n * one, d->d_trace.p, one, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    T.n++;
    return LUTLDPC_OK;
}

// Core: decode the B frames whose labels are already in tile layout (d_cha_t / d_msg0_t).
// Leaves the decided bits in d_hard (tile layout) and the iteration codes in d_iters.
int decode_tiles_launch(lutldpc_decoder *d, int B) {
    int rc;
    const int Bpad = d->bpad(B), G = Bpad / d->tile();
    const int N = d->nvar, E = d->E, I = d->max_iters;
    const int last_set = d->iter_set[(size_t)(I - 1)];
    if (!d->dec_plan[(size_t)last_set].valid)
        return fail(LUTLDPC_ERR_STATE, "the tree set of iteration max_iters-1 is not a decision tree set");
    if ((rc = launch_state(d, B, Bpad, 0, 0))) return rc;
    const bool tracing = d->trace.level > 1;
    if (resident_active(d) && !tracing) {         // the whole of lut_decode in one launch, messages in LDS (jit_resident.hpp)
        if ((rc = launch_resident(d, G, B))) return rc;
        if (d->profiling && d->ev_live.size() > 8192) prof_fold(d);
        return LUTLDPC_OK;
    }
    if (d->pisc) {   // :275-279
        if (is_pow2(d->Nq_Cha / 2)) {
            if ((rc = launch_syndrome_of_labels(d, G))) return rc;
        } else {
            // the decided bit `label < Nq_Cha/2` is the inverted sign BIT of the label only when Nq_Cha/2 is a power of two: any other
            // channel alphabet goes through decided-bit rows (SWAR compare) and the parity pass over them
            {
                Timed t(d, LUTLDPC_K_LAYOUT);
                const size_t n_words = (size_t)G * (size_t)N * kRowBytes / 4;
                PACK_DISPATCH(d, launch_k(hard_from_labels_kernel<PK>, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, d->stream, d->d_cha_t.p, d->d_hard.p, n_words, d->Nq_Cha / 2));
                LAUNCH_CHECK();
            }
            if ((rc = launch_syndrome(d, G))) return rc;
        }
        if ((rc = launch_state(d, B, Bpad, 1, 0))) return rc;
        {   // decided bits of the frames that passed = signs of their channel labels (:275); groups without such a frame return at once
            Timed t(d, LUTLDPC_K_LAYOUT);
            PACK_DISPATCH(d, launch_k(hard_from_labels_masked_kernel<PK>, dim3(std::min<unsigned>(1024u, (unsigned)((N + 3) / 4)), (unsigned)G), dim3(256), 0, d->stream,
                                                d->d_cha_t.p, d->d_hard.p, reinterpret_cast<const uint32_t *>(d->d_state.p), N, d->Nq_Cha / 2, 0));
            LAUNCH_CHECK();
        }
    }
    const bool skewed = d->skew && d->skew_ok && !tracing;      // (a single frame group runs the same launches with an empty second half)
    if (!(skewed && d->first_from_nodes)) {   // :284-289 (the fused pipeline's first check pass reads the initial-message rows itself)
        Timed t(d, LUTLDPC_K_LAYOUT);
        launch_k(init_edges_kernel, dim3((unsigned)((N + 3) / 4), (unsigned)G), dim3(256), 0, d->stream, d->d_msg0_t.p, d->d_msgs.p, d->d_vn_ptr.p, N, E);
        LAUNCH_CHECK();
    }
    if (tracing && (rc = trace_dump(d))) return rc;                              // :292-298
    if (skewed && (rc = iterate_skewed(d, B, Bpad, G))) return rc;
    for (int ii = 0; ii < I && !skewed; ii++) {   // :301-338
        const int set = d->iter_set[(size_t)ii];
        const int nz_in = d->Nq_Msg[(size_t)ii] / 2;
        const int chk_check = (d->psc && ii > 0) ? 1 : 0;    // finishes the test started by VN pass ii-1
        if (d->min_lut) rc = launch_cn_minsum(d, G, nz_in, chk_check);
        else rc = launch_tree_pass<TT_CHK>(d, d->chk_plan[(size_t)set], nullptr, d->chk_jit.empty() ? nullptr : &d->chk_jit[(size_t)set], G, nz_in, chk_check, 0, LUTLDPC_K_CN_PASS,
                                           (size_t)set < d->chk_full_tab.size() ? &d->chk_full_tab[(size_t)set] : nullptr);
        if (rc) return rc;
        if (chk_check && (rc = launch_state(d, B, Bpad, 2, ii))) return rc;   // :327-329 returns (ii-1)+1
        if (d->trace.level > 2 && (rc = trace_dump(d))) return rc;             // :311-317
        if (ii != I - 1) {
            const int nz_out = d->Nq_Msg[(size_t)(ii + 1)] / 2;
            rc = launch_tree_pass<TT_VAR>(d, d->var_plan[(size_t)set], &d->var_fast[(size_t)set], d->var_jit.empty() ? nullptr : &d->var_jit[(size_t)set], G, nz_out, d->psc ? 1 : 0,
                                          (d->psc && !late_hard_active(d, false, nullptr)) ? 1 : 0, LUTLDPC_K_VN_PASS);
            if (rc) return rc;
        }
        if (tracing && (rc = trace_dump(d))) return rc;                         // :331-337 (printed after the last iteration too)
    }
    {   // decided bits of the frames that left through the exit test, from their frozen messages (see late_hard_active)
        Timed t(d, LUTLDPC_K_LAYOUT);
        if ((rc = launch_late_hard(d, skewed, 0, G, nullptr))) return rc;
        if (skewed && d->psc && d->pisc && compaction_on(d, G) && late_hard_active(d, true, nullptr) && d->compact_keep) {
            // frames that passed the test on the channel decisions may have been moved by a permutation: their decided-bit rows
            // did not travel (no other decided bit exists during the iterations), their channel rows did -- write the bits again
            PACK_DISPATCH(d, launch_k(hard_from_labels_masked_kernel<PK>, dim3(std::min<unsigned>(1024u, (unsigned)((N + 3) / 4)), (unsigned)G), dim3(256), 0, d->stream,
                                                d->d_cha_t.p, d->d_hard.p, reinterpret_cast<const uint32_t *>(d->d_state.p), N, d->Nq_Cha / 2, 0));
            LAUNCH_CHECK();
        }
    }
    // :340-349
    if ((rc = launch_tree_pass<TT_DEC>(d, d->dec_plan[(size_t)last_set], &d->dec_fast[(size_t)last_set], d->dec_jit.empty() ? nullptr : &d->dec_jit[(size_t)last_set], G, 0, 0, 0, LUTLDPC_K_DECISION))) return rc;
    const int fsel = skewed ? (I & 1) : 0;            // the flag buffer no pass of the skewed pipeline has written since its last test
    if ((rc = launch_syndrome(d, G, fsel))) return rc;
    if ((rc = launch_state(d, B, Bpad, 3, I, 0, -1, fsel))) return rc;
    if (skewed && d->psc && compaction_on(d, G)) {
        const HalfRange half[2] = {{0, (G + 1) / 2}, {(G + 1) / 2, G - (G + 1) / 2}};
        if ((rc = launch_uncompaction(d, half, Bpad))) return rc;
    }
    if (d->profiling && d->ev_live.size() > 8192) prof_fold(d);
    return LUTLDPC_OK;
}

// A decode is 100-300 short launches whose arguments depend only on (B, exit conditions): from the
// second call with the same key on, the sequence is replayed as ONE hipGraph launch (the first call runs
// plainly and fills the item-table cache, whose uploads may not happen inside a capture).  Short codes
// are launch-bound, for them this is worth ~20 %.  Off while kernel events are being recorded.
int decode_tiles(lutldpc_decoder *d, int B) {
    if (int rc = check_batch_buffers(d, d->bpad(B))) return rc;
    if (resident_active(d)) {                     // generate / compile / load outside any stream capture
        lutldpc_decoder::ResidentPlan *pl = nullptr;
        if (int rc = resident_plan_for(d, d->bpad(B) / d->tile(), &pl)) return rc;
    }
    // (the resident decoder is ONE launch plus the state kernel: nothing to gain from a graph, and the frame-major pointers of the
    // caller would be frozen into it)
    if (!d->use_graph || d->profiling || d->trace.level > 1 || resident_active(d)) return decode_tiles_launch(d, B);
    const std::array<int, 4> key = {B, d->psc, d->pisc, d->max_iters};
    if (d->graphs.size() > 32 && !d->graphs.count(key)) d->drop_graphs();      // callers with ever-changing batch sizes: bound the cache
    auto &slot = d->graphs[key];
    if (slot.exec) {
        HIP_TRY(hipGraphLaunch(slot.exec, d->stream));
        return LUTLDPC_OK;
    }
    if (slot.seen++ == 0) return decode_tiles_launch(d, B);
    HIP_TRY(hipStreamBeginCapture(d->stream, hipStreamCaptureModeThreadLocal));
    const int rc = decode_tiles_launch(d, B);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(d->stream, &g);
    if (rc || e != hipSuccess || !g) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        d->use_graph = 0;                             // capture not possible here: plain launches from now on
        return rc ? rc : decode_tiles_launch(d, B);
    }
    const hipError_t ei = hipGraphInstantiate(&slot.exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ei != hipSuccess) { slot.exec = nullptr; d->use_graph = 0; (void)hipGetLastError(); return decode_tiles_launch(d, B); }
    HIP_TRY(hipGraphLaunch(slot.exec, d->stream));
    return LUTLDPC_OK;
}

// The batched lut_decode (src/LDPC_Code_LUT.cpp:259-353) on device-resident frame-major labels.
int decode_device(lutldpc_decoder *d, const uint8_t *d_cha, const uint8_t *d_msg0, int B, uint8_t *d_out_bits, int32_t *d_out_iters) {
    if (d->device < 0) return fail(LUTLDPC_ERR_STATE, "decoder was created without a device (host-only handle)");
    if (B <= 0) return fail(LUTLDPC_ERR_ARG, "B must be positive");
    HIP_TRY(hipSetDevice(d->device));
    int rc = ensure_batch(d, B);
    if (rc) return rc;
    const int Bpad = d->bpad(B), G = Bpad / d->tile();
    // the LDS-resident decoder reads the frame-major labels and writes the frame-major bits itself
    const bool direct = resident_active(d) && d->resident_fm && d->trace.level <= 1;
    if (!direct) {
        Timed t(d, LUTLDPC_K_LAYOUT);
        if ((rc = launch_transpose_in(d, d_cha, d->d_cha_t.p, B, G, d->Nq_Cha))) return rc;
        if ((rc = launch_transpose_in(d, d_msg0, d->d_msg0_t.p, B, G, d->Nq_Msg[0]))) return rc;
        LAUNCH_CHECK();
    }
    if (direct) { d->fm_cha = d_cha; d->fm_msg0 = d_msg0; d->fm_bits = d_out_bits; }
    rc = decode_tiles(d, B);
    d->fm_cha = d->fm_msg0 = nullptr; d->fm_bits = nullptr;
    if (rc) return rc;
    {
        Timed t(d, LUTLDPC_K_LAYOUT);
        if (!direct && (rc = launch_transpose_out(d, d->d_hard.p, d_out_bits, B, G))) return rc;
        HIP_TRY(hipMemcpyAsync(d_out_iters, d->d_iters.p, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToDevice, d->stream));
    }
    return LUTLDPC_OK;
}

int fill_cells(const lutldpc_channel_cells *c, const lutldpc_decoder *d, ChannelCells &C) {
    if (!c || !c->thr || !c->cha_label || !c->msg_label || !c->slicer_neg || !c->cha_label_mirror || !c->msg_label_mirror)
        return fail(LUTLDPC_ERR_ARG, "channel cells: NULL member");
    if (c->n_cells < 1 || c->n_cells > kMaxCells) return fail(LUTLDPC_ERR_ARG, "channel cells: n_cells outside [1,72]");
    std::memset(&C, 0, sizeof(C));
    C.n_cells = c->n_cells;
    for (int j = 0; j < c->n_cells; j++) {
        if (j < c->n_cells - 1) { C.thr[j] = c->thr[j]; if (j && c->thr[j] < c->thr[j - 1]) return fail(LUTLDPC_ERR_ARG, "channel cells: thresholds must ascend"); }
        if (c->cha_label[j] >= d->Nq_Cha || c->cha_label_mirror[j] >= d->Nq_Cha || c->msg_label[j] >= d->Nq_Msg[0] || c->msg_label_mirror[j] >= d->Nq_Msg[0])
            return fail(LUTLDPC_ERR_ARG, "channel cells: label outside its alphabet");
        C.cha[j] = c->cha_label[j]; C.msg[j] = c->msg_label[j]; C.neg[j] = c->slicer_neg[j] ? 1 : 0;
        C.cha_m[j] = c->cha_label_mirror[j]; C.msg_m[j] = c->msg_label_mirror[j];
    }
    return LUTLDPC_OK;
}

// sampler -> d_cha_t / d_msg0_t (tile layout); stats zeroed and slicer errors accumulated
int sample_tiles(lutldpc_decoder *d, const ChannelCells &C, uint64_t seed, uint32_t stream, uint64_t frame0, int B, const uint8_t *codewords_host) {
    int rc = ensure_batch(d, B);
    if (rc) return rc;
    const int Bpad = d->bpad(B), G = Bpad / d->tile(), N = d->nvar;
    HIP_TRY(d->d_stats.alloc((size_t)Bpad * 4));
    HIP_TRY(hipMemsetAsync(d->d_stats.p, 0, sizeof(int32_t) * (size_t)Bpad * 4, d->stream));
    const uint8_t *cw = nullptr;
    if (codewords_host) {
        HIP_TRY(d->d_codewords.alloc((size_t)B * N));
        HIP_TRY(hipMemcpyAsync(d->d_codewords.p, codewords_host, (size_t)B * N, hipMemcpyHostToDevice, d->stream));
        cw = d->d_codewords.p;
    }
    Timed t(d, LUTLDPC_K_FRONTEND);
    const int ppt = 8, npairs = (N + 1) / 2;
    dim3 grid((unsigned)((npairs + 4 * ppt - 1) / (4 * ppt)), (unsigned)G);
    DEV_PARAM(dC, d, C);
    PACK_DISPATCH(d, launch_k(sample_labels_kernel<PK>, grid, dim3(256), 0, d->stream, dC, (uint32_t)seed, (uint32_t)(seed >> 32), stream, frame0, B, N, cw,
                       d->d_cha_t.p, d->d_msg0_t.p, d->d_stats.p, ppt));
    LAUNCH_CHECK();
    return LUTLDPC_OK;
}

void make_describe(lutldpc_decoder *d) {
    std::ostringstream o;
    o << "{\"build\":\"" << __DATE__ << " " << __TIME__ << "\",\"kernel_sources\":\"" <<
#include "kernel_src_hash.inc"
      << "\",\"tile_frames\":" << d->tile() << ",\"message_bytes\":" << (d->pack == 2 ? "0.5" : "1") << ",\"pack\":" << d->pack << ",\"vector_bytes_per_lane\":4"
      << ",\"nodes_per_block\":" << d->nodes_per_block << ",\"vn_edges_per_wave\":" << d->vn_edges_per_wave << ",\"cn_edges_per_wave\":" << d->cn_edges_per_wave << ",\"use_fast\":" << d->use_fast
      << ",\"vn_classes\":[";
    for (size_t i = 0; i < d->vclass.size(); i++) {
        const bool f = d->use_fast && !d->var_fast.empty() && i < d->var_fast[0].size() && d->var_fast[0][i].ok && d->vclass[i].deg <= kFastMaxDeg;
        o << (i ? "," : "") << "{\"deg\":" << d->vclass[i].deg << ",\"nodes\":" << d->vclass[i].nodes.size() << ",\"kernel\":\""
          << (f ? "vn_balanced_fast_kernel" : (!d->var_jit.empty() && i < d->var_jit[0].size() && d->var_jit[0][i]) ? "lutldpc_jit_pass" : "tree_pass_kernel<VAR>") << "\"}";
    }
    o << "],\"cn_classes\":[";
    for (size_t i = 0; i < d->cclass.size(); i++) {
        const bool f = d->use_fast && d->min_lut && d->cclass[i].deg >= 2 && d->cclass[i].deg <= kFastMaxCnDeg && is_pow2(d->Nq_Msg[0] / 2);
        o << (i ? "," : "") << "{\"deg\":" << d->cclass[i].deg << ",\"nodes\":" << d->cclass[i].nodes.size() << ",\"kernel\":\""
          << (d->min_lut ? (f ? "cn_minsum_fast_kernel" : "cn_minsum_generic_kernel")
                         : (!d->chk_jit.empty() && i < d->chk_jit[0].size() && d->chk_jit[0][i]) ? "lutldpc_jit_pass" : "tree_pass_kernel<CHK>") << "\"}";
    }
    o << "],\"resident\":" << (resident_active(d) ? 1 : 0) << ",\"skewed_pipeline\":" << ((d->skew && d->skew_ok) ? 1 : 0) << ",\"fused_bucket\":" << d->fused_bucket_id << ",\"compaction\":" << (d->use_compact < 0 ? 2 : d->use_compact) << ",\"compaction_min_groups\":" << [&] { for (int G = 1; G <= 2 * kPermuteMaxGroups; G++) if (compaction_on(d, G)) return G; return -1; }() << ",\"chain_nodes\":" << (d->use_chain ? d->n_chain_nodes : 0) << ",\"placement\":" << d->place_info << "}";
    d->describe = o.str();
}

}  // namespace

// =============================================================================== C-ABI
extern "C" {

const char *lutldpc_last_error(void) { return g_err.c_str(); }
// used by the host mirror (host_capi.cpp) to report through the same channel
void lutldpc_set_last_error(const char *msg) { g_err = msg ? msg : ""; }
const char *lutldpc_version(void) { return "lut_ldpc_amd 0.1 gfx950"; }

int lutldpc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lutldpc_decoder_create(int nvar, int nchk, const int32_t *dv, const int32_t *dc, const int32_t *cn_msg_idx,
                           int Nq_Cha, const int32_t *Nq_Msg, const uint8_t *reuse_vec, int max_iters,
                           int min_lut, const char *var_trees_txt, const char *chk_trees_txt, int device,
                           lutldpc_decoder **out) {
    if (!out) return fail(LUTLDPC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (nvar <= 0 || nchk <= 0 || !dv || !dc || !cn_msg_idx || !Nq_Msg || !reuse_vec || max_iters < 1 || !var_trees_txt)
        return fail(LUTLDPC_ERR_ARG, "missing or non-positive argument");
    if (Nq_Cha < 2 || Nq_Cha > 128 || (Nq_Cha & 1)) return fail(LUTLDPC_ERR_ARG, "Nq_Cha must be even and in [2,128]");
    for (int i = 0; i < max_iters; i++)
        if (Nq_Msg[i] < 2 || Nq_Msg[i] > 128 || (Nq_Msg[i] & 1)) return fail(LUTLDPC_ERR_ARG, "Nq_Msg entries must be even and in [2,128]");
    // src/LDPC_Code_LUT.cpp:122
    if (reuse_vec[0] || reuse_vec[max_iters - 1]) return fail(LUTLDPC_ERR_ARG, "first and last iteration are exempt from tree reuse");
    std::unique_ptr<lutldpc_decoder> d(new lutldpc_decoder);
    d->nvar = nvar; d->nchk = nchk;
    d->dv.assign(dv, dv + nvar); d->dc.assign(dc, dc + nchk);
    long long ev = 0, ec = 0;
    for (int v = 0; v < nvar; v++) { if (dv[v] < 1 || dv[v] > 255) return fail(LUTLDPC_ERR_ARG, "variable degree outside [1,255]"); ev += dv[v]; }
    for (int c = 0; c < nchk; c++) { if (dc[c] < 1 || dc[c] > 255) return fail(LUTLDPC_ERR_ARG, "check degree outside [1,255]"); ec += dc[c]; }
    if (ev != ec || ev > (1ll << 30)) return fail(LUTLDPC_ERR_ARG, "sum(dv) != sum(dc)");
    d->E = (int)ev;
    d->cn_msg_idx.assign(cn_msg_idx, cn_msg_idx + d->E);
    d->vn_ptr.resize((size_t)nvar + 1); d->cn_ptr.resize((size_t)nchk + 1);
    for (int v = 0; v < nvar; v++) d->vn_ptr[(size_t)v + 1] = d->vn_ptr[(size_t)v] + dv[v];
    for (int c = 0; c < nchk; c++) d->cn_ptr[(size_t)c + 1] = d->cn_ptr[(size_t)c] + dc[c];
    {   // every edge must appear exactly once; derive chk_equ_idx (VN of each check edge)
        std::vector<int> edge_vn((size_t)d->E);
        for (int v = 0; v < nvar; v++) for (int e = d->vn_ptr[(size_t)v]; e < d->vn_ptr[(size_t)v + 1]; e++) edge_vn[(size_t)e] = v;
        std::vector<uint8_t> seen((size_t)d->E, 0);
        d->cn_vn.resize((size_t)d->E);
        for (int k = 0; k < d->E; k++) {
            int e = cn_msg_idx[k];
            if (e < 0 || e >= d->E || seen[(size_t)e]) return fail(LUTLDPC_ERR_ARG, "cn_msg_idx is not a permutation of the edges");
            seen[(size_t)e] = 1;
            d->cn_vn[(size_t)k] = edge_vn[(size_t)e];
        }
    }
    if ((uint64_t)d->E * kRowBytes >= (1ull << 32) || (uint64_t)nvar * kRowBytes >= (1ull << 32))
        return fail(LUTLDPC_ERR_UNSUPPORTED, "code too large: the rows of one frame group must stay below 4 GiB");
    build_classes(d->dv, d->vclass, d->vn_list);
    build_classes(d->dc, d->cclass, d->cn_list);
    d->Nq_Cha = Nq_Cha; d->Nq_Msg.assign(Nq_Msg, Nq_Msg + max_iters);
    d->max_iters_created = d->max_iters = max_iters; d->psc = 1; d->pisc = 0; d->min_lut = min_lut ? 1 : 0;
    d->iter_set.resize((size_t)max_iters);
    { int cum = 0; for (int i = 0; i < max_iters; i++) { cum += reuse_vec[i] ? 0 : 1; d->iter_set[(size_t)i] = cum - 1; } }
    std::string err;
    if (!parse_tree_array(var_trees_txt, d->var_trees, err)) return fail(LUTLDPC_ERR_PARSE, "var_trees_txt: " + err);
    if (!d->min_lut) {
        if (!chk_trees_txt || !parse_tree_array(chk_trees_txt, d->chk_trees, err)) return fail(LUTLDPC_ERR_PARSE, "chk_trees_txt: " + err);
    }
    if (const char *e = getenv("LUTLDPC_NODES_PER_BLOCK")) { int v = atoi(e); if (v >= 1 && v <= 4096) d->nodes_per_block = v; }
    if (const char *e = getenv("LUTLDPC_USE_FAST")) d->use_fast = atoi(e) ? 1 : 0;
    d->pack = 2;
    if (Nq_Cha > 16) d->pack = 1;
    for (int i = 0; i < max_iters; i++) if (Nq_Msg[i] > 16) d->pack = 1;
    if (const char *e = getenv("LUTLDPC_PACK")) { int v = atoi(e); if (v == 1) d->pack = 1; }
    if (const char *e = getenv("LUTLDPC_NODES_PER_WAVE")) { int v = atoi(e); if (v >= 1 && v <= 4096) d->nodes_per_wave = v; }
    d->nodes_per_wave_cn = d->nodes_per_wave;
    if (const char *e = getenv("LUTLDPC_VN_EDGES_PER_WAVE")) { int v = atoi(e); if (v >= 1 && v <= 65536) d->vn_edges_per_wave = v; }
    if (const char *e = getenv("LUTLDPC_FIRST_FROM_NODES")) d->first_from_nodes = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LUTLDPC_CN_EDGES_PER_WAVE")) { int v = atoi(e); if (v >= 1 && v <= 65536) { d->cn_edges_per_wave = v; d->cn_edges_from_env = true; } }
    if (const char *e = getenv("LUTLDPC_JIT")) d->use_jit = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LUTLDPC_COMPOSE")) d->use_compose = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LUTLDPC_COMPOSE_SPACE")) { int v = atoi(e); if (v >= 0 && v <= 65536) d->compose_space = v; }
    if (const char *e = getenv("LUTLDPC_RESIDENT")) { int v = atoi(e); d->use_resident = v >= 2 ? 2 : v ? 1 : 0; }
    if (const char *e = getenv("LUTLDPC_RESIDENT_S")) { int v = atoi(e); if (v >= 1 && v <= 64) d->resident_force_S = v; }
    if (const char *e = getenv("LUTLDPC_RESIDENT_NT")) { int v = atoi(e); if (v == 256 || v == 512 || v == 768 || v == 1024) d->resident_force_NT = v; }
    if (const char *e = getenv("LUTLDPC_RESIDENT_XCD")) d->resident_xcd = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LUTLDPC_RESIDENT_CN_PERSISTENT")) d->resident_cn_persistent = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LUTLDPC_RESIDENT_FM")) d->resident_fm = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LUTLDPC_RESIDENT_FLAG_REDUCE")) d->resident_flag_reduce = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LUTLDPC_RESIDENT_WAVES_EU")) { int v = atoi(e); if (v >= 0 && v <= 8) d->resident_waves_eu = v; }
    if (const char *e = getenv("LUTLDPC_RESIDENT_U")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4) d->resident_U = v; }
    if (const char *e = getenv("LUTLDPC_CHAIN")) d->use_chain = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LUTLDPC_CHK_FULL")) d->chk_full_labels = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LUTLDPC_PLACE")) { int v = atoi(e); if (v >= 0 && v <= 32) d->place_candidates = v; }
    if (const char *e = getenv("LUTLDPC_COMPACT")) d->use_compact = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LUTLDPC_COMPACT_KEEP")) d->compact_keep = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LUTLDPC_COMPACT_FIRST")) { int v = atoi(e); if (v >= 1) d->compact_first = v; }
    if (const char *e = getenv("LUTLDPC_COMPACT_EVERY")) { int v = atoi(e); if (v >= 1) d->compact_every = v; }
    if (const char *e = getenv("LUTLDPC_COMPACT_MARGIN")) { double v = atof(e); if (v >= 0 && v < 100) d->compact_margin = (float)v; }
    if (const char *e = getenv("LUTLDPC_COMPACT_MIN_SHARE")) { double v = atof(e); if (v >= 0 && v <= 1) d->compact_min_share = (float)v; }
    if (const char *e = getenv("LUTLDPC_GRAPH")) d->use_graph = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LUTLDPC_PRIO")) d->fused_prio = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LUTLDPC_TAIL_FRONT")) { double v = atof(e); if (v >= 0 && v < 0.9) d->tail_front = v; }
    if (const char *e = getenv("LUTLDPC_NODES_PER_WAVE_CN")) { int v = atoi(e); if (v >= 1 && v <= 4096) d->nodes_per_wave_cn = v; }
    if (const char *e = getenv("LUTLDPC_SKEW")) d->skew = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LUTLDPC_LATE_HARD")) d->late_hard = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LUTLDPC_REVERSE")) d->sweep_reverse = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LUTLDPC_VALIDATE")) { d->validate = atoi(e) ? 1 : 0; if (d->validate) d->use_graph = 0; }
    int rc = compile_all(d.get());
    if (rc) return rc;
    d->skew_ok = skew_eligible(d.get());
    {
        int max_cn = 0, max_vn = 0;
        for (auto &c : d->cclass) max_cn = std::max(max_cn, c.deg);
        for (auto &c : d->vclass) max_vn = std::max(max_vn, c.deg);
        d->fused_bucket_id = std::max(0, fused_bucket(max_vn, max_cn));
        // LUTLDPC_FUSED_BUCKET_MIN: run a code of small degrees through a wider bucket's kernel (measurement of what the bucket costs)
        if (const char *e = getenv("LUTLDPC_FUSED_BUCKET_MIN")) { int v = atoi(e); if (v >= 0 && v < kFusedBuckets && fused_bucket_rank(v) > fused_bucket_rank(d->fused_bucket_id)) d->fused_bucket_id = v; }
    }
    d->device = device;
    if (device >= 0) { rc = upload_static(d.get()); if (rc) return rc; }
    d->resident_ok = resident_eligible(d.get());
    make_describe(d.get());
    *out = d.release();
    return LUTLDPC_OK;
}

int lutldpc_decoder_destroy(lutldpc_decoder *d) {
    if (!d) return LUTLDPC_OK;
    if (d->device >= 0) {
        (void)hipSetDevice(d->device);
        if (d->stream) (void)hipStreamSynchronize(d->stream);
        for (auto &e : d->ev_live) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        for (auto &e : d->ev_pool) (void)hipEventDestroy(e);
        d->d_vn_ptr.release(); d->d_cn_ptr.release(); d->d_cn_idx.release(); d->d_cn_vn.release(); d->d_vn_list.release(); d->d_cn_list.release(); d->d_fast_idx.release(); d->d_chain_internal.release(); d->d_edge_vn.release();
        d->d_ops.release(); d->d_tables.release(); d->d_msgs.release(); d->d_cha_t.release(); d->d_msg0_t.release(); d->d_hard.release();
        d->d_state.release(); d->d_vfail.release(); d->d_iters.release(); d->d_in_cha.release(); d->d_in_msg.release(); d->d_out_bits.release();
        d->drop_graphs();
        d->d_frame_of.release(); d->d_perm.release(); d->d_tmp3.release(); d->d_ctl.release(); d->d_slot_of.release(); d->d_iters_tmp.release(); d->d_grp.release();
        d->drop_plans();
        d->params.release();
        d->d_out_iters.release(); d->d_trace.release(); d->d_llr.release(); d->d_qb_cha.release(); d->d_qb_msg.release(); d->d_map.release(); d->d_codewords.release(); d->d_stats.release();
        if (d->stream) (void)hipStreamDestroy(d->stream);
    }
    delete d;
    return LUTLDPC_OK;
}

int lutldpc_decoder_set_exit_conditions(lutldpc_decoder *d, int max_iters, int psc, int pisc) {
    if (!d) return fail(LUTLDPC_ERR_ARG, "NULL decoder");
    if (max_iters < 1 || max_iters > d->max_iters_created) return fail(LUTLDPC_ERR_ARG, "max_iters outside [1, value at creation]");
    if (!d->dec_plan[(size_t)d->iter_set[(size_t)(max_iters - 1)]].valid)
        return fail(LUTLDPC_ERR_ARG, "the tree set of iteration max_iters-1 is not a decision tree set");
    d->max_iters = max_iters; d->psc = psc ? 1 : 0; d->pisc = pisc ? 1 : 0;
    return LUTLDPC_OK;
}

int lutldpc_decoder_decode_batch_device(lutldpc_decoder *d, const uint8_t *d_cha, const uint8_t *d_msg0, int B,
                                        uint8_t *d_out_bits, int32_t *d_out_iters, int sync) {
    if (!d || !d_cha || !d_msg0 || !d_out_bits || !d_out_iters) return fail(LUTLDPC_ERR_ARG, "NULL argument");
    int rc = decode_device(d, d_cha, d_msg0, B, d_out_bits, d_out_iters);
    if (rc) return rc;
    if (sync) HIP_TRY(hipStreamSynchronize(d->stream));
    return LUTLDPC_OK;
}

int lutldpc_decoder_decode_batch(lutldpc_decoder *d, const uint8_t *cha, const uint8_t *msg0, int B, uint8_t *out_bits, int32_t *out_iters) {
    if (!d || !cha || !msg0 || !out_bits || !out_iters) return fail(LUTLDPC_ERR_ARG, "NULL argument");
    if (d->device < 0) return fail(LUTLDPC_ERR_STATE, "decoder was created without a device (host-only handle)");
    if (B <= 0) return fail(LUTLDPC_ERR_ARG, "B must be positive");
    HIP_TRY(hipSetDevice(d->device));
    size_t n = (size_t)B * (size_t)d->nvar;
    HIP_TRY(d->d_in_cha.alloc(n)); HIP_TRY(d->d_in_msg.alloc(n)); HIP_TRY(d->d_out_bits.alloc(n)); HIP_TRY(d->d_out_iters.alloc((size_t)B));
    HIP_TRY(hipMemcpyAsync(d->d_in_cha.p, cha, n, hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipMemcpyAsync(d->d_in_msg.p, msg0, n, hipMemcpyHostToDevice, d->stream));
    int rc = decode_device(d, d->d_in_cha.p, d->d_in_msg.p, B, d->d_out_bits.p, d->d_out_iters.p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out_bits, d->d_out_bits.p, n, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipMemcpyAsync(out_iters, d->d_out_iters.p, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return LUTLDPC_OK;
}

// lut_decode of a small batch with the message dumps of output_verbosity = level (2: initial + after every variable update,
// 3: after every check update as well): trace[dump][B][E] label bytes in the reference's print order, n_dumps = 1 + I * (level - 1).
int lutldpc_decoder_decode_batch_trace(lutldpc_decoder *d, const uint8_t *cha, const uint8_t *msg0, int B, int level, uint8_t *out_bits, int32_t *out_iters,
                                       uint8_t *trace, int64_t trace_cap, int32_t *n_dumps) {
    if (!d || !cha || !msg0 || !out_bits || !out_iters || !trace) return fail(LUTLDPC_ERR_ARG, "NULL argument");
    if (level < 2 || level > 3) return fail(LUTLDPC_ERR_ARG, "trace level must be 2 or 3");
    if (B <= 0 || B > 4096) return fail(LUTLDPC_ERR_ARG, "the message trace is a debug path: 1..4096 frames");
    const int64_t need = (int64_t)(1 + d->max_iters * (level - 1)) * B * d->E;
    if (trace_cap < need) return fail(LUTLDPC_ERR_ARG, "trace buffer too small: " + std::to_string(need) + " bytes needed");
    d->trace.level = level; d->trace.host = trace; d->trace.cap = (size_t)trace_cap; d->trace.n = 0; d->trace.B = B;
    const int rc = lutldpc_decoder_decode_batch(d, cha, msg0, B, out_bits, out_iters);
    if (n_dumps) *n_dumps = d->trace.n;
    d->trace = lutldpc_decoder::Trace();
    return rc;
}

int lutldpc_decoder_decode_llr_batch(lutldpc_decoder *d, const double *llr, int B, const double *qb_Cha, int n_qb_Cha,
                                     const double *qb_Msg, int n_qb_Msg, int mode, const int32_t *map,
                                     uint8_t *out_bits, int32_t *out_iters) {
    if (!d || !llr || !qb_Cha || !out_bits || !out_iters) return fail(LUTLDPC_ERR_ARG, "NULL argument");
    if (d->device < 0) return fail(LUTLDPC_ERR_STATE, "decoder was created without a device (host-only handle)");
    if (B <= 0) return fail(LUTLDPC_ERR_ARG, "B must be positive");
    if (n_qb_Cha != d->Nq_Cha - 1) return fail(LUTLDPC_ERR_ARG, "qb_Cha must hold Nq_Cha-1 boundaries");
    if (mode == 0 && (!qb_Msg || n_qb_Msg != d->Nq_Msg[0] - 1)) return fail(LUTLDPC_ERR_ARG, "qb_Msg must hold Nq_Msg[0]-1 boundaries");
    if (mode == 1 && !map) return fail(LUTLDPC_ERR_ARG, "QCHA mode needs cha2msg_map");
    if (mode != 0 && mode != 1) return fail(LUTLDPC_ERR_ARG, "initial_message_mode must be 0 (CONT) or 1 (QCHA)");   // src/LDPC_Code_LUT.cpp:218-220
    HIP_TRY(hipSetDevice(d->device));
    size_t n = (size_t)B * (size_t)d->nvar;
    HIP_TRY(d->d_llr.alloc(n)); HIP_TRY(d->d_in_cha.alloc(n)); HIP_TRY(d->d_in_msg.alloc(n)); HIP_TRY(d->d_out_bits.alloc(n)); HIP_TRY(d->d_out_iters.alloc((size_t)B));
    HIP_TRY(d->d_qb_cha.alloc((size_t)n_qb_Cha)); HIP_TRY(d->d_qb_msg.alloc((size_t)(n_qb_Msg > 0 ? n_qb_Msg : 1))); HIP_TRY(d->d_map.alloc((size_t)d->Nq_Cha));
    HIP_TRY(hipMemcpyAsync(d->d_llr.p, llr, n * sizeof(double), hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipMemcpyAsync(d->d_qb_cha.p, qb_Cha, sizeof(double) * (size_t)n_qb_Cha, hipMemcpyHostToDevice, d->stream));
    if (mode == 0) HIP_TRY(hipMemcpyAsync(d->d_qb_msg.p, qb_Msg, sizeof(double) * (size_t)n_qb_Msg, hipMemcpyHostToDevice, d->stream));
    else HIP_TRY(hipMemcpyAsync(d->d_map.p, map, sizeof(int32_t) * (size_t)d->Nq_Cha, hipMemcpyHostToDevice, d->stream));
    launch_k(quantize_llr_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d->stream, d->d_llr.p, n, d->d_qb_cha.p, n_qb_Cha,
                       d->d_qb_msg.p, n_qb_Msg, mode, d->d_map.p, d->d_in_cha.p, d->d_in_msg.p);
    LAUNCH_CHECK();
    int rc = decode_device(d, d->d_in_cha.p, d->d_in_msg.p, B, d->d_out_bits.p, d->d_out_iters.p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out_bits, d->d_out_bits.p, n, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipMemcpyAsync(out_iters, d->d_out_iters.p, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return LUTLDPC_OK;
}

int lutldpc_decoder_sim_batch(lutldpc_decoder *d, const lutldpc_channel_cells *cells, uint64_t seed, uint32_t stream, uint64_t frame0, int B,
                              const uint8_t *codewords, int K_info, int32_t *frame_stats, uint8_t *cha_out, uint8_t *bits_out) {
    if (!d || !frame_stats) return fail(LUTLDPC_ERR_ARG, "NULL argument");
    if (d->device < 0) return fail(LUTLDPC_ERR_STATE, "decoder was created without a device (host-only handle)");
    if (B <= 0 || K_info < 0 || K_info > d->nvar) return fail(LUTLDPC_ERR_ARG, "bad B / K_info");
    ChannelCells C;
    int rc = fill_cells(cells, d, C);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(d->device));
    if ((rc = sample_tiles(d, C, seed, stream, frame0, B, codewords))) return rc;
    if ((rc = decode_tiles(d, B))) return rc;
    {
        Timed t(d, LUTLDPC_K_FRONTEND);
        const int Bpad = d->bpad(B), G = Bpad / d->tile(), rpw = 64;
        const int rows = K_info > 0 ? K_info : 1;
        PACK_DISPATCH(d, launch_k(count_errors_kernel<PK>, dim3((unsigned)((rows + 4 * rpw - 1) / (4 * rpw)), (unsigned)G), dim3(256), 0, d->stream, d->d_hard.p,
                           codewords ? d->d_codewords.p : nullptr, B, d->nvar, K_info, d->d_iters.p, d->d_stats.p, rpw));
        LAUNCH_CHECK();
    }
    HIP_TRY(hipMemcpyAsync(frame_stats, d->d_stats.p, sizeof(int32_t) * (size_t)B * 4, hipMemcpyDeviceToHost, d->stream));
    if (cha_out || bits_out) {
        const int Bpad = d->bpad(B), G = Bpad / d->tile(), N = d->nvar;
        const size_t n = (size_t)B * N;
        HIP_TRY(d->d_out_bits.alloc(n));
        if (cha_out) {
            if ((rc = launch_transpose_out(d, d->d_cha_t.p, d->d_out_bits.p, B, G))) return rc;
            LAUNCH_CHECK();
            HIP_TRY(hipMemcpyAsync(cha_out, d->d_out_bits.p, n, hipMemcpyDeviceToHost, d->stream));
        }
        if (bits_out) {
            if ((rc = launch_transpose_out(d, d->d_hard.p, d->d_out_bits.p, B, G))) return rc;
            LAUNCH_CHECK();
            HIP_TRY(hipMemcpyAsync(bits_out, d->d_out_bits.p, n, hipMemcpyDeviceToHost, d->stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(d->stream));
    return LUTLDPC_OK;
}

int lutldpc_decoder_sample_labels(lutldpc_decoder *d, const lutldpc_channel_cells *cells, uint64_t seed, uint32_t stream, uint64_t frame0, int B,
                                  const uint8_t *codewords, uint8_t *cha, uint8_t *msg0) {
    if (!d || !cha || !msg0) return fail(LUTLDPC_ERR_ARG, "NULL argument");
    if (d->device < 0) return fail(LUTLDPC_ERR_STATE, "decoder was created without a device (host-only handle)");
    if (B <= 0) return fail(LUTLDPC_ERR_ARG, "B must be positive");
    ChannelCells C;
    int rc = fill_cells(cells, d, C);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(d->device));
    if ((rc = sample_tiles(d, C, seed, stream, frame0, B, codewords))) return rc;
    const int Bpad = d->bpad(B), G = Bpad / d->tile(), N = d->nvar;
    const size_t n = (size_t)B * N;
    HIP_TRY(d->d_in_cha.alloc(n)); HIP_TRY(d->d_in_msg.alloc(n));
    if ((rc = launch_transpose_out(d, d->d_cha_t.p, d->d_in_cha.p, B, G))) return rc;
    if ((rc = launch_transpose_out(d, d->d_msg0_t.p, d->d_in_msg.p, B, G))) return rc;
    HIP_TRY(hipMemcpyAsync(cha, d->d_in_cha.p, n, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipMemcpyAsync(msg0, d->d_in_msg.p, n, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return LUTLDPC_OK;
}

void *lutldpc_decoder_stream(lutldpc_decoder *d) { return d ? (void *)d->stream : nullptr; }

int lutldpc_decoder_set_profiling(lutldpc_decoder *d, int enable) {
    if (!d) return fail(LUTLDPC_ERR_ARG, "NULL decoder");
    if (!enable && d->device >= 0) prof_fold(d);
    d->profiling = enable != 0;
    return LUTLDPC_OK;
}
int lutldpc_decoder_get_profile(lutldpc_decoder *d, int kind, double *total_ms, int64_t *launches) {
    if (!d || kind < 0 || kind >= LUTLDPC_K_COUNT) return fail(LUTLDPC_ERR_ARG, "bad kind");
    if (d->device >= 0) prof_fold(d);
    if (total_ms) *total_ms = d->prof_ms[kind];
    if (launches) *launches = d->prof_n[kind];
    return LUTLDPC_OK;
}
int lutldpc_decoder_reset_profile(lutldpc_decoder *d) {
    if (!d) return fail(LUTLDPC_ERR_ARG, "NULL decoder");
    if (d->device >= 0) prof_fold(d);
    for (int i = 0; i < LUTLDPC_K_COUNT; i++) { d->prof_ms[i] = 0; d->prof_n[i] = 0; }
    return LUTLDPC_OK;
}
int64_t lutldpc_decoder_device_bytes(lutldpc_decoder *d) {
    if (!d) return 0;
    return (int64_t)(d->d_msgs.bytes() + d->d_cha_t.bytes() + d->d_msg0_t.bytes() + d->d_hard.bytes() + d->d_state.bytes() + d->d_vfail.bytes() +
                     d->d_iters.bytes() + d->d_in_cha.bytes() + d->d_in_msg.bytes() + d->d_out_bits.bytes() + d->d_out_iters.bytes() + d->d_llr.bytes() +
                     d->d_ops.bytes() + d->d_tables.bytes() + d->d_cn_idx.bytes() + d->d_cn_vn.bytes());
}
const char *lutldpc_decoder_describe(lutldpc_decoder *d) { return d ? d->describe.c_str() : ""; }

// kind + 16: the program of the same tree after table composition (compose_tree)
// kind + 32 (checks): the program over full labels the generated check kernels run (chk_full_label_program)
static const Program *find_prog(lutldpc_decoder *d, int kind, int set, int cls) {
    if (kind == TT_CHK + 32) {
        if (set < 0 || set >= (int)d->chk_prog_full.size() || cls < 0 || cls >= (int)d->chk_prog_full[(size_t)set].size() || d->chk_full_tab[(size_t)set][(size_t)cls].second == 0) return nullptr;
        return &d->chk_prog_full[(size_t)set][(size_t)cls];
    }
    const bool comp = (kind & 16) != 0;
    kind &= 15;
    auto &v = comp ? (kind == TT_VAR ? d->var_prog_c : kind == TT_CHK ? d->chk_prog_c : d->dec_prog_c) : (kind == TT_VAR ? d->var_prog : kind == TT_CHK ? d->chk_prog : d->dec_prog);
    if (set < 0 || set >= (int)v.size() || cls < 0 || cls >= (int)v[(size_t)set].size()) return nullptr;
    return &v[(size_t)set][(size_t)cls];
}
int lutldpc_selftest_program_eval(lutldpc_decoder *d, int kind, int set, int cls, const int32_t *in, int n_in, int32_t *out, int n_out) {
    if (!d || !in || !out) return fail(LUTLDPC_ERR_ARG, "NULL argument");
    const Program *p = find_prog(d, kind, set, cls);
    if (!p) return fail(LUTLDPC_ERR_ARG, "no such program");
    if (n_in != p->n_in || n_out != p->n_out) return fail(LUTLDPC_ERR_ARG, "program arity mismatch");
    if (!eval_program(*p, in, out)) return fail(LUTLDPC_ERR_ARG, "input label outside its alphabet");
    return LUTLDPC_OK;
}
int lutldpc_selftest_program_stats(lutldpc_decoder *d, int kind, int set, int cls, int32_t *n_ops, int32_t *n_ops_naive, int32_t *n_slots) {
    if (!d) return fail(LUTLDPC_ERR_ARG, "NULL decoder");
    const Program *p = find_prog(d, kind, set, cls);
    if (!p) return fail(LUTLDPC_ERR_ARG, "no such program");
    if (n_ops) *n_ops = (int32_t)p->ops.size();
    if (n_ops_naive) *n_ops_naive = p->n_ops_naive;
    if (n_slots) *n_slots = p->n_slots;
    return LUTLDPC_OK;
}

int64_t lutldpc_selftest_jit_source(lutldpc_decoder *d, int kind, int set, int cls, char *buf, int64_t cap, int compile) {
    if (!d) return fail(LUTLDPC_ERR_ARG, "NULL decoder");
    const Program *p = find_prog(d, kind, set, cls);
    if (!p) return fail(LUTLDPC_ERR_ARG, "no such program");
    const PassPlan &plan = kind == TT_VAR ? d->var_plan[(size_t)set] : kind == TT_DEC ? d->dec_plan[(size_t)set] : d->chk_plan[(size_t)set];
    std::string src, err;
    const bool gen = kind == TT_CHK ? jit_cn_source(*p, d->cclass[(size_t)cls].deg, d->pack, plan.P.seg[cls].tab_bytes, src, err)
                                    : jit_vn_source(*p, kind, d->vclass[(size_t)cls].deg, d->pack, plan.P.seg[cls].tab_bytes, src, err);
    if (!gen) return fail(LUTLDPC_ERR_UNSUPPORTED, err);
    if (compile) {
        std::vector<char> code;
        std::string log;
        if (!jit_compile(src, code, log)) return fail(LUTLDPC_ERR_HIP, "hiprtc: " + log);
    }
    if (buf && cap > (int64_t)src.size()) std::memcpy(buf, src.c_str(), src.size() + 1);
    return (int64_t)src.size() + 1;
}

// Source of the LDS-resident decode kernel for a batch of G frame groups (jit_resident.hpp); compile != 0 also runs hiprtc (no
// device needed).  info (optional, 3 ints): sets per workgroup, threads per workgroup, LDS bytes.  Works on host-only handles.
int64_t lutldpc_selftest_resident_source(lutldpc_decoder *d, int G, char *buf, int64_t cap, int compile, int32_t *info) {
    if (!d || G < 1) return fail(LUTLDPC_ERR_ARG, "NULL decoder / bad G");
    int S = 0, NT = 0, lds = 0;
    if (!resident_pick(d, G, S, NT, lds)) return fail(LUTLDPC_ERR_UNSUPPORTED, "resident decoder: the code does not fit the LDS");
    std::string src, err;
    if (!jit_resident_source(resident_spec(d, S, NT), src, err)) return fail(LUTLDPC_ERR_UNSUPPORTED, "resident decoder: " + err);
    if (info) { info[0] = S; info[1] = NT; info[2] = lds; }
    if (compile) {
        std::vector<char> code;
        std::string log;
        if (!jit_compile(src, code, log)) return fail(LUTLDPC_ERR_HIP, "hiprtc: " + log.substr(0, 4000));
    }
    if (buf && cap > (int64_t)src.size()) std::memcpy(buf, src.c_str(), src.size() + 1);
    return (int64_t)src.size() + 1;
}

}  // extern "C"
