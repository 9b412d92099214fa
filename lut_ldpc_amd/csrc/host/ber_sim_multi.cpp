// ber_sim_multi.cpp -- ber_sim over several MI355X of one node, natively in C++ (BASELINE config 4).
//
// The reference runs one single-threaded process per seed and adds the result files up afterwards
// (scripts/aggregate_results.m:73-84).  Here ONE run of a parameter file shards the frames of every SNR point over the devices:
//   * one host thread, one LDPC_BER_Sim_LUT (its own codec replica, decoder handle and HIP stream) per *lane*; `lanes` lanes per
//     device (default 2: while lane A's decode occupies the device, lane B's sampler runs and lane A's counters of the batch
//     before go back over PCIe and through the host prefix -- the overlap a double buffer would give, without sharing buffers);
//   * frames are Philox-addressed (kernels_frontend.hpp), so rank r of R simply takes the batch [f0 + r * batch, + batch) of every
//     round; no data-path exchange;
//   * per round two tiny exchanges: all-gather of {frames, frame errors} per rank (2 x int64) and all-reduce of the rank's
//     contribution to the five counters of src/LDPC_BER_Sim.hpp:80-85 -- the batch before the stopping frame whole, the batch
//     containing it truncated, later ones nothing -- so that the stop rule of src/LDPC_BER_Sim.cpp:289 (`> Nfers`) is applied
//     in global frame order and the counters equal those of the frame-by-frame loop (same rule as lut_ldpc_amd/ber_sim.py:73-108);
//   * the exchanges run over RCCL (ncclAllGather / ncclAllReduce on the devices' own communicators, ncclCommInitAll; xGMI between
//     the GPUs of a node) with the lanes of a device folded on the host first; `host` exchange (threads + barrier only) is the
//     rehearsal mode for ranks that share a device, which RCCL refuses.
// librccl is loaded with dlopen: the library itself does not depend on it.
#include "ber_sim_driver.hpp"
#include "ini.hpp"

#include <hip/hip_runtime_api.h>
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <iostream>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>

namespace lut_ldpc {

namespace {

// ---- a reusable barrier (C++17: no std::barrier)
class Barrier {
public:
    explicit Barrier(int n) : n_(n) {}
    void wait() {
        std::unique_lock<std::mutex> lk(mu_);
        const int gen = gen_;
        if (++count_ == n_) { count_ = 0; gen_++; cv_.notify_all(); }
        else cv_.wait(lk, [&] { return gen != gen_ || failed_; });
        if (failed_) throw std::runtime_error("another rank failed");
    }
    void fail() { std::lock_guard<std::mutex> lk(mu_); failed_ = true; cv_.notify_all(); }
private:
    std::mutex mu_;
    std::condition_variable cv_;
    int n_, count_ = 0, gen_ = 0;
    bool failed_ = false;
};

// ---- RCCL through dlopen (types as in <rccl/rccl.h>: ncclInt64 = 4, ncclSum = 0)
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*AllReduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t s) = nullptr;
    int (*AllGather)(const void *send, void *recv, size_t sendcount, int dtype, void *comm, hipStream_t s) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load(std::string &err) {
        const char *name = std::getenv("LUTLDPC_RCCL_LIB");
        lib = dlopen(name && *name ? name : "librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib && !(name && *name)) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) { err = std::string("cannot load RCCL: ") + dlerror(); return false; }
        auto sym = [&](const char *s) { void *p = dlsym(lib, s); if (!p) err = std::string("RCCL symbol missing: ") + s; return p; };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        return CommInitAll && CommDestroy && AllReduce && AllGather && GetErrorString;
    }
};
constexpr int kNcclInt64 = 4, kNcclSum = 0;

// ---- the counter exchange of one run: R = devices x lanes ranks, rank = lane * n_dev + device slot
struct Exchange {
    int n_dev = 1, lanes = 1, R = 1;
    bool use_rccl = false;
    Rccl rccl;
    std::vector<int> devices;
    std::vector<void *> comms;                       // one per device slot
    std::vector<hipStream_t> streams;
    std::vector<int64_t *> d_send, d_recv;           // device buffers of the device leaders
    std::vector<int64_t> gather;                     // [R][2] of the current round
    std::vector<int64_t> reduce_in;                  // [R][5]
    std::vector<int64_t> reduce_out;                 // [5]
    std::unique_ptr<Barrier> bar;
    std::vector<std::unique_ptr<Barrier>> dev_bar;   // the lanes of one device

    void hip_check(hipError_t e, const char *what) { if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)); }
    void nccl_check(int rc, const char *what) { if (rc != 0) throw std::runtime_error(std::string(what) + ": " + rccl.GetErrorString(rc)); }

    void init(const std::vector<int> &devs, int lanes_, const std::string &mode) {
        devices = devs; n_dev = (int)devs.size(); lanes = lanes_; R = n_dev * lanes;
        gather.assign((size_t)R * 2, 0); reduce_in.assign((size_t)R * 5, 0); reduce_out.assign(5, 0);
        bar.reset(new Barrier(R));
        for (int i = 0; i < n_dev; i++) dev_bar.emplace_back(new Barrier(lanes));
        bool distinct = true;
        for (int i = 0; i < n_dev; i++) for (int j = 0; j < i; j++) distinct = distinct && devs[(size_t)i] != devs[(size_t)j];
        if (mode == "host" || (mode == "auto" && (!distinct || n_dev == 1))) return;
        if (!distinct) throw std::runtime_error("RCCL needs one device per rank: the device list repeats a device (use --exchange host to rehearse)");
        std::string err;
        if (!rccl.load(err)) { if (mode == "rccl") throw std::runtime_error(err); std::cerr << "ber_sim: " << err << " -- counters are exchanged on the host\n"; return; }
        comms.assign((size_t)n_dev, nullptr);
        nccl_check(rccl.CommInitAll(comms.data(), n_dev, devices.data()), "ncclCommInitAll");
        streams.assign((size_t)n_dev, nullptr); d_send.assign((size_t)n_dev, nullptr); d_recv.assign((size_t)n_dev, nullptr);
        for (int i = 0; i < n_dev; i++) {
            hip_check(hipSetDevice(devices[(size_t)i]), "hipSetDevice");
            hip_check(hipStreamCreateWithFlags(&streams[(size_t)i], hipStreamNonBlocking), "hipStreamCreate");
            hip_check(hipMalloc((void **)&d_send[(size_t)i], sizeof(int64_t) * (size_t)(5 + 2 * lanes)), "hipMalloc");
            hip_check(hipMalloc((void **)&d_recv[(size_t)i], sizeof(int64_t) * (size_t)(5 + 2 * R)), "hipMalloc");
        }
        use_rccl = true;
    }
    void destroy() {
        for (size_t i = 0; i < comms.size(); i++) {
            (void)hipSetDevice(devices[i]);
            if (d_send[i]) (void)hipFree(d_send[i]);
            if (d_recv[i]) (void)hipFree(d_recv[i]);
            if (streams[i]) (void)hipStreamDestroy(streams[i]);
            if (comms[i]) (void)rccl.CommDestroy(comms[i]);
        }
        comms.clear();
    }

    // every rank contributes mine[2]; afterwards all[R][2] holds every rank's pair, in rank order
    void all_gather2(int rank, const int64_t *mine, int64_t *all) {
        const int slot = rank % n_dev, lane = rank / n_dev;
        gather[(size_t)rank * 2] = mine[0]; gather[(size_t)rank * 2 + 1] = mine[1];
        if (use_rccl) {
            dev_bar[(size_t)slot]->wait();                                 // the lanes of this device have written their pairs
            if (lane == 0) {
                // payload of this device: its lanes' pairs; gathered order = [device slot][lane]
                std::vector<int64_t> h((size_t)2 * lanes), g((size_t)2 * R);
                for (int l = 0; l < lanes; l++) { h[(size_t)2 * l] = gather[(size_t)(l * n_dev + slot) * 2]; h[(size_t)2 * l + 1] = gather[(size_t)(l * n_dev + slot) * 2 + 1]; }
                hip_check(hipSetDevice(devices[(size_t)slot]), "hipSetDevice");
                hip_check(hipMemcpyAsync(d_send[(size_t)slot], h.data(), sizeof(int64_t) * h.size(), hipMemcpyHostToDevice, streams[(size_t)slot]), "hipMemcpyAsync");
                nccl_check(rccl.AllGather(d_send[(size_t)slot], d_recv[(size_t)slot], (size_t)2 * lanes, kNcclInt64, comms[(size_t)slot], streams[(size_t)slot]), "ncclAllGather");
                hip_check(hipMemcpyAsync(g.data(), d_recv[(size_t)slot], sizeof(int64_t) * g.size(), hipMemcpyDeviceToHost, streams[(size_t)slot]), "hipMemcpyAsync");
                hip_check(hipStreamSynchronize(streams[(size_t)slot]), "hipStreamSynchronize");
                if (slot == 0)       // one writer puts what came over the links into the shared table (every device received the same)
                    for (int s2 = 0; s2 < n_dev; s2++) for (int l = 0; l < lanes; l++) { gather[(size_t)(l * n_dev + s2) * 2] = g[(size_t)(s2 * lanes + l) * 2]; gather[(size_t)(l * n_dev + s2) * 2 + 1] = g[(size_t)(s2 * lanes + l) * 2 + 1]; }
            }
        }
        bar->wait();
        std::memcpy(all, gather.data(), sizeof(int64_t) * (size_t)R * 2);
        bar->wait();                                                       // nobody overwrites the table before everybody has read it
    }
    // sum of every rank's v[5], returned in v on every rank
    void all_reduce5(int rank, int64_t *v) {
        const int slot = rank % n_dev, lane = rank / n_dev;
        std::memcpy(&reduce_in[(size_t)rank * 5], v, sizeof(int64_t) * 5);
        if (use_rccl) {
            dev_bar[(size_t)slot]->wait();
            if (lane == 0) {
                int64_t h[5] = {0, 0, 0, 0, 0}, g[5];
                for (int l = 0; l < lanes; l++) for (int k = 0; k < 5; k++) h[k] += reduce_in[(size_t)(l * n_dev + slot) * 5 + (size_t)k];
                hip_check(hipSetDevice(devices[(size_t)slot]), "hipSetDevice");
                hip_check(hipMemcpyAsync(d_send[(size_t)slot], h, sizeof(h), hipMemcpyHostToDevice, streams[(size_t)slot]), "hipMemcpyAsync");
                nccl_check(rccl.AllReduce(d_send[(size_t)slot], d_recv[(size_t)slot], 5, kNcclInt64, kNcclSum, comms[(size_t)slot], streams[(size_t)slot]), "ncclAllReduce");
                hip_check(hipMemcpyAsync(g, d_recv[(size_t)slot], sizeof(g), hipMemcpyDeviceToHost, streams[(size_t)slot]), "hipMemcpyAsync");
                hip_check(hipStreamSynchronize(streams[(size_t)slot]), "hipStreamSynchronize");
                if (slot == 0) std::memcpy(reduce_out.data(), g, sizeof(g));
            }
            bar->wait();
        } else {
            bar->wait();
            if (rank == 0) for (int k = 0; k < 5; k++) { int64_t s = 0; for (int r = 0; r < R; r++) s += reduce_in[(size_t)r * 5 + (size_t)k]; reduce_out[(size_t)k] = s; }
            bar->wait();
        }
        std::memcpy(v, reduce_out.data(), sizeof(int64_t) * 5);
        bar->wait();
    }
};

// counters of the frames of st[0..n) (in order) up to and including the one that makes the running frame-error count exceed
// nfers (lut_ldpc_amd/ber_sim.py: _prefix_until_stop)
void prefix_until_stop(const FrameStats *st, int n, int K, int64_t nfers, int64_t ferr_before, int64_t out[5]) {
    int64_t run = ferr_before;
    int m = n;
    for (int i = 0; i < n; i++) { run += st[i].frame_error ? 1 : 0; if (run > nfers) { m = i + 1; break; } }
    out[0] = m; out[1] = (int64_t)m * K; out[2] = out[3] = out[4] = 0;
    for (int i = 0; i < m; i++) { out[2] += st[i].frame_error ? 1 : 0; out[3] += st[i].bit_errors; out[4] += st[i].uncoded_errors; }
}

}  // namespace

// The frame loop of sim_snr_point (src/LDPC_BER_Sim.cpp:260-291) of one rank among ex.R; returns the counters of the point
// (identical on every rank).
static SnrPointCounters sim_snr_point_sharded(LDPC_BER_Sim &sim, Exchange &ex, int rank, double snr, int snr_index) {
    const int K = sim.get_dataword_length(), R = ex.R;
    const int64_t total_frames = (int64_t)sim.Nframes, nfers = sim.Nfers;
    int64_t tot[5] = {0, 0, 0, 0, 0};
    int64_t f0 = 0;
    int batch = std::min(512, sim.batch_frames);            // one full frame group of nibble rows
    std::vector<FrameStats> stats;
    std::vector<int64_t> all((size_t)R * 2);
    while (f0 < total_frames) {
        const int64_t lo = std::min(total_frames, f0 + (int64_t)rank * batch), hi = std::min(total_frames, lo + batch);
        const int B = (int)(hi - lo);
        stats.assign((size_t)std::max(B, 0), FrameStats{});
        if (B > 0) sim.sim_batch(snr, snr_index, lo, B, stats.data());
        int64_t mine[2] = {B, 0};
        for (int i = 0; i < B; i++) mine[1] += stats[(size_t)i].frame_error ? 1 : 0;
        ex.all_gather2(rank, mine, all.data());
        int64_t ferr_before = tot[2], run = tot[2];
        for (int r = 0; r < rank; r++) ferr_before += all[(size_t)r * 2 + 1];
        int first_stop = -1;
        for (int r = 0; r < R; r++) { run += all[(size_t)r * 2 + 1]; if (run > nfers) { first_stop = r; break; } }
        int64_t contrib[5] = {0, 0, 0, 0, 0};
        if (first_stop < 0 || rank < first_stop) prefix_until_stop(stats.data(), B, K, INT64_MAX / 2, 0, contrib);
        else if (rank == first_stop) prefix_until_stop(stats.data(), B, K, nfers, ferr_before, contrib);
        ex.all_reduce5(rank, contrib);
        for (int k = 0; k < 5; k++) tot[k] += contrib[k];
        if (first_stop >= 0) break;
        f0 += (int64_t)R * batch;
        batch = std::min(batch * 4, sim.batch_frames);
    }
    SnrPointCounters c;
    c.frames = tot[0]; c.databits = tot[1]; c.frame_errors = tot[2]; c.data_bit_errors = tot[3]; c.uncoded_bit_errors = tot[4];
    return c;
}

int ber_sim_run_multi(const std::string &params_path, const std::string &base_dir, int seed, const std::string &custom_name,
                      const std::vector<int> &devices, int lanes, const std::string &exchange_mode, bool quiet) {
    if (devices.empty() || lanes < 1 || lanes > 8) throw std::runtime_error("ber_sim: need at least one device and 1..8 lanes");
    Ini ini(params_path);
    const bool is_lut = ini.has_section("LUT") || ini.get("Sim.codec_type", "none") == "LUT";
    const bool is_bp = !is_lut && (ini.has_section("BP") || ini.get("Sim.codec_type", "none") == "BP");
    if (!is_lut && !is_bp) throw std::runtime_error("You must specify the type of decoder in the params file ([LUT] section or Sim.codec_type)");
    Exchange ex;
    ex.init(devices, lanes, exchange_mode);
    const int R = ex.R;
    std::vector<std::unique_ptr<LDPC_BER_Sim>> sims((size_t)R);
    std::vector<std::string> errors((size_t)R);
    std::atomic<bool> failed{false};
    Barrier start(R);
    auto worker = [&](int rank) {
        try {
            std::unique_ptr<LDPC_BER_Sim> sim;
            if (is_lut) sim.reset(new LDPC_BER_Sim_LUT(params_path, base_dir)); else sim.reset(new LDPC_BER_Sim_BP(params_path, base_dir));
            sim->rand_seed = seed;
            sim->device = devices[(size_t)(rank % ex.n_dev)];
            sim->append_custom_name(custom_name);
            sim->quiet = true;
            if (rank != 0) sim->save_codec = -1 - seed;             // one rank writes lut_codec.it
            sim->load();
            sims[(size_t)rank] = std::move(sim);
            LDPC_BER_Sim &S = *sims[(size_t)rank];
            start.wait();
            const auto t0 = std::chrono::steady_clock::now();
            size_t ss = 0;
            const int N = S.get_codeword_length();
            while (ss < S.SNRdB.size()) {                            // src/LDPC_BER_Sim.cpp:121-155 on every rank, same decisions everywhere
                const SnrPointCounters c = sim_snr_point_sharded(S, ex, rank, S.SNRdB[ss], (int)ss);
                const double ber = c.databits ? (double)c.data_bit_errors / (double)c.databits : 0.0;
                const double uber = c.frames ? (double)c.uncoded_bit_errors / ((double)c.frames * N) : 0.0;
                const double fer = c.frames ? (double)c.frame_errors / (double)c.frames : 0.0;
                if (rank == 0 && !quiet)
                    std::cout << "SNR = " << S.SNRdB[ss] << "  Simulated " << c.frames << " frames and " << c.databits << " data bits. "
                              << "Obtained " << c.data_bit_errors << " data bit errors. " << " Data BER: " << ber << " Uncoded BER: " << uber
                              << " FER: " << fer << std::endl << std::flush;
                S.results.add_snr_point(S.SNRdB[ss], c.frames, c.databits, c.frame_errors, c.data_bit_errors, c.uncoded_bit_errors);
                ss++;
                if (ber < S.ber_min || fer < S.fer_min) break;       // :307
            }
            for (; ss < S.SNRdB.size(); ss++) S.results.add_snr_point(S.SNRdB[ss], 0, 0, 0, 0, 0);     // :142-149
            S.results.save_runtime(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        } catch (const std::exception &e) {
            errors[(size_t)rank] = e.what();
            failed = true;
            start.fail(); ex.bar->fail();
            for (auto &b : ex.dev_bar) b->fail();
        }
    };
    std::vector<std::thread> pool;
    for (int r = 1; r < R; r++) pool.emplace_back(worker, r);
    worker(0);
    for (auto &t : pool) t.join();
    ex.destroy();
    if (failed) {
        std::string msg;
        for (int r = 0; r < R; r++) if (!errors[(size_t)r].empty() && errors[(size_t)r] != "another rank failed") msg += "rank " + std::to_string(r) + ": " + errors[(size_t)r] + "; ";
        throw std::runtime_error(msg.empty() ? "a rank failed" : msg);
    }
    if (!quiet)
        std::cout << "Done simulating on " << ex.n_dev << " device(s) x " << lanes << " lane(s), counters over " << (ex.use_rccl ? "RCCL" : "the host")
                  << ". Runtime = " << sims[0]->results.runtime << " seconds" << std::endl;
    sims[0]->save();
    return 0;
}

}  // namespace lut_ldpc
