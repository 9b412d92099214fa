// ber_sim_driver.cpp -- see ber_sim_driver.hpp.  Reference: src/LDPC_BER_Sim.cpp, prog/ber_sim.cpp.
#include "ber_sim_driver.hpp"
#include "ini.hpp"
#include "itfile.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <cstdlib>

namespace fs = std::filesystem;

namespace lut_ldpc {

// ------------------------------------------------------------------ results
void LDPC_BER_Sim_Results::add_snr_point(double snr, int64_t frames, int64_t databits, int64_t ferr, int64_t berr, int64_t uerr) {
    sim_SNRdB.push_back(snr); sim_Nframes.push_back(frames); sim_Ndatabits.push_back(databits);
    sim_frame_errors.push_back(ferr); sim_data_bit_errors.push_back(berr); sim_uncoded_bit_errors.push_back(uerr);
}

void LDPC_BER_Sim_Results::write_itfile(const std::string &filename) const {
    auto as_vec = [](const std::vector<int64_t> &v) { return std::vector<double>(v.begin(), v.end()); };   // to_vec(): counters go out as doubles
    it_file_writer f(filename);
    f.write("sim_SNRdB", sim_SNRdB);
    f.write("sim_Nframes", as_vec(sim_Nframes));
    f.write("sim_Ndatabits", as_vec(sim_Ndatabits));
    f.write("sim_frame_errors", as_vec(sim_frame_errors));
    f.write("sim_data_bit_errors", as_vec(sim_data_bit_errors));
    f.write("sim_uncoded_bit_errors", as_vec(sim_uncoded_bit_errors));
    f.write("ldpc_nvar", std::vector<double>{(double)ldpc_nvar});
    f.write("ldpc_nchk", std::vector<double>{(double)ldpc_nchk});
    f.write("ldpc_code_rate", std::vector<double>{ldpc_code_rate});
    f.write("runtime", runtime);
    f.write("gitversion", std::string("lut_ldpc_amd-0.1"));
    f.close();
}

// ------------------------------------------------------------------ channel cells
lutldpc_channel_cells ChannelCellTable::view() const {
    lutldpc_channel_cells v;
    v.n_cells = (int32_t)cha.size();
    v.thr = thr.data(); v.cha_label = cha.data(); v.msg_label = msg.data(); v.slicer_neg = neg.data();
    v.cha_label_mirror = cha_m.data(); v.msg_label_mirror = msg_m.data();
    return v;
}

ChannelCellTable make_channel_cells(double N0, const vec &qb_Cha, const vec &qb_Msg, int mode, const ivec &map) {
    // thresholds in the domain of the received value x: LLR = 4x/N0 > b  <=>  x > b*N0/4
    struct T { double t; bool is_cha, is_msg; };
    std::vector<T> th;
    auto add = [&](double t, bool c, bool m) {
        for (auto &e : th) if (e.t == t) { e.is_cha |= c; e.is_msg |= m; return; }
        th.push_back({t, c, m});
    };
    for (double b : qb_Cha) add(b * N0 / 4, true, false);
    if (mode == 0) for (double b : qb_Msg) add(b * N0 / 4, false, true);
    add(0.0, false, false);
    std::sort(th.begin(), th.end(), [](const T &a, const T &b) { return a.t < b.t; });
    const int n = (int)th.size();
    const int Nq_Cha = (int)qb_Cha.size() + 1, Nq_Msg = mode == 0 ? (int)qb_Msg.size() + 1 : 0;
    const double sigma = std::sqrt(N0 / 2);
    const double two64 = 18446744073709551616.0;
    ChannelCellTable C;
    int lc = 0, lm = 0;
    for (int j = 0; j <= n; j++) {      // cell j = (th[j-1], th[j]]
        C.cha.push_back((uint8_t)lc);
        C.msg.push_back((uint8_t)(mode == 0 ? lm : map.at((size_t)lc)));
        C.neg.push_back((uint8_t)((j < n && th[(size_t)j].t <= 0.0) ? 1 : 0));
        C.cha_m.push_back((uint8_t)(Nq_Cha - 1 - lc));
        C.msg_m.push_back((uint8_t)(mode == 0 ? Nq_Msg - 1 - lm : map.at((size_t)(Nq_Cha - 1 - lc))));
        if (j == n) break;
        // P(x <= t | +1 sent), x ~ N(1, sigma^2); evaluated from the nearer tail
        const double z = (th[(size_t)j].t - 1.0) / sigma;
        uint64_t thr;
        if (z <= 0) {
            const double w = 0.5 * std::erfc(-z * 0.70710678118654752440) * two64;
            thr = w >= two64 ? UINT64_MAX : (uint64_t)w;
        } else {
            const double w = 0.5 * std::erfc(z * 0.70710678118654752440) * two64;
            thr = UINT64_MAX - (w >= two64 ? UINT64_MAX : (uint64_t)w);
        }
        if (!C.thr.empty() && thr < C.thr.back()) thr = C.thr.back();
        C.thr.push_back(thr);
        if (th[(size_t)j].is_cha) lc++;
        if (th[(size_t)j].is_msg) lm++;
    }
    return C;
}

namespace {
inline void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
}  // namespace

void random_info_bits(uint64_t seed, uint32_t stream, uint64_t frame, int K, unsigned char *out) {
    for (int k0 = 0; k0 < K; k0 += 128) {
        uint32_t c[4] = {(uint32_t)frame, (uint32_t)(frame >> 32), (uint32_t)(k0 / 128), stream | 0x80000000u};
        uint32_t ka = (uint32_t)seed, kb = (uint32_t)(seed >> 32);
        for (int r = 0; r < 10; r++) { philox_round(c, ka, kb); ka += 0x9E3779B9u; kb += 0xBB67AE85u; }
        for (int k = k0; k < std::min(K, k0 + 128); k++) out[k] = (unsigned char)((c[(k - k0) / 32] >> ((k - k0) % 32)) & 1u);
    }
}

bool accumulate_in_order(const FrameStats *st, int n, int K, int64_t Nfers, SnrPointCounters &c) {
    for (int i = 0; i < n; i++) {
        c.frames++; c.databits += K;
        c.frame_errors += st[i].frame_error ? 1 : 0;
        c.data_bit_errors += st[i].bit_errors;
        c.uncoded_bit_errors += st[i].uncoded_errors;
        if (c.frame_errors > Nfers) return true;          // src/LDPC_BER_Sim.cpp:289
    }
    return false;
}

// ------------------------------------------------------------------ LDPC_BER_Sim
namespace {
std::string join(const std::string &base, const std::string &p) { return fs::path(p).is_relative() ? (fs::path(base) / p).string() : p; }
}  // namespace

LDPC_BER_Sim::LDPC_BER_Sim(const std::string &params, const std::string &base) : params_file_path(params), base_dir(base) {
    if (!fs::exists(params)) throw std::runtime_error("Parameter file" + params + " does not exist!");
    Ini ini(params);
    SNRdB = parse_vec(ini.require("Sim.SNRdB"));
    Nframes = ini.get("Sim.Nframes", 1e5);
    Nfers = ini.get("Sim.Nfers", 20);
    ber_min = ini.get("Sim.ber_min", 1e-7);
    fer_min = ini.get("Sim.fer_min", 1e-5);
    rand_seed_offset = ini.get("Sim.rand_seed_offset", 0);
    save_codec = ini.get("Sim.save_codec", 0);
    custom_name = ini.get("Sim.custom_name", "");
    results_prefix = ini.get("Sim.results_prefix", "RES");
    results_dir = ini.get("Sim.results_dir", "results");
    codes_dir = ini.get("Sim.codes_dir", "codes");
    codec_filename = ini.get("Sim.codec_filename", "");
    parity_filename = ini.get("LDPC.parity_filename", "");
    zero_codeword = ini.get("LDPC.zero_codeword", true);
    save_permuted = ini.get("LDPC.save_permuted", false);
    parity_check_iter = ini.get("LDPC.parity_check_iter", true);
    max_iter = ini.get("BP.max_iter", 30);
    // build-side: upper bound of the frames per device call.  32768 is the measured optimum on MI355X for the N = 64800 codes
    // (+7 % over 4096: launch tails weigh less; +3 % over 16384 with early termination: two halves of 32 frame groups, the most
    // the compaction of the surviving frames takes; 6.8 GB of rows) and costs the short codes nothing; the frame loop still
    // starts at 512 frames (one frame group) and quadruples (sim_snr_point), so a point that stops after Nfers errors wastes little work
    batch_frames = ini.get("Sim.batch_frames", 32768);
    codes_path = join(base, codes_dir);
    results_path = join(base, results_dir);
    fs::create_directories(codes_path);
    fs::create_directories(results_path);
}

std::string LDPC_BER_Sim::gen_filename() const {
    if (!decoder_set) throw std::logic_error("LDPC_BER_Sim::gen_filename(): Decoder has not been set!");
    std::ostringstream fn;
    fn << results_prefix << "_N" << codeword_length << "_R" << code_rate << "_maxIter" << max_iter << "_zcw" << (int)zero_codeword
       << "_frames" << Nframes << custom_name;
    return fn.str();
}

void LDPC_BER_Sim::run() {
    if (!decoder_set) throw std::logic_error("LDPC_BER_Sim::run(): Decoder has not been set!");
    const auto t0 = std::chrono::steady_clock::now();
    size_t ss = 0;
    while (ss < SNRdB.size()) {
        const bool exit_cond = sim_snr_point(SNRdB[ss], (int)ss);
        ss++;
        if (exit_cond) break;
    }
    for (; ss < SNRdB.size(); ss++) results.add_snr_point(SNRdB[ss], 0, 0, 0, 0, 0);     // :142-149
    const double runtime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    results.save_runtime(runtime);
    if (!quiet) std::cout << "Done simulating. Runtime = " << runtime << " seconds" << std::endl;
}

std::string LDPC_BER_Sim::results_file_path() const {
    std::ostringstream name;
    name << gen_filename() << "_rseed" << std::setfill('0') << std::setw(4) << rand_seed + rand_seed_offset << ".it";
    return (fs::path(results_path) / gen_filename() / name.str()).string();
}

void LDPC_BER_Sim::save() {
    const fs::path sub = fs::path(results_path) / gen_filename();
    fs::create_directories(sub);
    results.write_itfile(results_file_path());
    const fs::path dst = sub / fs::path(params_file_path).filename();
    if (!fs::exists(dst)) { std::error_code ec; fs::copy_file(params_file_path, dst, ec); }
}

// ------------------------------------------------------------------ LDPC_BER_Sim_LUT
LDPC_BER_Sim_LUT::LDPC_BER_Sim_LUT(const std::string &params, const std::string &base) : LDPC_BER_Sim(params, base) {
    Ini ini(params);
    max_iter = ini.get("LUT.max_iter", 30);
    if (auto v = ini.lookup("LUT.design_thr")) design_thr = std::stod(*v);
    if (auto v = ini.lookup("LUT.design_SNRdB")) design_SNRdB = std::stod(*v);
    decoder_output_verbosity = ini.get("LUT.output_verbosity", 0);
    initial_message_mode = ini.get("LUT.initial_message_mode", "from_continuous_input");
    if (!design_thr && !design_SNRdB && codec_filename.empty())
        throw std::runtime_error("LDPC_BER_Sim_LUT::LDPC_BER_Sim_LUT(): No design SNR or noise thresold specified");
    Nq_Cha = 1 << ini.get("LUT.qbits_channel", 4);
    const int qbits_msg = ini.get("LUT.qbits_message_uniform", 3);
    if (auto v = ini.lookup("LUT.qbits_messages")) {
        for (double q : parse_vec(*v)) Nq_Msg.push_back((int)std::pow(2.0, q));
    } else Nq_Msg.assign((size_t)max_iter, 1 << qbits_msg);
    tree_mode = ini.get("LUT.tree_mode", "auto_bin_balanced");
    trees_dir = ini.get("LUT.trees_dir", "trees");
    trees_filename = ini.get("LUT.trees_filename", "");
    min_lut = ini.get("LUT.min_lut", true);
    if (auto v = ini.lookup("LUT.reuse_lut")) { for (double r : parse_vec(*v)) reuse_lut.push_back(r != 0 ? 1 : 0); }
    else reuse_lut.assign((size_t)max_iter, 0);
    allow_degree_one = ini.get("LUT.allow_degree_one", false);
    known_rank = ini.get("LDPC.known_rank", 0);
    trees_path = join(base, trees_dir);
    fs::create_directories(trees_path);
}

void LDPC_BER_Sim_LUT::load() {
    if (codec_filename.empty()) {     // design the codec, :436-521
        const fs::path parity_path = fs::path(codes_path) / (parity_filename + ".alist");
        if (!fs::exists(parity_path)) throw std::runtime_error("Parity file" + parity_path.string() + " does not exist!");
        H.reset(new LDPC_Parity(parity_path.string()));
        if (!zero_codeword) {
            const fs::path gen_path = fs::path(codes_path) / (parity_filename + ".gen.it");
            G.reset(new LDPC_Generator_Systematic());
            if (fs::exists(gen_path)) G->load(gen_path.string());
            if (!G->is_initialized()) {
                G->construct(H.get());
                if (save_permuted) {
                    H->save_alist(parity_path.string());
                    { it_file_writer f(gen_path.string()); f.write("Fileversion", 2); f.close(); }
                    G->save(gen_path.string());
                }
            }
            encoder_set = true;
        }
        C.reset(new LDPC_Code_LUT());
        C->set_device(device);
        C->set_code_with_rank(H.get(), G.get(), known_rank);
        codeword_length = C->get_nvar(); dataword_length = C->get_ninfo(); code_rate = C->get_rate();
        decoder_set = true;
        double sigma2_design;
        if (design_thr) sigma2_design = (*design_thr) * (*design_thr);
        else if (design_SNRdB) sigma2_design = std::pow(10.0, -(*design_SNRdB) / 10) / (2 * C->get_rate());
        else throw std::runtime_error("LDPC_BER_Sim_LUT::load(): No design noise threshold specified");
        if ((int)Nq_Msg.size() != max_iter || (int)reuse_lut.size() != max_iter)
            throw std::runtime_error("LDPC_BER_Sim_LUT::load(): qbits_messages / reuse_lut must have max_iter entries");
        if (tree_mode == "auto_bin_balanced" || tree_mode == "auto_bin_high" || tree_mode == "root_only") {
            // the reference passes the literal "auto_bin_balanced" whatever auto mode was asked for (:487-489)
            C->design_luts("auto_bin_balanced", get_empirical_ensemble(*H), min_lut, sigma2_design, max_iter, reuse_lut, Nq_Cha, Nq_Msg, allow_degree_one);
        } else if (tree_mode == "file") {
            if (trees_filename.empty()) throw std::runtime_error("LDPC_BER_Sim_LUT::design_lut_codec(): Specify tree file name!");
            const fs::path tree_file = fs::path(trees_path) / trees_filename;
            if (!fs::exists(tree_file)) throw std::runtime_error("LDPC_BER_Sim_LUT::design_lut_codec(): Tree file " + tree_file.string() + " could not be located");
            C->design_luts("filename=" + tree_file.string(), get_empirical_ensemble(*H), min_lut, sigma2_design, max_iter, reuse_lut, Nq_Cha, Nq_Msg);
        } else throw std::runtime_error("LDPC_BER_Sim_LUT::load(): tree_mode " + tree_mode + " unknown");
        C->set_exit_conditions(max_iter, parity_check_iter, parity_check_iter);     // psc AND pisc, :500
        C->set_output_verbosity(decoder_output_verbosity);
        if (initial_message_mode == "from_continuous_input") C->set_initial_message_mode(LDPC_Code_LUT::CONT);
        else if (initial_message_mode == "from_quantized_channel_llrs") C->set_initial_message_mode(LDPC_Code_LUT::QCHA);
        else throw std::runtime_error("LDPC_BER_Sim_LUT::load(): Initial message mode undefined!");
        if (rand_seed == save_codec) {
            const fs::path sub = fs::path(results_path) / gen_filename();
            fs::create_directories(sub);
            C->save_code((sub / "lut_codec.it").string());
        }
    } else {        // load a codec, :522-545
        const fs::path codec_path = fs::path(codes_path) / codec_filename;
        if (!fs::exists(codec_path)) throw std::runtime_error("Codec file" + codec_path.string() + " does not exist!");
        G.reset(new LDPC_Generator_Systematic());
        C.reset(new LDPC_Code_LUT(codec_path.string(), G.get()));
        C->set_device(device);
        decoder_set = true;
        encoder_set = G->is_initialized();
        if (!encoder_set) G.reset();
        codeword_length = C->get_nvar(); dataword_length = C->get_ninfo(); code_rate = C->get_rate();
        max_iter = C->get_nrof_iterations(); Nq_Msg = C->Nq_Msg; reuse_lut = C->reuse_vec;
        C->set_exit_conditions(max_iter, parity_check_iter, parity_check_iter);
    }
    results = LDPC_BER_Sim_Results(codeword_length, codeword_length - dataword_length);
}

std::string LDPC_BER_Sim_LUT::gen_filename() const {
    if (!decoder_set) throw std::logic_error("LDPC_BER_Sim_LUT::gen_filename(): Decoder has not been set!");
    std::ostringstream fn;
    fn << results_prefix << "_N" << codeword_length << "_R" << code_rate << "_maxIter" << max_iter << "_zcw" << (int)zero_codeword
       << "_frames" << Nframes << (min_lut ? "_minLUT" : "_LUT") << custom_name;
    return fn.str();
}

void LDPC_BER_Sim_LUT::sim_batch(double snr, int snr_index, int64_t frame0, int B, FrameStats *stats) {
    const double N0 = std::pow(10.0, -snr / 10.0) / C->get_rate();      // :248
    const int N = codeword_length, K = dataword_length;
    const int mode = C->get_initial_message_mode() == LDPC_Code_LUT::QCHA ? 1 : 0;
    const ChannelCellTable cells = make_channel_cells(N0, C->get_qb_Cha(), C->get_qb_Msg(), mode, C->get_Nq_Cha_2_Nq_Msg_map());
    const lutldpc_channel_cells view = cells.view();
    const uint64_t seed = (uint64_t)(int64_t)(rand_seed + rand_seed_offset);      // RNG_reset(rand_seed + rand_seed_offset), :129
    if (!zero_codeword && !encoder_set) throw std::runtime_error("Non zero codewords require the encoder to be set!");
    std::vector<unsigned char> codewords;
    const uint8_t *cwp = nullptr;
    if (!zero_codeword) {
        bvec info((size_t)K), cw;
        codewords.resize((size_t)B * N);
        for (int i = 0; i < B; i++) {
            random_info_bits(seed, (uint32_t)snr_index, (uint64_t)(frame0 + i), K, info.data());
            C->encode(info, cw);
            std::memcpy(&codewords[(size_t)i * N], cw.data(), (size_t)N);
        }
        cwp = codewords.data();
    }
    if (decoder_output_verbosity > 1) {
        // output_verbosity 2 / 3: the message dumps of lut_decode (src/LDPC_Code_LUT.cpp:292-298,311-317,331-337) on top of the
        // stimuli.  A debug path: the labels of the batch come back to the host (same sampler, same frames), lut_decode_batch
        // prints the dumps frame after frame (LDPC_Code_LUT::lut_decode_batch_dump), the counters are taken on the host.
        std::vector<uint8_t> cha((size_t)B * N), msg((size_t)B * N), bits((size_t)B * N);
        std::vector<int32_t> its((size_t)B);
        if (lutldpc_decoder_sample_labels(C->device_handle(), &view, seed, (uint32_t)snr_index, (uint64_t)frame0, B, cwp, cha.data(), msg.data()) != LUTLDPC_OK)
            throw std::runtime_error(std::string("LDPC_BER_Sim_LUT::sim_snr_point(): ") + lutldpc_last_error());
        const int nzc = C->get_Nq_Cha() / 2;
        for (int i = 0; i < B; i++) {
            C->lut_decode_batch(&cha[(size_t)i * N], &msg[(size_t)i * N], 1, &bits[(size_t)i * N], &its[(size_t)i]);     // (prints the dumps)
            C->print_stimuli(&cha[(size_t)i * N], &bits[(size_t)i * N]);
            int be = 0, ue = 0;
            for (int v = 0; v < N; v++) {
                const int sent = cwp ? cwp[(size_t)i * N + v] : 0;
                ue += ((cha[(size_t)i * N + v] < nzc) ? 1 : 0) != sent;
                if (v < K) be += bits[(size_t)i * N + v] != sent;
            }
            stats[i] = FrameStats{its[(size_t)i], be ? 1 : 0, be, ue};
        }
        return;
    }
    // output_verbosity > 0: the reference prints every frame's labels and decided bits (src/LDPC_Code_LUT.cpp:228-238)
    std::vector<uint8_t> cha_dump, bits_dump;
    if (decoder_output_verbosity > 0) { cha_dump.resize((size_t)B * N); bits_dump.resize((size_t)B * N); }
    if (lutldpc_decoder_sim_batch(C->device_handle(), &view, seed, (uint32_t)snr_index, (uint64_t)frame0, B, cwp, K, reinterpret_cast<int32_t *>(stats),
                                  cha_dump.empty() ? nullptr : cha_dump.data(), bits_dump.empty() ? nullptr : bits_dump.data()) != LUTLDPC_OK)
        throw std::runtime_error(std::string("LDPC_BER_Sim_LUT::sim_snr_point(): ") + lutldpc_last_error());
    for (int i = 0; i < B && !cha_dump.empty(); i++) C->print_stimuli(&cha_dump[(size_t)i * N], &bits_dump[(size_t)i * N]);
}

bool LDPC_BER_Sim::sim_snr_point(double snr, int snr_index) {
    const int N = codeword_length, K = dataword_length;
    SnrPointCounters c;
    const int64_t total = (int64_t)Nframes;
    int64_t f = 0;
    int batch = std::min(512, batch_frames);            // one full frame group of nibble rows (a 256-frame start ran half-empty groups)
    std::vector<FrameStats> stats;
    while (f < total) {
        const int B = (int)std::min<int64_t>(batch, total - f);
        stats.assign((size_t)B, FrameStats{});
        sim_batch(snr, snr_index, f, B, stats.data());
        f += B;
        if (accumulate_in_order(stats.data(), B, K, Nfers, c)) break;
        batch = std::min(batch * 4, batch_frames);
    }
    const double ber = c.databits ? (double)c.data_bit_errors / (double)c.databits : 0.0;
    const double uber = c.frames ? (double)c.uncoded_bit_errors / ((double)c.frames * N) : 0.0;
    const double fer = c.frames ? (double)c.frame_errors / (double)c.frames : 0.0;
    if (!quiet)
        std::cout << "SNR = " << snr << "  Simulated " << c.frames << " frames and " << c.databits << " data bits. "
                  << "Obtained " << c.data_bit_errors << " data bit errors. " << " Data BER: " << ber << " Uncoded BER: " << uber
                  << " FER: " << fer << std::endl << std::flush;
    results.add_snr_point(snr, c.frames, c.databits, c.frame_errors, c.data_bit_errors, c.uncoded_bit_errors);
    return ber < ber_min || fer < fer_min;       // :307
}

// ------------------------------------------------------------------ LDPC_BER_Sim_BP
void awgn_llr_frames(uint64_t seed, uint32_t stream, uint64_t frame0, int B, int N, double N0, const unsigned char *codewords, double *llr, int32_t *uncoded) {
    const double sigma = std::sqrt(N0 / 2), two_pi = 6.283185307179586476925286766559;
    for (int i = 0; i < B; i++) {
        const uint64_t frame = frame0 + (uint64_t)i;
        int32_t unc = 0;
        for (int p = 0; p < (N + 1) / 2; p++) {
            // one Philox block per bit pair: two 53-bit uniforms -> one Box-Muller pair
            uint32_t c[4] = {(uint32_t)frame, (uint32_t)(frame >> 32), (uint32_t)p, stream | 0x40000000u};
            uint32_t ka = (uint32_t)seed, kb = (uint32_t)(seed >> 32);
            for (int r = 0; r < 10; r++) { philox_round(c, ka, kb); ka += 0x9E3779B9u; kb += 0xBB67AE85u; }
            const double u1 = ((double)((((uint64_t)c[0] << 32) | c[1]) >> 11) + 1.0) * (1.0 / 9007199254740992.0);    // (0, 1]
            const double u2 = (double)((((uint64_t)c[2] << 32) | c[3]) >> 11) * (1.0 / 9007199254740992.0);            // [0, 1)
            const double rad = std::sqrt(-2.0 * std::log(u1));
            double sn, cs;
            ::sincos(two_pi * u2, &sn, &cs);       // ONE libm entry point for both (a compiler may or may not merge sin + cos itself)
            const double z[2] = {rad * cs, rad * sn};
            for (int k = 0; k < 2 && 2 * p + k < N; k++) {
                const int v = 2 * p + k;
                const int bit = codewords ? codewords[(size_t)i * N + v] : 0;
                const double x = (bit ? -1.0 : 1.0) + sigma * z[k];
                llr[(size_t)i * N + v] = 4.0 * x / N0;
                unc += ((x < 0) ? 1 : 0) != bit;
            }
        }
        uncoded[i] = unc;
    }
}

LDPC_BER_Sim_BP::LDPC_BER_Sim_BP(const std::string &params, const std::string &base) : LDPC_BER_Sim(params, base) {
    Ini ini(params);
    max_iter = ini.get("BP.max_iter", 30);
    llr_calc_d1 = ini.get("BP.qllr_scale_res", 12);
    llr_calc_d2 = ini.get("BP.qllr_table_size", 300);
    llr_calc_d3 = ini.get("BP.qllr_spacing_res", 7);
    llr_calc_d4 = ini.get("BP.qllr_total_res", 28);           // 8 * sizeof(QLLR) - 4
}

LDPC_BER_Sim_BP::~LDPC_BER_Sim_BP() { if (dec) lutldpc_bp_destroy(dec); }

void LDPC_BER_Sim_BP::load() {
    if (!codec_filename.empty())
        throw std::runtime_error("LDPC_BER_Sim::load(): loading an IT++ bp_codec.it file is not supported (the IT++ fork's file layout is absent); give LDPC.parity_filename");
    const fs::path parity_path = fs::path(codes_path) / (parity_filename + ".alist");
    if (!fs::exists(parity_path)) throw std::runtime_error("Parity file" + parity_path.string() + " does not exist!");
    H.reset(new LDPC_Parity(parity_path.string()));
    if (!zero_codeword) {
        const fs::path gen_path = fs::path(codes_path) / (parity_filename + ".gen.it");
        G.reset(new LDPC_Generator_Systematic());
        if (fs::exists(gen_path)) G->load(gen_path.string());
        if (!G->is_initialized()) {
            G->construct(H.get());
            if (save_permuted) {
                H->save_alist(parity_path.string());
                { it_file_writer f(gen_path.string()); f.write("Fileversion", 2); f.close(); }
                G->save(gen_path.string());
            }
        }
        encoder_set = true;
    }
    // the decoder sees the (possibly column-permuted) matrix through the same index arrays as the LUT decoder
    LDPC_Code_LUT graph;
    graph.set_device(-1);
    graph.set_code_with_rank(H.get(), nullptr, H->get_ncheck());
    if (lutldpc_bp_create(graph.get_nvar(), graph.get_nchk(), graph.get_dv_vec().data(), graph.get_dc_vec().data(), graph.get_cn_msg_idx().data(),
                          llr_calc_d1, llr_calc_d2, llr_calc_d3, llr_calc_d4, device, &dec) != LUTLDPC_OK)
        throw std::runtime_error(std::string("LDPC_BER_Sim::load(): ") + lutldpc_last_error());
    lutldpc_bp_set_exit_conditions(dec, max_iter, parity_check_iter, parity_check_iter);     // :199
    codeword_length = H->get_nvar();
    dataword_length = H->get_nvar() - H->get_ncheck();         // itpp::LDPC_Code::get_ninfo()
    code_rate = 1.0 - (double)H->get_ncheck() / H->get_nvar(); // itpp::LDPC_Code::get_rate()
    decoder_set = true;
    results = LDPC_BER_Sim_Results(codeword_length, codeword_length - dataword_length);
}

void LDPC_BER_Sim_BP::sim_batch(double snr, int snr_index, int64_t frame0, int B, FrameStats *stats) {
    const int N = codeword_length, K = dataword_length;
    const double N0 = std::pow(10.0, -snr / 10.0) / code_rate;      // :248
    const uint64_t seed = (uint64_t)(int64_t)(rand_seed + rand_seed_offset);
    if (!zero_codeword && !encoder_set) throw std::runtime_error("Non zero codewords require the encoder to be set!");
    std::vector<unsigned char> codewords;
    if (!zero_codeword) {
        bvec info((size_t)K), cw;
        codewords.resize((size_t)B * N);
        for (int i = 0; i < B; i++) {
            random_info_bits(seed, (uint32_t)snr_index, (uint64_t)(frame0 + i), K, info.data());
            G->encode(info, cw);
            std::memcpy(&codewords[(size_t)i * N], cw.data(), (size_t)N);
        }
    }
    std::vector<double> llr((size_t)B * N);
    std::vector<int32_t> unc((size_t)B), iters((size_t)B);
    std::vector<uint8_t> bits((size_t)B * N);
    // the noise of a batch is generated on all host cores (frames are Philox-addressed: any split gives the same samples)
    const unsigned nthr = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nthr; t++)
        pool.emplace_back([&, t] {
            const int b0 = (int)((int64_t)B * t / nthr), b1 = (int)((int64_t)B * (t + 1) / nthr);
            if (b1 > b0) awgn_llr_frames(seed, (uint32_t)snr_index, (uint64_t)(frame0 + b0), b1 - b0, N, N0, codewords.empty() ? nullptr : &codewords[(size_t)b0 * N],
                                         &llr[(size_t)b0 * N], &unc[(size_t)b0]);
        });
    for (auto &th : pool) th.join();
    if (lutldpc_bp_decode_llr_batch(dec, llr.data(), B, bits.data(), iters.data(), nullptr) != LUTLDPC_OK)
        throw std::runtime_error(std::string("LDPC_BER_Sim::sim_snr_point(): ") + lutldpc_last_error());
    for (int i = 0; i < B; i++) {
        int be = 0;
        for (int v = 0; v < K; v++) be += bits[(size_t)i * N + v] != (codewords.empty() ? 0 : codewords[(size_t)i * N + v]);
        stats[i] = FrameStats{iters[(size_t)i], be ? 1 : 0, be, unc[(size_t)i]};
    }
}

// ------------------------------------------------------------------ ber_sim main
int ber_sim_main(int argc, char **argv) {
    int seed = 0, device = 0, lanes = 0;
    std::vector<int> devices;
    std::string exchange = "auto";
    std::string base_dir = fs::current_path().string(), custom_name, params;
    bool have_params = false;
    auto usage = [] {
        std::cout << "OPTIONS:\n"
                     "  -b [ --basedir ] arg      paths in params files are relative to this directory. Default: current direcroy\n"
                     "  -c [ --custom-name ] arg  append this string at the end of the results file name\n"
                     "  -h [ --help ]             produce help message\n"
                     "  -p [ --params ] arg       input parameter file\n"
                     "  -s [ --seed ] arg (=0)    random seed\n"
                     "  -d [ --device ] arg (=0)  HIP device ordinal, a list 0,1,2,3 or `all`: the frames of every SNR point are sharded\n"
                     "                            over the devices, counters over RCCL (build-side option)\n"
                     "  --lanes arg (=0)          host threads (simulation objects, streams) per device: batches of one overlap the\n"
                     "                            sampler / transfers of the other; 1 = the plain synchronous loop; 0 = two, or one\n"
                     "                            where a single batch occupies the whole device (N x batch_frames >= 2^30)\n"
                     "  --exchange arg (=auto)    rccl | host | auto: how the counters of the ranks are combined\n";
    };
    try {
        for (int i = 1; i < argc; i++) {
            const std::string a = argv[i];
            auto value = [&](const char *shortn, const char *longn, std::string &out) {
                const std::string lp = std::string(longn) + "=";
                if (a == shortn || a == longn) { if (i + 1 >= argc) throw std::runtime_error("missing value for " + a); out = argv[++i]; return true; }
                if (a.rfind(lp, 0) == 0) { out = a.substr(lp.size()); return true; }
                return false;
            };
            std::string v;
            if (a == "-h" || a == "--help") { usage(); return 0; }
            else if (value("-b", "--basedir", v)) base_dir = v;
            else if (value("-c", "--custom-name", v)) custom_name = v;
            else if (value("-p", "--params", v)) { params = v; have_params = true; }
            else if (value("-s", "--seed", v)) seed = std::stoi(v);
            else if (value("-d", "--device", v)) {
                devices.clear();
                if (v == "all") {
                    const int n = lutldpc_device_count();
                    for (int i = 0; i < n; i++) devices.push_back(i);
                    if (devices.empty()) throw std::runtime_error("no HIP device visible");
                } else {
                    std::stringstream ss(v);
                    std::string tok;
                    while (std::getline(ss, tok, ',')) if (!tok.empty()) devices.push_back(std::stoi(tok));
                    if (devices.empty()) throw std::runtime_error("empty device list");
                }
                device = devices[0];
            }
            else if (value("--lanes", "--lanes", v)) lanes = std::stoi(v);
            else if (value("--exchange", "--exchange", v)) exchange = v;
            else throw std::runtime_error("unrecognised option '" + a + "'");
        }
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    if (!have_params) { std::cout << "No input parameters specified. To learn more, use the --help option.\n"; return 0; }
    if (fs::path(base_dir).is_relative()) { std::cout << "Base directory must be specified as absolut path\n"; return 0; }
    try {
        std::string params_path = fs::path(params).is_relative() ? (fs::path(base_dir) / params).string() : params;
        if (!fs::exists(params_path)) throw std::runtime_error("Parameter file" + params_path + " does not exist!");
        Ini ini(params_path);
        const std::string codec_type = ini.get("Sim.codec_type", "none");
        if (devices.empty()) devices.push_back(device);
        if (const char *e = std::getenv("LUTLDPC_LANES")) lanes = std::atoi(e);
        if (lanes == 0) {
            // auto: two lanes hide the sampler, the transfers and the host prefix of one batch behind the decode of the other -- worth
            // 30 % on the short codes; a batch of a long code (DVB-S2 at 32768 frames: 6.8 GB of rows, 120 ms of HBM-bound launches)
            // leaves nothing to overlap with, and the second lane's set-up (its own decoder, generated kernels, placement search)
            // costs a 1e6-frame point 5 % (profiles/r03_config4_one_gpu_ber_sim.txt: 4.11 s against 4.31 s)
            lanes = 2;
            const std::string codes_dir = ini.get("Sim.codes_dir", "codes"), parity = ini.get("LDPC.parity_filename", "");
            std::ifstream al(fs::path(join(base_dir, codes_dir)) / (parity + ".alist"));
            long long n_var = 0;
            if (al && (al >> n_var) && n_var > 0 && n_var * (long long)ini.get("Sim.batch_frames", 32768) >= (1ll << 30)) lanes = 1;
        }
        const bool is_bp = !(ini.has_section("LUT") || codec_type == "LUT");
        if (is_bp && devices.size() == 1) lanes = 1;               // (the [BP] comparison decoder draws its noise on all host cores already)
        // output_verbosity > 0 prints every frame's stimuli / message dumps to std::cout in frame order (src/LDPC_Code_LUT.cpp:228-238,
        // 292-337): that text is only meaningful from ONE thread
        if (ini.get("LUT.output_verbosity", 0) > 0) { lanes = 1; if (devices.size() > 1) devices.resize(1); }
        // The placement search of the row buffers (decoder.hip: place_rows, 0.1-0.4 s per batch size for +2-6 % of the streaming
        // kernels' rate) pays for itself after some eight seconds of decoding at that size: a rank that sees fewer than 64
        // full batches of its largest SNR point runs without (config 4 on eight GPUs: four batches per rank).  LUTLDPC_PLACE
        // set by the user wins; the variable is read when a decoder is created.
        {
            const double per_rank = ini.get("Sim.Nframes", 1e2) / ((double)devices.size() * (double)std::max(lanes, 1) * (double)std::max(1, ini.get("Sim.batch_frames", 32768)));
            if (per_rank < 64.0) setenv("LUTLDPC_PLACE", "0", 0);
        }
        if (devices.size() > 1 || lanes > 1)
            return ber_sim_run_multi(params_path, base_dir, seed, custom_name, devices, lanes, exchange, false);
        std::unique_ptr<LDPC_BER_Sim> sim;
        if (ini.has_section("LUT") || codec_type == "LUT") sim.reset(new LDPC_BER_Sim_LUT(params_path, base_dir));
        else if (ini.has_section("BP") || codec_type == "BP") sim.reset(new LDPC_BER_Sim_BP(params_path, base_dir));
        else throw std::runtime_error("You must specify the type of decoder in the params file ([LUT] section or Sim.codec_type)");
        sim->rand_seed = seed;
        sim->device = device;
        sim->append_custom_name(custom_name);
        sim->load();
        sim->run();
        sim->save();
    } catch (const std::exception &e) {
        std::cerr << "ber_sim: " << e.what() << "\n";
        return 1;
    }
    return 0;
}

}  // namespace lut_ldpc
