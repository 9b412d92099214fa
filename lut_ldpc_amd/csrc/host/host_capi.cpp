// host_capi.cpp -- C-ABI of include/lut_ldpc_host.h over the C++ host classes.
#include "lut_ldpc_host.h"
#include "ldpc_code_lut.hpp"
#include "ber_sim_driver.hpp"
#include "ini.hpp"

#include <cmath>
#include <vector>

#include <cstring>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>

using namespace lut_ldpc;

struct lutldpc_codec {
    std::unique_ptr<LDPC_Parity> H;
    std::unique_ptr<LDPC_Generator_Systematic> G;
    std::unique_ptr<LDPC_Code_LUT> C;
};

extern "C" void lutldpc_set_last_error(const char *msg);   // decoder.hip

namespace {
template <class F>
int guarded(F &&f) {
    try { return f(); }
    catch (const std::invalid_argument &e) { lutldpc_set_last_error(e.what()); return LUTLDPC_ERR_ARG; }
    catch (const std::logic_error &e) { lutldpc_set_last_error(e.what()); return LUTLDPC_ERR_STATE; }
    catch (const std::exception &e) { lutldpc_set_last_error(e.what()); return LUTLDPC_ERR_ARG; }
}
int64_t copy_out(const std::string &s, char *buf, int64_t cap) {
    const int64_t need = (int64_t)s.size() + 1;
    if (buf && cap >= need) std::memcpy(buf, s.c_str(), (size_t)need);
    return need;
}
}  // namespace

extern "C" {

int lutldpc_codec_create(const char *alist_path, int with_generator, int known_rank, int device, lutldpc_codec **out) {
    return guarded([&] {
        if (!alist_path || !out) throw std::invalid_argument("NULL argument");
        std::unique_ptr<lutldpc_codec> c(new lutldpc_codec);
        c->H.reset(new LDPC_Parity(alist_path));
        if (with_generator) c->G.reset(new LDPC_Generator_Systematic(c->H.get()));
        c->C.reset(new LDPC_Code_LUT());
        c->C->set_device(device);
        // known_rank > 0: rank(H) supplied by the caller (dense elimination of a 64800-column
        // code takes minutes; the reference pays that price at every start-up)
        c->C->set_code_with_rank(c->H.get(), c->G.get(), known_rank);
        *out = c.release();
        return LUTLDPC_OK;
    });
}

int lutldpc_codec_load(const char *codec_path, int device, lutldpc_codec **out) {
    return guarded([&] {
        if (!codec_path || !out) throw std::invalid_argument("NULL argument");
        std::unique_ptr<lutldpc_codec> c(new lutldpc_codec);
        c->G.reset(new LDPC_Generator_Systematic());
        c->C.reset(new LDPC_Code_LUT(std::string(codec_path), c->G.get()));
        c->C->set_device(device);
        *out = c.release();
        return LUTLDPC_OK;
    });
}

int lutldpc_codec_save(lutldpc_codec *c, const char *path) {
    return guarded([&] { if (!c || !path) throw std::invalid_argument("NULL argument"); c->C->save_code(path); return LUTLDPC_OK; });
}

int lutldpc_codec_destroy(lutldpc_codec *c) { delete c; return LUTLDPC_OK; }

int lutldpc_codec_design_from_cache(lutldpc_codec *c) { return (c && c->C && c->C->design_came_from_cache()) ? 1 : 0; }

int lutldpc_codec_design_luts(lutldpc_codec *c, const char *tree_method, int min_lut, double sigma2, int max_iters, const uint8_t *reuse_vec,
                              int Nq_Cha, const int32_t *Nq_Msg, int allow_degree_one, double *sigma_out) {
    return guarded([&] {
        if (!c || !tree_method || !reuse_vec || !Nq_Msg || max_iters < 1) throw std::invalid_argument("NULL / bad argument");
        if (!c->H) throw std::logic_error("design_luts needs the parity-check matrix (codec was loaded from a codec file)");
        const bvec reuse(reuse_vec, reuse_vec + max_iters);
        const ivec nq(Nq_Msg, Nq_Msg + max_iters);
        const double s = c->C->design_luts(tree_method, get_empirical_ensemble(*c->H), min_lut != 0, sigma2, max_iters, reuse, Nq_Cha, nq, allow_degree_one != 0);
        if (sigma_out) *sigma_out = s;
        return LUTLDPC_OK;
    });
}

int lutldpc_codec_set_exit_conditions(lutldpc_codec *c, int max_iters, int psc, int pisc) {
    return guarded([&] { if (!c) throw std::invalid_argument("NULL codec"); c->C->set_exit_conditions(max_iters, psc != 0, pisc != 0); return LUTLDPC_OK; });
}
int lutldpc_codec_set_initial_message_mode(lutldpc_codec *c, int mode) {
    return guarded([&] {
        if (!c || mode < 0 || mode > 1) throw std::invalid_argument("mode must be 0 (CONT) or 1 (QCHA)");
        c->C->set_initial_message_mode(mode ? LDPC_Code_LUT::QCHA : LDPC_Code_LUT::CONT);
        return LUTLDPC_OK;
    });
}
int lutldpc_codec_set_output_verbosity(lutldpc_codec *c, int level) {
    return guarded([&] { if (!c) throw std::invalid_argument("NULL codec"); c->C->set_output_verbosity(level); return LUTLDPC_OK; });
}

int lutldpc_codec_dims(lutldpc_codec *c, int32_t *nvar, int32_t *nchk, int32_t *nedges, int32_t *rank) {
    return guarded([&] {
        if (!c) throw std::invalid_argument("NULL codec");
        if (nvar) *nvar = c->C->get_nvar();
        if (nchk) *nchk = c->C->get_nchk();
        if (nedges) *nedges = (int32_t)c->C->get_cn_msg_idx().size();
        if (rank) *rank = c->C->get_nchk_lin_indep();
        return LUTLDPC_OK;
    });
}
int lutldpc_codec_graph(lutldpc_codec *c, int32_t *dv, int32_t *dc, int32_t *cn_msg_idx) {
    return guarded([&] {
        if (!c || !dv || !dc || !cn_msg_idx) throw std::invalid_argument("NULL argument");
        std::memcpy(dv, c->C->get_dv_vec().data(), sizeof(int32_t) * c->C->get_dv_vec().size());
        std::memcpy(dc, c->C->get_dc_vec().data(), sizeof(int32_t) * c->C->get_dc_vec().size());
        std::memcpy(cn_msg_idx, c->C->get_cn_msg_idx().data(), sizeof(int32_t) * c->C->get_cn_msg_idx().size());
        return LUTLDPC_OK;
    });
}
int64_t lutldpc_codec_var_trees_txt(lutldpc_codec *c, char *buf, int64_t cap) { return c ? copy_out(to_string(c->C->get_var_trees()), buf, cap) : 0; }
int64_t lutldpc_codec_chk_trees_txt(lutldpc_codec *c, char *buf, int64_t cap) { return c ? copy_out(to_string(c->C->get_chk_trees()), buf, cap) : 0; }
int lutldpc_codec_qb(lutldpc_codec *c, int which, double *out, int cap) {
    if (!c) return 0;
    const vec &q = which == 0 ? c->C->get_qb_Cha() : c->C->get_qb_Msg();
    if (out && cap >= (int)q.size()) std::memcpy(out, q.data(), sizeof(double) * q.size());
    return (int)q.size();
}
int lutldpc_codec_cha2msg_map(lutldpc_codec *c, int32_t *out, int cap) {
    if (!c) return 0;
    const ivec &m = c->C->get_Nq_Cha_2_Nq_Msg_map();
    if (out && cap >= (int)m.size()) std::memcpy(out, m.data(), sizeof(int32_t) * m.size());
    return (int)m.size();
}
double lutldpc_codec_rate(lutldpc_codec *c) { return c ? c->C->get_rate() : 0.0; }

lutldpc_decoder *lutldpc_codec_decoder(lutldpc_codec *c) {
    lutldpc_decoder *d = nullptr;
    guarded([&] { if (!c) throw std::invalid_argument("NULL codec"); d = c->C->device_handle(); return LUTLDPC_OK; });
    return d;
}

int lutldpc_codec_decode_llr_batch(lutldpc_codec *c, const double *llr, int B, uint8_t *bits, int32_t *iters) {
    return guarded([&] { if (!c || !llr || !bits || !iters) throw std::invalid_argument("NULL argument"); c->C->decode_batch(llr, B, bits, iters); return LUTLDPC_OK; });
}
// lut_decode with the message dumps of output_verbosity = level as text (what the reference streams to std::cout); returns the
// number of bytes the text needs including the terminating NUL (call with buf = NULL first), or a negative error code
int64_t lutldpc_codec_lut_decode_dump(lutldpc_codec *c, const uint8_t *cha, const uint8_t *msg0, int B, int level, uint8_t *bits, int32_t *iters, char *buf, int64_t cap) {
    int64_t need = 0;
    const int rc = guarded([&] {
        if (!c || !cha || !msg0 || !bits || !iters) throw std::invalid_argument("NULL argument");
        std::ostringstream os;
        c->C->lut_decode_batch_dump(cha, msg0, B, bits, iters, level, os);
        const std::string t = os.str();
        need = (int64_t)t.size() + 1;
        if (buf && cap >= need) std::memcpy(buf, t.c_str(), (size_t)need);
        return LUTLDPC_OK;
    });
    return rc == LUTLDPC_OK ? need : rc;
}

int lutldpc_codec_lut_decode_batch(lutldpc_codec *c, const uint8_t *cha, const uint8_t *msg0, int B, uint8_t *bits, int32_t *iters) {
    return guarded([&] { if (!c || !cha || !msg0 || !bits || !iters) throw std::invalid_argument("NULL argument"); c->C->lut_decode_batch(cha, msg0, B, bits, iters); return LUTLDPC_OK; });
}
int lutldpc_codec_encode(lutldpc_codec *c, const uint8_t *info, uint8_t *codeword) {
    return guarded([&] {
        if (!c || !info || !codeword) throw std::invalid_argument("NULL argument");
        bvec in(info, info + c->C->get_ninfo()), out;
        c->C->encode(in, out);
        std::memcpy(codeword, out.data(), out.size());
        return LUTLDPC_OK;
    });
}

}  // extern "C"

namespace {
ChannelCellTable cells_for(lutldpc_codec *c, double snr_db) {
    const double N0 = std::pow(10.0, -snr_db / 10.0) / c->C->get_rate();
    const int mode = c->C->get_initial_message_mode() == LDPC_Code_LUT::QCHA ? 1 : 0;
    return make_channel_cells(N0, c->C->get_qb_Cha(), c->C->get_qb_Msg(), mode, c->C->get_Nq_Cha_2_Nq_Msg_map());
}
void make_codewords(lutldpc_codec *c, uint64_t seed, uint32_t stream, uint64_t frame0, int B, std::vector<unsigned char> &cw) {
    const int N = c->C->get_nvar(), K = c->C->get_ninfo();
    cw.resize((size_t)B * N);
    bvec info((size_t)K), one;
    for (int i = 0; i < B; i++) {
        random_info_bits(seed, stream, frame0 + (uint64_t)i, K, info.data());
        c->C->encode(info, one);
        std::memcpy(&cw[(size_t)i * N], one.data(), (size_t)N);
    }
}
}  // namespace

extern "C" {

int lutldpc_codec_sim_batch(lutldpc_codec *c, double snr_db, uint64_t seed, uint32_t stream, uint64_t frame0, int B, int zero_codeword, int32_t *stats) {
    return guarded([&] {
        if (!c || !stats || B <= 0) throw std::invalid_argument("NULL / bad argument");
        const ChannelCellTable cells = cells_for(c, snr_db);
        const lutldpc_channel_cells view = cells.view();
        std::vector<unsigned char> cw;
        if (!zero_codeword) make_codewords(c, seed, stream, frame0, B, cw);
        return lutldpc_decoder_sim_batch(c->C->device_handle(), &view, seed, stream, frame0, B, zero_codeword ? nullptr : cw.data(), c->C->get_ninfo(), stats, nullptr, nullptr);
    });
}

int lutldpc_codec_sample_labels(lutldpc_codec *c, double snr_db, uint64_t seed, uint32_t stream, uint64_t frame0, int B, int zero_codeword,
                                uint8_t *cha, uint8_t *msg0, uint8_t *codewords) {
    return guarded([&] {
        if (!c || !cha || !msg0 || B <= 0) throw std::invalid_argument("NULL / bad argument");
        const ChannelCellTable cells = cells_for(c, snr_db);
        const lutldpc_channel_cells view = cells.view();
        std::vector<unsigned char> cw;
        if (!zero_codeword) make_codewords(c, seed, stream, frame0, B, cw);
        if (codewords) { if (zero_codeword) std::memset(codewords, 0, (size_t)B * c->C->get_nvar()); else std::memcpy(codewords, cw.data(), cw.size()); }
        return lutldpc_decoder_sample_labels(c->C->device_handle(), &view, seed, stream, frame0, B, zero_codeword ? nullptr : cw.data(), cha, msg0);
    });
}

int lutldpc_codec_channel_cells(lutldpc_codec *c, double snr_db, uint64_t *thr, uint8_t *cha, uint8_t *msg, uint8_t *neg, uint8_t *cha_m, uint8_t *msg_m) {
    int n = 0;
    int rc = guarded([&] {
        if (!c || !thr || !cha || !msg || !neg || !cha_m || !msg_m) throw std::invalid_argument("NULL argument");
        const ChannelCellTable t = cells_for(c, snr_db);
        n = (int)t.cha.size();
        if (n > 72) throw std::invalid_argument("more than 72 cells");
        std::memcpy(thr, t.thr.data(), sizeof(uint64_t) * t.thr.size());
        std::memcpy(cha, t.cha.data(), (size_t)n); std::memcpy(msg, t.msg.data(), (size_t)n); std::memcpy(neg, t.neg.data(), (size_t)n);
        std::memcpy(cha_m, t.cha_m.data(), (size_t)n); std::memcpy(msg_m, t.msg_m.data(), (size_t)n);
        return LUTLDPC_OK;
    });
    return rc == LUTLDPC_OK ? n : rc;
}

// prog/ber_sim.cpp:128-142: the section present in the parameter file (or Sim.codec_type) picks the simulation class
static std::unique_ptr<LDPC_BER_Sim> make_sim(const char *params_path, const char *base_dir) {
    Ini ini(params_path);
    const std::string codec_type = ini.get("Sim.codec_type", "none");
    if (ini.has_section("LUT") || codec_type == "LUT") return std::unique_ptr<LDPC_BER_Sim>(new LDPC_BER_Sim_LUT(params_path, base_dir));
    if (ini.has_section("BP") || codec_type == "BP") return std::unique_ptr<LDPC_BER_Sim>(new LDPC_BER_Sim_BP(params_path, base_dir));
    throw std::runtime_error("You must specify the type of decoder in the params file ([LUT] or [BP] section, or Sim.codec_type)");
}

int lutldpc_ber_sim_run(const char *params_path, const char *base_dir, int seed, const char *custom_name, int device, int save_results, int quiet,
                        double *snr, int64_t *counters, int cap) {
    int n = 0;
    int rc = guarded([&] {
        if (!params_path || !base_dir) throw std::invalid_argument("NULL argument");
        std::unique_ptr<LDPC_BER_Sim> simp = make_sim(params_path, base_dir);
        LDPC_BER_Sim &sim = *simp;
        sim.rand_seed = seed; sim.device = device; sim.quiet = quiet != 0;
        if (custom_name) sim.append_custom_name(custom_name);
        sim.load();
        sim.run();
        if (save_results) sim.save();
        const LDPC_BER_Sim_Results &r = sim.results;
        n = (int)r.sim_SNRdB.size();
        for (int i = 0; i < n && i < cap; i++) {
            if (snr) snr[i] = r.sim_SNRdB[(size_t)i];
            if (counters) {
                counters[i * 5 + 0] = r.sim_Nframes[(size_t)i]; counters[i * 5 + 1] = r.sim_Ndatabits[(size_t)i];
                counters[i * 5 + 2] = r.sim_frame_errors[(size_t)i]; counters[i * 5 + 3] = r.sim_data_bit_errors[(size_t)i];
                counters[i * 5 + 4] = r.sim_uncoded_bit_errors[(size_t)i];
            }
        }
        return LUTLDPC_OK;
    });
    return rc == LUTLDPC_OK ? n : rc;
}

int lutldpc_ber_sim_main(int argc, char **argv) { return ber_sim_main(argc, argv); }

// LDPC_BER_Sim_Results::save (src/LDPC_BER_Sim.cpp:342-362) on given numbers: n SNR points with five counters each.  Used by the
// byte-level known-answer test of the .it writer (the expected bytes are assembled by hand from scripts/itsave.m / itload.m).
int lutldpc_selftest_write_results_it(const char *path, const double *snr, const int64_t *counters, int n, int nvar, int nchk, double runtime) {
    return guarded([&] {
        if (!path || (n > 0 && (!snr || !counters))) throw std::invalid_argument("NULL argument");
        LDPC_BER_Sim_Results r(nvar, nchk);
        for (int i = 0; i < n; i++) r.add_snr_point(snr[i], counters[i * 5], counters[i * 5 + 1], counters[i * 5 + 2], counters[i * 5 + 3], counters[i * 5 + 4]);
        r.save_runtime(runtime);
        r.write_itfile(path);
        return LUTLDPC_OK;
    });
}

int lutldpc_awgn_llr(uint64_t seed, uint32_t stream, uint64_t frame0, int B, int N, double N0, const uint8_t *codewords, double *llr, int32_t *uncoded) {
    return guarded([&] {
        if (!llr || !uncoded || B <= 0 || N <= 0 || !(N0 > 0)) throw std::invalid_argument("NULL / bad argument");
        awgn_llr_frames(seed, stream, frame0, B, N, N0, codewords, llr, uncoded);
        return LUTLDPC_OK;
    });
}

struct lutldpc_bersim { std::unique_ptr<LDPC_BER_Sim> sim; };

int lutldpc_bersim_create(const char *params_path, const char *base_dir, int seed, const char *custom_name, int device, lutldpc_bersim **out) {
    return guarded([&] {
        if (!params_path || !base_dir || !out) throw std::invalid_argument("NULL argument");
        std::unique_ptr<lutldpc_bersim> s(new lutldpc_bersim);
        s->sim = make_sim(params_path, base_dir);
        s->sim->rand_seed = seed; s->sim->device = device; s->sim->quiet = true;
        if (custom_name) s->sim->append_custom_name(custom_name);
        s->sim->load();
        *out = s.release();
        return LUTLDPC_OK;
    });
}
int lutldpc_bersim_destroy(lutldpc_bersim *s) { delete s; return LUTLDPC_OK; }
int lutldpc_bersim_info(lutldpc_bersim *s, int64_t *info, double *limits, double *snr, int snr_cap) {
    return guarded([&] {
        if (!s || !info || !limits) throw std::invalid_argument("NULL argument");
        LDPC_BER_Sim &m = *s->sim;
        info[0] = (int64_t)m.SNRdB.size(); info[1] = (int64_t)m.Nframes; info[2] = m.Nfers; info[3] = m.get_codeword_length();
        info[4] = m.get_dataword_length(); info[5] = m.max_iter; info[6] = m.zero_codeword ? 1 : 0; info[7] = m.batch_frames;
        limits[0] = m.ber_min; limits[1] = m.fer_min;
        if (snr) for (int i = 0; i < (int)m.SNRdB.size() && i < snr_cap; i++) snr[i] = m.SNRdB[(size_t)i];
        return LUTLDPC_OK;
    });
}
int lutldpc_bersim_batch(lutldpc_bersim *s, int snr_index, int64_t frame0, int B, int32_t *stats) {
    return guarded([&] {
        if (!s || !stats || B <= 0) throw std::invalid_argument("NULL / bad argument");
        if (snr_index < 0 || snr_index >= (int)s->sim->SNRdB.size()) throw std::invalid_argument("snr_index out of range");
        s->sim->sim_batch(s->sim->SNRdB[(size_t)snr_index], snr_index, frame0, B, reinterpret_cast<FrameStats *>(stats));
        return LUTLDPC_OK;
    });
}
int lutldpc_bersim_add_point(lutldpc_bersim *s, double snr, const int64_t *c) {
    return guarded([&] {
        if (!s || !c) throw std::invalid_argument("NULL argument");
        s->sim->results.add_snr_point(snr, c[0], c[1], c[2], c[3], c[4]);
        return LUTLDPC_OK;
    });
}
int lutldpc_bersim_save(lutldpc_bersim *s, double runtime_s) {
    return guarded([&] { if (!s) throw std::invalid_argument("NULL argument"); s->sim->results.save_runtime(runtime_s); s->sim->save(); return LUTLDPC_OK; });
}
int64_t lutldpc_bersim_results_path(lutldpc_bersim *s, char *buf, int64_t cap) {
    if (!s) return 0;
    std::string p;
    if (guarded([&] { p = s->sim->results_file_path(); return LUTLDPC_OK; }) != LUTLDPC_OK) return 0;
    return copy_out(p, buf, cap);
}

int lutldpc_de_threshold(const int32_t *dl, const double *lam, int nl, const int32_t *dr, const double *rho, int nr, int qbits_cha, int qbits_msg,
                         int maxiter_de, int min_lut, const char *tree_mode, const char *strategy, double thr_min, double thr_prec, double Pe_max,
                         int maxiter_bisec, int max_ni_de_iters, double LLR_max, int Nq_fine, double *thr_out) {
    int iters = -1;
    int rc = guarded([&] {
        if (!dl || !lam || !dr || !rho || !tree_mode || !strategy || !thr_out) throw std::invalid_argument("NULL argument");
        LDPC_Ensemble ens(ivec(dl, dl + nl), vec(lam, lam + nl), ivec(dr, dr + nr), vec(rho, rho + nr));
        const ivec Nq((size_t)maxiter_de, 1 << qbits_msg);
        LUT_Tree_Array var_t, chk_t;
        get_lut_tree_templates(tree_mode, ens, Nq, 1 << qbits_cha, min_lut != 0, var_t, chk_t);
        LDPC_DE_LUT de(ens, 1 << qbits_cha, Nq, maxiter_de, var_t, chk_t, bvec(), thr_prec, Pe_max, maxiter_bisec, LLR_max, Nq_fine, strategy);
        de.set_bisec_window(thr_min, rate_to_shannon_thr(ens.get_rate()));
        de.set_exit_conditions(maxiter_de, maxiter_bisec, max_ni_de_iters, Pe_max, thr_prec);
        iters = de.bisec_search(*thr_out);
        return LUTLDPC_OK;
    });
    return rc == LUTLDPC_OK ? iters : rc;
}

}  // extern "C"
