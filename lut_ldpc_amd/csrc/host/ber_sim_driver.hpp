// ber_sim_driver.hpp -- LDPC_BER_Sim / LDPC_BER_Sim_LUT / LDPC_BER_Sim_Results: the Monte-Carlo
// driver of the reference (src/LDPC_BER_Sim.hpp:57-215) with the frame loop batched on the GPU.
//
// Same parameter file, same public fields and methods (load / run / save / sim_snr_point /
// gen_filename / append_custom_name), same results file.  What changes underneath:
//   * frames are simulated in batches through lutldpc_decoder_sim_batch (sampler + decode + error
//     counting on the device); the stop rule of sim_snr_point (src/LDPC_BER_Sim.cpp:289: stop once
//     MORE than Nfers frame errors were seen) is applied to the per-frame results in frame order,
//     so the counters are exactly what a frame-by-frame loop over the same frames would give;
//   * random numbers are Philox-addressed per (seed, SNR index, frame), see kernels_frontend.hpp;
//   * the [BP] branch (IT++'s own BP decoder, src/LDPC_BER_Sim.cpp:157-244) is LDPC_BER_Sim_BP: the comparison decoder of
//     include/lut_ldpc_bp.h on the device behind a host AWGN front end (PARITY UNPINNED: the IT++ fork is absent).
#pragma once
#include "ldpc_code_lut.hpp"
#include "lut_ldpc_hip.h"
#include "lut_ldpc_bp.h"

#include <cstdint>
#include <functional>
#include <memory>
#include <optional>
#include <string>
#include <vector>

namespace lut_ldpc {

// src/LDPC_BER_Sim.hpp:57-88
class LDPC_BER_Sim_Results {
public:
    LDPC_BER_Sim_Results() = default;
    LDPC_BER_Sim_Results(int nvar, int nchk) : ldpc_nvar(nvar), ldpc_nchk(nchk), ldpc_code_rate(1.0 - (double)nchk / nvar) {}
    void add_snr_point(double snr, int64_t frames, int64_t databits, int64_t frame_errors, int64_t data_bit_errors, int64_t uncoded_bit_errors);
    void save_runtime(double t) { runtime = t; }
    void write_itfile(const std::string &filename) const;       // src/LDPC_BER_Sim.cpp:342-362
    std::vector<double> sim_SNRdB;
    std::vector<int64_t> sim_Nframes, sim_Ndatabits, sim_frame_errors, sim_data_bit_errors, sim_uncoded_bit_errors;
    int ldpc_nvar = 0, ldpc_nchk = 0;
    double ldpc_code_rate = 0, runtime = 0;
};

// The received-value partition handed to the device sampler (see kernels_frontend.hpp).
struct ChannelCellTable {
    std::vector<uint64_t> thr;
    std::vector<uint8_t> cha, msg, neg, cha_m, msg_m;
    lutldpc_channel_cells view() const;
};
// N0: noise spectral density (variance N0/2 per dimension), boundaries in the LLR domain; mode as
// LDPC_Code_LUT::initial_message_mode_t (QCHA derives the message label from the channel label)
ChannelCellTable make_channel_cells(double N0, const vec &qb_Cha, const vec &qb_Msg, int initial_message_mode, const ivec &cha2msg_map);

// Philox-addressed data bits of frame f (zero_codeword = false), shared with the tests
void random_info_bits(uint64_t seed, uint32_t stream, uint64_t frame, int K, unsigned char *out);

// per-frame result of one simulated frame: {lut_decode code, frame error, data bit errors, uncoded errors}
struct FrameStats { int32_t iters, frame_error, bit_errors, uncoded_errors; };
struct SnrPointCounters { int64_t frames = 0, databits = 0, frame_errors = 0, data_bit_errors = 0, uncoded_bit_errors = 0; };

// Apply the stop rule of sim_snr_point to frames given in order; returns true when the frame loop
// must stop (frame errors > Nfers) after consuming a prefix of `stats`.
bool accumulate_in_order(const FrameStats *stats, int n, int K, int64_t Nfers, SnrPointCounters &c);

class LDPC_BER_Sim {
public:
    LDPC_BER_Sim(const std::string &params_file_path, const std::string &base_dir_path);   // src/LDPC_BER_Sim.cpp:42-102
    virtual ~LDPC_BER_Sim() = default;
    virtual void load() = 0;
    virtual void run();                                          // :121-155
    virtual void save();                                         // :317-340
    std::string results_file_path() const;                       // <results_dir>/<gen_filename>/<gen_filename>_rseedNNNN.it
    virtual bool sim_snr_point(double snr, int snr_index);       // :246-311 (batched through sim_batch)
    // frames frame0 .. frame0+B-1 of SNR point `snr_index` (any order, any sharding): the body of the
    // frame loop, :262-286.  The stop rule is applied by the caller (accumulate_in_order).
    virtual void sim_batch(double snr, int snr_index, int64_t frame0, int B, FrameStats *stats) = 0;
    int get_codeword_length() const { return codeword_length; }
    int get_dataword_length() const { return dataword_length; }
    virtual std::string gen_filename() const;                    // :104-115
    void append_custom_name(const std::string &ext) { custom_name += ext; }

    // public parameter fields, as in the reference
    int rand_seed = 0;
    vec SNRdB;
    double Nframes = 1e5;
    int Nfers = 20;
    double ber_min = 1e-7, fer_min = 1e-5;
    int rand_seed_offset = 0, save_codec = 0;
    std::string custom_name, results_prefix = "RES", results_dir = "results", codes_dir = "codes", codec_filename, parity_filename;
    bool zero_codeword = true, save_permuted = false, parity_check_iter = true;
    int max_iter = 30;
    LDPC_BER_Sim_Results results;
    // build-side knobs
    int device = 0;
    int batch_frames = 32768;        // frames per device call (upper bound; INI key Sim.batch_frames)
    bool quiet = false;

protected:
    std::string params_file_path, base_dir, codes_path, results_path;
    int codeword_length = 0, dataword_length = 0;
    bool decoder_set = false, encoder_set = false;
    double code_rate = 0;
};

class LDPC_BER_Sim_LUT : public LDPC_BER_Sim {
public:
    LDPC_BER_Sim_LUT(const std::string &params_file_path, const std::string &base_dir_path);   // src/LDPC_BER_Sim.cpp:376-430
    void load() override;                                        // :434-550
    void sim_batch(double snr, int snr_index, int64_t frame0, int B, FrameStats *stats) override;
    std::string gen_filename() const override;                   // :553-568
    LDPC_Code_LUT *codec() { return C.get(); }

    std::optional<double> design_thr, design_SNRdB;
    int decoder_output_verbosity = 0;
    std::string initial_message_mode = "from_continuous_input", tree_mode = "auto_bin_balanced", trees_dir = "trees", trees_filename;
    int Nq_Cha = 16;
    ivec Nq_Msg;
    bool min_lut = true;
    bvec reuse_lut;
    bool allow_degree_one = false;   // LUT.allow_degree_one (build-side extension, default off = reference behaviour)
    int known_rank = 0;              // LDPC.known_rank (build-side: skip the GF(2) rank computation)

private:
    std::string trees_path;
    std::unique_ptr<LDPC_Parity> H;
    std::unique_ptr<LDPC_Generator_Systematic> G;
    std::unique_ptr<LDPC_Code_LUT> C;
};

// BPSK over AWGN in double precision on the host, Philox-addressed per (seed, stream = SNR index, frame, bit pair), Box-Muller:
// llr[i*N + v] = 4 x / N0 with x = (1 - 2 bit) + sqrt(N0/2) z  (src/LDPC_BER_Sim.cpp:270-279, BPSK::demodulate_soft_bits);
// codewords: B*N sent bits or NULL for the all-zero codeword; uncoded[i] = slicer errors of frame i (:283).
void awgn_llr_frames(uint64_t seed, uint32_t stream, uint64_t frame0, int B, int N, double N0, const unsigned char *codewords,
                     double *llr, int32_t *uncoded);

// The reference's base class as it is used for the [BP] section (src/LDPC_BER_Sim.cpp:157-244): itpp::LDPC_Code with
// LLR_calc_unit(qllr_scale_res, qllr_table_size, qllr_spacing_res, qllr_total_res) -> lutldpc_bp_decoder.
class LDPC_BER_Sim_BP : public LDPC_BER_Sim {
public:
    LDPC_BER_Sim_BP(const std::string &params_file_path, const std::string &base_dir_path);   // :42-102
    ~LDPC_BER_Sim_BP() override;
    void load() override;                                        // :157-244
    void sim_batch(double snr, int snr_index, int64_t frame0, int B, FrameStats *stats) override;
    int llr_calc_d1 = 12, llr_calc_d2 = 300, llr_calc_d3 = 7, llr_calc_d4 = 28;                // :75-78
    lutldpc_bp_decoder *decoder() { return dec; }

private:
    std::unique_ptr<LDPC_Parity> H;
    std::unique_ptr<LDPC_Generator_Systematic> G;
    lutldpc_bp_decoder *dec = nullptr;
};

// One run of a parameter file over several devices of a node (ber_sim_multi.cpp): `lanes` host threads per device, each with its
// own simulation object / decoder handle / stream; frames sharded by rank, counters exchanged over RCCL ("rccl"), on the host
// ("host": ranks may share a device) or whichever applies ("auto").  Writes the same result file as the single-device run.
int ber_sim_run_multi(const std::string &params_path, const std::string &base_dir, int seed, const std::string &custom_name,
                      const std::vector<int> &devices, int lanes, const std::string &exchange_mode, bool quiet);

// ber_sim's main (prog/ber_sim.cpp:46-160): returns the process exit code
int ber_sim_main(int argc, char **argv);

}  // namespace lut_ldpc
