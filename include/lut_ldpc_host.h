/*
 * lut_ldpc_host.h -- C-ABI over the C++ host mirror of the reference's LDPC_Code_LUT /
 * LDPC_BER_Sim_LUT (lut_ldpc_amd/csrc/host), so that non-C++ callers (the Python tests and
 * bench.py, a cgo/JNI binding, ...) can drive the same objects ber_sim uses.
 *
 * A `lutldpc_codec` bundles what LDPC_BER_Sim_LUT::load builds (src/LDPC_BER_Sim.cpp:434-550):
 * the parity-check matrix (alist), optionally the systematic generator, and the
 * LDPC_Code_LUT object whose decode path runs on the MI355X (include/lut_ldpc_hip.h).
 * Return codes and lutldpc_last_error() as in lut_ldpc_hip.h.
 */
#ifndef LUT_LDPC_HOST_H
#define LUT_LDPC_HOST_H

#include "lut_ldpc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lutldpc_codec lutldpc_codec;

/* new LDPC_Parity(alist) [+ LDPC_Generator_Systematic(H)] + new LDPC_Code_LUT(H, G)
 * (src/LDPC_BER_Sim.cpp:443-476).  known_rank > 0 skips the GF(2) rank computation.
 * device = -1 gives a host-only object (LUT design, files) that cannot decode. */
int lutldpc_codec_create(const char *alist_path, int with_generator, int known_rank, int device, lutldpc_codec **out);
/* LDPC_Code_LUT(filename, G): load a lut_codec.it (src/LDPC_Code_LUT.cpp:568-641) */
int lutldpc_codec_load(const char *codec_path, int device, lutldpc_codec **out);
int lutldpc_codec_save(lutldpc_codec *c, const char *codec_path);     /* save_code, :643-697 */
int lutldpc_codec_destroy(lutldpc_codec *c);

/* LDPC_Code_LUT::design_luts with the empirical ensemble of H (src/LDPC_Code_LUT.cpp:699-746,
 * src/LDPC_BER_Sim.cpp:487-495).  allow_degree_one: build-side extension for degree-1 VNs. */
int lutldpc_codec_design_luts(lutldpc_codec *c, const char *tree_method, int min_lut, double sigma2, int max_iters,
                              const uint8_t *reuse_vec, int Nq_Cha, const int32_t *Nq_Msg, int allow_degree_one,
                              double *sigma_out);
/* 1 when the last design_luts was read from the design cache (environment LUTLDPC_DESIGN_CACHE=<directory>: designs are
 * kept there keyed by a hash of every design input, trees in the reference's own text serialisation,
 * src/LUT_Tree.cpp:847-865), 0 when density evolution ran. */
int lutldpc_codec_design_from_cache(lutldpc_codec *c);
int lutldpc_codec_set_exit_conditions(lutldpc_codec *c, int max_iters, int psc, int pisc);
int lutldpc_codec_set_initial_message_mode(lutldpc_codec *c, int mode);   /* 0 CONT, 1 QCHA */
int lutldpc_codec_set_output_verbosity(lutldpc_codec *c, int level);

/* getters */
int lutldpc_codec_dims(lutldpc_codec *c, int32_t *nvar, int32_t *nchk, int32_t *nedges, int32_t *rank);
int lutldpc_codec_graph(lutldpc_codec *c, int32_t *dv, int32_t *dc, int32_t *cn_msg_idx);
/* text of the tree arrays; returns the length needed (incl. NUL); copies when cap suffices */
int64_t lutldpc_codec_var_trees_txt(lutldpc_codec *c, char *buf, int64_t cap);
int64_t lutldpc_codec_chk_trees_txt(lutldpc_codec *c, char *buf, int64_t cap);
int lutldpc_codec_qb(lutldpc_codec *c, int which /*0 Cha, 1 Msg*/, double *out, int cap);   /* returns count */
int lutldpc_codec_cha2msg_map(lutldpc_codec *c, int32_t *out, int cap);                       /* returns count */
double lutldpc_codec_rate(lutldpc_codec *c);
/* the underlying HIP decoder (created on first use; NULL + error when there is no device) */
lutldpc_decoder *lutldpc_codec_decoder(lutldpc_codec *c);

/* batched LDPC_Code_LUT::decode / lut_decode (host buffers, all nvar bits per frame) */
int lutldpc_codec_decode_llr_batch(lutldpc_codec *c, const double *llr, int B, uint8_t *bits, int32_t *iters);
int lutldpc_codec_lut_decode_batch(lutldpc_codec *c, const uint8_t *cha, const uint8_t *msg0, int B, uint8_t *bits, int32_t *iters);
/* lut_decode with LDPC_Code_LUT::set_output_verbosity(level), level 2 or 3: the message dumps of src/LDPC_Code_LUT.cpp:292-298,
 * 311-317, 331-337 as text, frame after frame (what the reference streams to std::cout).  Returns the bytes needed for the text
 * including the terminating NUL (buf may be NULL / too small: nothing is copied then), or a negative error code. */
int64_t lutldpc_codec_lut_decode_dump(lutldpc_codec *c, const uint8_t *cha, const uint8_t *msg0, int B, int level, uint8_t *bits, int32_t *iters,
                                      char *buf, int64_t cap);
/* LDPC_Code_LUT::encode: info[K] -> codeword[nvar] (needs with_generator) */
int lutldpc_codec_encode(lutldpc_codec *c, const uint8_t *info, uint8_t *codeword);

/* One batch of the Monte-Carlo loop (front end + decode + error counting on the device) at Eb/N0
 * snr_db: frames frame0 .. frame0+B-1 of stream `stream` (= index of the SNR point).  The channel
 * cells are derived from the codec's boundaries; with zero_codeword = 0 the data bits come from the
 * Philox stream and are encoded on the host (needs with_generator).  stats: host [B*4] int32 as in
 * lutldpc_decoder_sim_batch. */
int lutldpc_codec_sim_batch(lutldpc_codec *c, double snr_db, uint64_t seed, uint32_t stream, uint64_t frame0, int B,
                            int zero_codeword, int32_t *stats);
/* the labels (and sent codewords, may be NULL) of those frames, for tests */
int lutldpc_codec_sample_labels(lutldpc_codec *c, double snr_db, uint64_t seed, uint32_t stream, uint64_t frame0, int B,
                                int zero_codeword, uint8_t *cha, uint8_t *msg0, uint8_t *codewords);
/* the channel cell table for that SNR (see lut_ldpc_hip.h): returns n_cells; arrays need 72 entries */
int lutldpc_codec_channel_cells(lutldpc_codec *c, double snr_db, uint64_t *thr, uint8_t *cha, uint8_t *msg, uint8_t *neg,
                                uint8_t *cha_m, uint8_t *msg_m);

/* ber_sim: load() + run() [+ save()] of LDPC_BER_Sim_LUT for a parameter file (prog/ber_sim.cpp).
 * Returns the number of SNR points written to snr[cap] / counters[cap*5]
 * ({frames, data bits, frame errors, data bit errors, uncoded bit errors} per point), <0 on error. */
int lutldpc_ber_sim_run(const char *params_path, const char *base_dir, int seed, const char *custom_name, int device,
                        int save_results, int quiet, double *snr, int64_t *counters, int cap);
/* The same simulation object driven step by step, for callers that shard the frames of an SNR point
 * over several processes / GPUs (lut_ldpc_amd/ber_sim.py): create = constructor + load(). */
typedef struct lutldpc_bersim lutldpc_bersim;
int lutldpc_bersim_create(const char *params_path, const char *base_dir, int seed, const char *custom_name, int device,
                          lutldpc_bersim **out);
int lutldpc_bersim_destroy(lutldpc_bersim *s);
/* info[8] = {n_snr, Nframes, Nfers, nvar, ninfo, max_iter, zero_codeword, batch_frames}; limits[2] = {ber_min, fer_min} */
int lutldpc_bersim_info(lutldpc_bersim *s, int64_t *info, double *limits, double *snr, int snr_cap);
/* frames frame0..frame0+B-1 of SNR point snr_index -> stats[B*4] (see lutldpc_decoder_sim_batch) */
int lutldpc_bersim_batch(lutldpc_bersim *s, int snr_index, int64_t frame0, int B, int32_t *stats);
/* results.add_snr_point / save_runtime + save() */
int lutldpc_bersim_add_point(lutldpc_bersim *s, double snr, const int64_t *counters5);
int lutldpc_bersim_save(lutldpc_bersim *s, double runtime_s);
/* the results-file path save() writes (copied into buf when cap suffices; returns needed length) */
int64_t lutldpc_bersim_results_path(lutldpc_bersim *s, char *buf, int64_t cap);

/* the command line itself */
int lutldpc_ber_sim_main(int argc, char **argv);
/* the results file of src/LDPC_BER_Sim.cpp:342-362 for given counters (n SNR points x {frames, data bits, frame errors, data bit errors,
 * uncoded bit errors}): the writer behind ber_sim's result files, exposed for its byte-level known-answer test */
int lutldpc_selftest_write_results_it(const char *path, const double *snr, const int64_t *counters, int n, int nvar, int nchk, double runtime);

/* LDPC_DE_LUT::bisec_search for an ensemble given by its active degrees (prog/de_sim.cpp:137-260
 * set-up: auto trees, no reuse, uniform resolution).  Returns the bisection count, <0 on error. */
int lutldpc_de_threshold(const int32_t *dl, const double *lam, int nl, const int32_t *dr, const double *rho, int nr,
                         int qbits_cha, int qbits_msg, int maxiter_de, int min_lut, const char *tree_mode, const char *strategy,
                         double thr_min, double thr_prec, double Pe_max, int maxiter_bisec, int max_ni_de_iters,
                         double LLR_max, int Nq_fine, double *thr_out);

/* The [BP] path's front end (LDPC_BER_Sim_BP::sim_batch): BPSK over AWGN in double precision on the host, Philox-addressed per
 * (seed, stream = SNR index, frame, bit pair), Box-Muller; llr[B*N] = 4 x / N0 (src/LDPC_BER_Sim.cpp:270-279), uncoded[B] =
 * slicer errors (:283).  codewords: B*N sent bits or NULL (all-zero codeword). */
int lutldpc_awgn_llr(uint64_t seed, uint32_t stream, uint64_t frame0, int B, int N, double N0, const uint8_t *codewords, double *llr, int32_t *uncoded);

#ifdef __cplusplus
}
#endif
#endif
