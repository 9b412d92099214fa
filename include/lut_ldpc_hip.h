/*
 * lut_ldpc_hip.h -- C-ABI of the MI355X LUT-LDPC decode path (liblut_ldpc_amd.so).
 *
 * This is the drop-in boundary for the hot path of mmeidlinger/lut_ldpc.  The reference has
 * no FFI layer: the seam is the C++ virtual call `C->decode(softbits)` made once per frame by
 * LDPC_BER_Sim::sim_snr_point (src/LDPC_BER_Sim.cpp:281), landing in
 * LDPC_Code_LUT::decode / lut_decode (src/LDPC_Code_LUT.cpp:204,259).  The entry points below
 * are what an LDPC_Code_LUT-shaped class binds instead (see INTEGRATION.md for the stub);
 * the class shipped in lut_ldpc_amd/csrc/host does exactly that.
 *
 * Conventions: plain pointers and sizes, int return codes (0 = ok, <0 = error, text via
 * lutldpc_last_error()), no exceptions cross the boundary.  A decoder handle owns its device
 * memory and one HIP stream and is NOT re-entrant (like the reference object, whose
 * lut_decode mutates the member `msgs`, src/LDPC_Code_LUT.hpp:311): use one handle per host
 * thread.  All label arrays are uint8 and frame-major: element [f*nvar + v].
 */
#ifndef LUT_LDPC_HIP_H
#define LUT_LDPC_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LUTLDPC_OK               0
#define LUTLDPC_ERR_ARG         -1   /* bad argument / inconsistent sizes                */
#define LUTLDPC_ERR_PARSE       -2   /* malformed tree text                               */
#define LUTLDPC_ERR_UNSUPPORTED -3   /* e.g. node fan-in or table beyond device limits    */
#define LUTLDPC_ERR_HIP         -4   /* HIP runtime failure (no device, OOM, launch)      */
#define LUTLDPC_ERR_STATE       -5   /* call order (e.g. decode before set-up)            */

typedef struct lutldpc_decoder lutldpc_decoder;

/* Text of the last error on the calling thread ("" if none). */
const char *lutldpc_last_error(void);

/* Library / build identification, e.g. "lut_ldpc_amd 0.1 gfx950". */
const char *lutldpc_version(void);

/* Number of visible HIP devices (0 without a GPU; never fails). */
int lutldpc_device_count(void);

/*
 * Create a decoder.  Replaces the state built by LDPC_Code_LUT::set_code ->
 * decoder_parameterization (src/LDPC_Code_LUT.cpp:471-541) and set_trees (:120-169).
 *
 *   nvar, nchk      code dimensions
 *   dv[nvar]        variable-node degrees   (reference member dv_vec)
 *   dc[nchk]        check-node degrees      (reference member dc_vec)
 *   cn_msg_idx[E]   for each check, in order, the VN-major edge ids of its edges
 *                   (reference member cn_msg_idx, :507-527); E = sum(dv) = sum(dc)
 *   Nq_Cha          channel alphabet size
 *   Nq_Msg[max_iters], reuse_vec[max_iters]   message alphabets / LUT reuse pattern
 *   min_lut         1: min-sum check update (chk_update_minsum, :355), 0: CHKTREE LUTs
 *   var_trees_txt   Array<Array<LUT_Tree>> in the reference's own text serialisation
 *                   (src/LUT_Tree.cpp:847-865, the `var_tree_string` of lut_codec.it,
 *                   src/LDPC_Code_LUT.cpp:677-678): [tree set][degree class]
 *   chk_trees_txt   same for check trees; NULL or "" when min_lut
 *   device          HIP device ordinal
 * The exit conditions default to the reference constructor's (max_iters, psc=1, pisc=0).
 */
int lutldpc_decoder_create(int nvar, int nchk, const int32_t *dv, const int32_t *dc,
                           const int32_t *cn_msg_idx,
                           int Nq_Cha, const int32_t *Nq_Msg, const uint8_t *reuse_vec, int max_iters,
                           int min_lut, const char *var_trees_txt, const char *chk_trees_txt,
                           int device, lutldpc_decoder **out);

int lutldpc_decoder_destroy(lutldpc_decoder *d);

/* LDPC_Code_LUT::set_exit_conditions (src/LDPC_Code_LUT.cpp:176-185).  max_iters must not
 * exceed the value given at creation (the tree sets are indexed by iteration). */
int lutldpc_decoder_set_exit_conditions(lutldpc_decoder *d, int max_iters, int psc, int pisc);

/*
 * Batched LDPC_Code_LUT::lut_decode (src/LDPC_Code_LUT.cpp:259-353) on host buffers.
 *   cha[B*nvar]       quantised channel labels   (LLRin_cha)
 *   msg0[B*nvar]      initial message labels     (LLRin_msg)
 *   out_bits[B*nvar]  decoded bits 0/1           (LLRout)
 *   out_iters[B]      the reference's return value per frame: 0, ii+1, +max_iters or -max_iters
 * Frames are independent; results are bit-identical to calling lut_decode frame by frame.
 * Errors instead of the reference's it_assert aborts: B <= 0 or a null pointer -> LUTLDPC_ERR_ARG; a handle without a device
 * -> LUTLDPC_ERR_STATE.  A label outside its alphabet (>= Nq_Cha, >= Nq_Msg[0]) is undefined behaviour in the reference (it
 * indexes the tables unchecked); here it is clamped to the largest label when the rows are built, so it can never reach
 * another frame's lane.
 */
int lutldpc_decoder_decode_batch(lutldpc_decoder *d, const uint8_t *cha, const uint8_t *msg0, int B,
                                 uint8_t *out_bits, int32_t *out_iters);

/* Same with device-resident buffers (same frame-major layout); asynchronous on the decoder's
 * stream unless `sync` is non-zero.  The call runs on the decoder's OWN non-blocking stream (lutldpc_decoder_stream): the input
 * buffers must be complete, and the four buffers must not be memory that another stream is still working on (a stream-ordered
 * allocator may hand out a block whose last users have not finished on THEIR stream) -- synchronise, or make the decoder's
 * stream wait on an event, before the call.  The inputs are read for the whole duration of the decode (the LDS-resident
 * decoder reads them in place), the outputs are written by it. */
int lutldpc_decoder_decode_batch_device(lutldpc_decoder *d, const uint8_t *d_cha, const uint8_t *d_msg0, int B,
                                        uint8_t *d_out_bits, int32_t *d_out_iters, int sync);

/* lut_decode of a small batch with the message dumps of LDPC_Code_LUT::set_output_verbosity(level), level 2 or 3
 * (src/LDPC_Code_LUT.cpp:292-298 initial messages, :311-317 after every check update (level 3), :331-337 after every variable
 * update): trace[dump][B][E] label bytes in the reference's print order, *n_dumps = 1 + max_iters * (level - 1).  Debug path
 * (per-class streaming launches, one copy per dump); which dumps the reference would have PRINTED for a frame follows from its
 * iteration code (a frame that leaves through the exit test returns before the dump of its last variable update). */
int lutldpc_decoder_decode_batch_trace(lutldpc_decoder *d, const uint8_t *cha, const uint8_t *msg0, int B, int level, uint8_t *out_bits,
                                       int32_t *out_iters, uint8_t *trace, int64_t trace_cap, int32_t *n_dumps);

/*
 * Batched LDPC_Code_LUT::decode(const vec&, bvec&) (src/LDPC_Code_LUT.cpp:204-226):
 * quant_nonlin with the boundaries qb_Cha / qb_Msg (src/common.cpp:120-138), initial
 * messages by mode (0 = CONT: quant_nonlin(llr, qb_Msg); 1 = QCHA: cha2msg_map[label]),
 * then lut_decode.  llr[B*nvar] double, host memory.  out_bits holds all nvar bits per
 * frame; the systematic part is the first nvar - rank(H) of them (:225).
 */
int lutldpc_decoder_decode_llr_batch(lutldpc_decoder *d, const double *llr, int B,
                                     const double *qb_Cha, int n_qb_Cha,
                                     const double *qb_Msg, int n_qb_Msg,
                                     int initial_message_mode, const int32_t *cha2msg_map,
                                     uint8_t *out_bits, int32_t *out_iters);

/*
 * Monte-Carlo front end + decode + error counting for B consecutive frames of one SNR point:
 * the body of the frame loop of LDPC_BER_Sim::sim_snr_point (src/LDPC_BER_Sim.cpp:260-286)
 * without the stop rule, which stays with the caller (it needs the frames in order).
 *
 * The channel is described by the partition of the received value into cells (see
 * lut_ldpc_amd/csrc/hip/kernels_frontend.hpp): n_cells cells with cumulative probabilities
 * thr[j] = floor(2^64 * P(cell <= j | bit 0 sent)) for j < n_cells-1 and, per cell, the channel
 * label, initial message label and slicer sign for a sent 0 and the labels of the mirrored cell
 * for a sent 1.  Frame f of the SNR point uses the Philox4x32-10 stream
 * (key = seed, counter = (f_lo, f_hi, bit-pair index, stream)), so any split of the frame range
 * over calls / GPUs generates the same frames.
 *   codewords   host [B*nvar] sent bits, or NULL for the all-zero codeword
 *   K_info      number of leading bits counted as data bits (nvar - rank(H))
 *   frame_stats host [B*4] int32: {lut_decode return value, frame error 0/1, data bit errors,
 *               uncoded (slicer) bit errors over all nvar bits}
 *   cha_out, bits_out  optional host [B*nvar]: the sampled channel labels and the decided bits
 *               (for the "Stimuli Pair" dump of src/LDPC_Code_LUT.cpp:228-238); NULL to skip
 */
typedef struct {
    int32_t n_cells;
    const uint64_t *thr;        /* n_cells-1 */
    const uint8_t *cha_label;   /* n_cells each */
    const uint8_t *msg_label;
    const uint8_t *slicer_neg;
    const uint8_t *cha_label_mirror;
    const uint8_t *msg_label_mirror;
} lutldpc_channel_cells;

int lutldpc_decoder_sim_batch(lutldpc_decoder *d, const lutldpc_channel_cells *cells, uint64_t seed, uint32_t stream,
                              uint64_t frame0, int B, const uint8_t *codewords, int K_info, int32_t *frame_stats,
                              uint8_t *cha_out, uint8_t *bits_out);

/* The labels the sampler would produce for those frames (host, frame-major [B*nvar]); for tests. */
int lutldpc_decoder_sample_labels(lutldpc_decoder *d, const lutldpc_channel_cells *cells, uint64_t seed, uint32_t stream,
                                  uint64_t frame0, int B, const uint8_t *codewords, uint8_t *cha, uint8_t *msg0);

/* The decoder's HIP stream (hipStream_t as void*), for callers that enqueue their own work. */
void *lutldpc_decoder_stream(lutldpc_decoder *d);

/* ---- measurement hooks (bench.py) ------------------------------------------------------- */
/* Kernel kinds for the per-kernel timers. */
#define LUTLDPC_K_CN_PASS   0   /* check-node pass (min-sum or CHKTREE)           */
#define LUTLDPC_K_VN_PASS   1   /* variable-node LUT pass                          */
#define LUTLDPC_K_DECISION  2   /* decision-tree pass                              */
#define LUTLDPC_K_SYNDROME  3   /* parity checks                                   */
#define LUTLDPC_K_LAYOUT    4   /* transposes / edge initialisation / state update */
#define LUTLDPC_K_FRONTEND  5   /* channel sampler + error counting                */
#define LUTLDPC_K_FUSED_PASS 6  /* skewed pipeline: check pass of one half + variable pass of the other */
#define LUTLDPC_K_RESIDENT  7   /* LDS-resident decode: all iterations of a batch in one launch */
#define LUTLDPC_K_COUNT     8

/* When enabled every launch is bracketed by HIP events recorded on the decoder's stream. */
int lutldpc_decoder_set_profiling(lutldpc_decoder *d, int enable);
/* Accumulated since the last reset: total milliseconds and launch count of one kernel kind.
 * Synchronises the stream. */
int lutldpc_decoder_get_profile(lutldpc_decoder *d, int kind, double *total_ms, int64_t *launches);
int lutldpc_decoder_reset_profile(lutldpc_decoder *d);
/* Bytes of device memory held for a batch of B frames (after the first decode of that size). */
int64_t lutldpc_decoder_device_bytes(lutldpc_decoder *d);
/* Algorithmic description of the kernels actually selected, as a JSON string owned by the
 * handle (kernel names, vector width, message bytes b, tile size). */
const char *lutldpc_decoder_describe(lutldpc_decoder *d);

/* ---- self test of the host-side tree compiler (no GPU needed) ---------------------------- */
/*
 * Evaluates the compiled node program of tree [set][cls] of `kind` (0 VARTREE, 1 CHKTREE,
 * 2 DECTREE) for ONE node on the host, exactly as the device interpreter would, so that the
 * CSE / slot allocation can be checked without a GPU.  in[n_in] are the node's inputs
 * (messages, then the channel label for kind 0/2); out[n_out] the results.  This is a
 * verifier of the compile step only -- no decode path uses it.
 */
int lutldpc_selftest_program_eval(lutldpc_decoder *d, int kind, int set, int cls,
                                  const int32_t *in, int n_in, int32_t *out, int n_out);
/* Number of LUT look-ups of that program per node (after sharing) and for the naive walk. */
int lutldpc_selftest_program_stats(lutldpc_decoder *d, int kind, int set, int cls,
                                   int32_t *n_ops, int32_t *n_ops_naive, int32_t *n_slots);

/* The HIP source jit.hpp generates for a variable (kind 0) / CHKTREE check (kind 1) / decision (kind 2) class -- copied into buf when
 * cap suffices; returns the length needed including the terminator, < 0 on error.  With compile != 0 the
 * source is also compiled for gfx950 with hiprtc (no device needed); a failure returns LUTLDPC_ERR_HIP and
 * leaves the compiler log in lutldpc_last_error(). */
int64_t lutldpc_selftest_jit_source(lutldpc_decoder *d, int kind, int set, int cls, char *buf, int64_t cap, int compile);
/* source of the LDS-resident decode kernel for a batch of G frame groups (info: sets per workgroup, threads, LDS bytes) */
int64_t lutldpc_selftest_resident_source(lutldpc_decoder *d, int G, char *buf, int64_t cap, int compile, int32_t *info);

#ifdef __cplusplus
}
#endif
#endif
