/*
 * lut_ldpc_bp.h -- C-ABI of the [BP] comparison decoder (liblut_ldpc_amd.so), the second back-end behind the ber_sim driver.
 *
 * The reference compares its LUT decoders against IT++'s belief-propagation decoder: LDPC_BER_Sim (src/LDPC_BER_Sim.cpp:157-244)
 * builds an itpp::LDPC_Code, calls set_exit_conditions(max_iter, parity_check_iter, parity_check_iter) and
 * set_llrcalc(LLR_calc_unit(d1, d2, d3, d4)) (:199-200) and decodes frame by frame through C->decode(softbits) (:281).
 * That decoder lives in the forked IT++ (mmeidlinger/itpp, branch lut_ldpc), which is ABSENT from the reference tree (empty
 * submodule): its arithmetic cannot be read, only the published IT++ 4.3.1 algorithm can be restated.  PARITY UNPINNED: what
 * is specified below is this build's own statement of that algorithm; the GPU kernels and the oracle (oracle/or_bp.c) agree
 * bit for bit with EACH OTHER, nothing ties them to the fork.
 *
 * Arithmetic (integers throughout; "QLLR" = int32):
 *   to_qllr(l)   = clip(floor(0.5 + 2^d1 * l), +-QMAX),   QMAX = 2^(d4-1) - 1          (d4: BP.qllr_total_res, default 28)
 *   T[i]         = to_qllr(log(1 + exp(-i * 2^(d3-d1)))),  i = 0 .. d2-1                (d2: BP.qllr_table_size, 0 = min-sum)
 *   logexp(x)    = (x >> d3) >= d2 ? 0 : T[x >> d3]                                      (x >= 0, no interpolation)
 *   boxplus(a,b) = sgn(a) sgn(b) min(|a|,|b|) + logexp(|a+b|) - logexp(|a-b|)            (a > 0 counts as positive, 0 as negative)
 * Decoding (flooding, IT++ convention: LLR > 0 <=> bit 0):
 *   start        : if pisc and every check is satisfied by the signs of the input -> return 0, output = input
 *                  every edge of variable node v carries LLRin[v]
 *   iteration k  : check node of degree 2: the two messages are swapped;
 *                  degree 3..6: the association orders IT++ 4.3.1 spells out (boxplus with the table correction is not
 *                  associative), with mXY = boxplus of inputs X..Y built pairwise:
 *                    3: out0 = (m1,m2), out1 = (m0,m2), out2 = (m0,m1)
 *                    4: m01, m23;  out0 = (m1,m23), out1 = (m0,m23), out2 = (m01,m3), out3 = (m01,m2)
 *                    5: m01, m02 = (m01,m2), m34, m24 = (m2,m34);  out0 = (m1,m24), out1 = (m0,m24), out2 = (m01,m34),
 *                       out3 = (m02,m4), out4 = (m02,m3)
 *                    6: m01, m23, m45, m03 = (m01,m23), m25 = (m23,m45), m0145 = (m01,m45);  out0 = (m1,m25), out1 = (m0,m25),
 *                       out2 = (m0145,m3), out3 = (m0145,m2), out4 = (m03,m5), out5 = (m03,m4)
 *                  (stated from the published IT++ 4.3.1 source as recalled -- the source is not available here to check);
 *                  degree >= 7: left / right partial boxplus sums ml[i] = boxplus(ml[i-1], m[i]),
 *                  mr[i] = boxplus(mr[i-1], m[n-1-i]); out[0] = mr[n-2], out[n-1] = ml[n-2], out[i] = boxplus(ml[i-1], mr[n-2-i]);
 *                  variable node: s = LLRin[v] + sum of incoming; LLRout[v] = clip(s); message to check c = clip(s - incoming from c);
 *                  if psc and every check is satisfied by the signs of LLRout -> return k
 *   end          : return -max_iters  (IT++ reports success only through the syndrome check: without psc every frame returns
 *                  -max_iters)
 * Decided bits: LLRout < 0.  Frames are independent; a batch is bit-identical to decoding frame by frame.
 */
#ifndef LUT_LDPC_BP_H
#define LUT_LDPC_BP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lutldpc_bp_decoder lutldpc_bp_decoder;

/* Graph as in lutldpc_decoder_create (include/lut_ldpc_hip.h): dv[nvar], dc[nchk], cn_msg_idx[E] = VN-major edge ids check by
 * check.  d1..d4 = BP.qllr_scale_res, qllr_table_size, qllr_spacing_res, qllr_total_res (src/LDPC_BER_Sim.cpp:75-78).
 * device < 0: host-only handle (table construction, to_qllr), refuses to decode.  Returns LUTLDPC_OK / a negative code
 * (include/lut_ldpc_hip.h), text through lutldpc_last_error(). */
int lutldpc_bp_create(int nvar, int nchk, const int32_t *dv, const int32_t *dc, const int32_t *cn_msg_idx,
                      int d1, int d2, int d3, int d4, int device, lutldpc_bp_decoder **out);
int lutldpc_bp_destroy(lutldpc_bp_decoder *d);
/* LDPC_Code::set_exit_conditions(max_iters, syndr_check_each_iter, syndr_check_at_start) */
int lutldpc_bp_set_exit_conditions(lutldpc_bp_decoder *d, int max_iters, int psc, int pisc);
/* The log-exp table T (d2 entries) as built at creation; returns d2. */
int lutldpc_bp_logexp_table(lutldpc_bp_decoder *d, int32_t *out, int cap);
/* LDPC_Code::decode(const vec &llr, bvec &bits) for B frames: llr[B*nvar] double (host), out_bits[B*nvar] = LLRout < 0,
 * out_iters[B] = the return value of bp_decode, out_qllr (optional, B*nvar) = LLRout. */
int lutldpc_bp_decode_llr_batch(lutldpc_bp_decoder *d, const double *llr, int B, uint8_t *out_bits, int32_t *out_iters, int32_t *out_qllr);
/* LDPC_Code::bp_decode(QLLRvec) for B frames on quantised input (host). */
int lutldpc_bp_decode_qllr_batch(lutldpc_bp_decoder *d, const int32_t *qllr, int B, uint8_t *out_bits, int32_t *out_iters, int32_t *out_qllr);

#ifdef __cplusplus
}
#endif
#endif
