"""The configurations of BASELINE.json, in the DEFAULT configuration of the library (nibble rows, skewed two-half
pipeline through pass_fused_kernel, chain fusion, hipGraph replay), against the oracle.  This file runs FIRST in the
GPU suite (alphabetical order), knob variants run last (test_zz_*): a failing debug knob must not hide these.

    C1  params/ber.ini.irregular.example end to end          -> test_20_frontend_gpu.py (needs the Monte-Carlo driver)
    C2  (3,6) N=10000, 4-bit, 50 iterations, batch 4096       -> test_c2_*
    C3  DVB-S2 N=64800, 4-bit, 50 iterations (+ its twin)     -> test_c3_*
    C4  C3 sharded over 8 GPUs                                -> tests/test_sharded_gloo.py (CPU, gloo) + bench.py --gpus N
    C5  (6,32) N=2048, 3-bit, file trees, both check updates  -> test_c5_*

Bit-exact: every decided bit and every iteration code.  The faithful oracle decodes DVB-S2 at ~4 frames/s, so at the
full iteration count a sample of the batch is compared (the failing frames first), and the whole batch through
size-independent properties."""
import numpy as np
import pytest

from helpers import awgn_labels, compare, oracle_codec, product_decoder

pytestmark = pytest.mark.gpu


def _sample(it, n):
    """frames to hand to the oracle: every kind of outcome first (failed, early exit, full count), then the head"""
    idx = list(np.flatnonzero(it < 0)[:n // 3]) + list(np.flatnonzero((it > 0) & (it < it.max()))[:n // 3])
    for f in range(len(it)):
        if len(idx) >= n:
            break
        if f not in idx:
            idx.append(f)
    return np.array(sorted(idx[:n]))


@pytest.mark.parametrize("name,snr,n_oracle", [("dvbs2_q4", 1.3, 24), ("reg36_n10000_q4", 1.9, 48)])
def test_c3_c2_full_iteration_count_default_path(name, snr, n_oracle):
    """C3 / C2 at 50 iterations through the production path: three frame groups (uneven halves, ragged last group),
    decoded three times so that the third call is the hipGraph replay; as shipped (psc = pisc = 1) and fixed work."""
    cd = oracle_codec(name)
    assert cd.max_iters == 50
    dec = product_decoder(cd)
    desc = dec.describe()
    assert desc["pack"] == 2 and desc["skewed_pipeline"] == 1 and desc["use_fast"] == 1, desc
    if name == "dvbs2_q4":
        assert desc["fused_bucket"] == 0 and desc["chain_nodes"] == 29699 and desc["resident"] == 0, desc
    else:
        assert desc["resident"] == 1, desc                      # (3,6) N=10000: 120 KB of edge messages per 8 frames -> decoded out of LDS
    B = 1100
    cha, msg, _ = awgn_labels(cd, B, snr, seed=2026)
    for psc in (True, False):
        dec.set_exit_conditions(50, psc, psc)
        runs = [dec.lut_decode_batch(cha, msg) for _ in range(3)]      # plain launches, capture + replay, replay
        bits, it = runs[0]
        for b, i in runs[1:]:
            assert (i == it).all() and (b == bits).all()
        assert (it > 0).sum() > 0
        if psc:
            assert len(set(it.tolist())) > 3                               # frames leave at different iterations
            # a frame reported as converged satisfies every parity check; the all-zero codeword was sent
            assert bits[it > 0].sum() == 0
        idx = _sample(it, n_oracle)
        cd.set_exit_conditions(50, psc, psc)
        wb, wi = cd.lut_decode_batch(cha[idx], msg[idx])
        assert (wi == it[idx]).all(), (idx[wi != it[idx]][:8], wi[:8], it[idx][:8])
        assert (wb == bits[idx]).all()
        # a batch decodes like its parts and in any frame order (frames are independent)
        b1, i1 = dec.lut_decode_batch(cha[:600], msg[:600])
        b2, i2 = dec.lut_decode_batch(cha[600:], msg[600:])
        assert (np.concatenate([i1, i2]) == it).all() and (np.concatenate([b1, b2]) == bits).all()
        perm = np.random.default_rng(7).permutation(B)
        bp, ip = dec.lut_decode_batch(cha[perm], msg[perm])
        assert (ip == it[perm]).all() and (bp == bits[perm]).all()
    dec.close()


def test_c2_batch_4096():
    """BASELINE config 2 at its full size (N=10000, 50 iterations, batch 4096 = eight frame groups)."""
    cd = oracle_codec("reg36_n10000_q4")
    dec = product_decoder(cd)
    B = 4096
    cha, msg, _ = awgn_labels(cd, B, 1.8, seed=42)
    dec.set_exit_conditions(50, True, True)
    bits, it = dec.lut_decode_batch(cha, msg)
    ok = it > 0
    assert ok.mean() > 0.9
    assert bits[ok].sum() == 0
    perm = np.random.default_rng(0).permutation(B)
    bits2, it2 = dec.lut_decode_batch(cha[perm], msg[perm])
    assert (it2 == it[perm]).all() and (bits2 == bits[perm]).all()
    idx = _sample(it, 64)
    cd.set_exit_conditions(50, True, True)
    wb, wi = cd.lut_decode_batch(cha[idx], msg[idx])
    assert (wi == it[idx]).all() and (wb == bits[idx]).all()
    dec.close()


@pytest.mark.parametrize("name,B", [("dvbs2_q4_i6", 1030), ("twin64800_q4_i6", 1030)])
def test_c3_and_twin_every_frame_against_the_oracle(name, B):
    """C3 (with the degree-1 extension) and the twin the reference runs as is (check degrees up to 9: the
    middle bucket of the fused kernel), six iterations so that the oracle can follow the WHOLE batch."""
    cd = oracle_codec(name)
    dec = product_decoder(cd)
    assert dec.describe()["skewed_pipeline"] == 1
    cha, msg, _ = awgn_labels(cd, B, 1.0, seed=4242)
    n = 96                                                             # oracle: ~25 frames/s at six iterations
    for psc, pisc in [(True, True), (True, False), (False, False)]:
        dec.set_exit_conditions(cd.max_iters, psc, pisc)
        bits, it = dec.lut_decode_batch(cha, msg)
        cd.set_exit_conditions(cd.max_iters, psc, pisc)
        idx = np.r_[0:n // 2, B - n // 2:B]                            # both halves of the pipeline, the ragged last group
        wb, wi = cd.lut_decode_batch(cha[idx], msg[idx])
        assert (wi == it[idx]).all() and (wb == bits[idx]).all()
    dec.close()


@pytest.mark.parametrize("name", ["c5_minlut", "c5_chklut"])
def test_c5_wide_checks(name):
    """BASELINE config 5: (6,32) N=2048, 3-bit messages, trees/6_32_wide.ini, QCHA initial messages; min-sum checks
    (as shipped) and the 31-leaf CHKTREE (min_lut = false).  700 frames = two frame groups, generated kernels."""
    cd = oracle_codec(name)
    dec = product_decoder(cd)
    assert dec.describe()["vn_classes"][0]["kernel"] == "lutldpc_jit_pass" and dec.describe()["resident"] == (1 if name == "c5_minlut" else 0), dec.describe()      # (the 31-leaf CHKTREE runs faster through the streaming pass kernels)
    cha, msg, _ = awgn_labels(cd, 700, 4.0, seed=31, mode=1)
    compare(cd, dec, cha, msg, True, True)
    compare(cd, dec, cha, msg, False, False)
    dec.close()


def test_noise_free_and_saturated_frames_at_full_size():
    """All labels at the positive extreme: the all-zero codeword is returned, with the iteration codes of the
    reference (0 when the initial syndrome check is on, 1 with parity_check_iter only, +50 in fixed-work mode)."""
    cd = oracle_codec("dvbs2_q4")
    N = cd.code.nvar
    cha = np.full((520, N), cd.nq_cha - 1, np.uint8)
    msg = np.full((520, N), cd.nq_msg[0] - 1, np.uint8)
    dec = product_decoder(cd)
    for psc, pisc, want in [(True, True, 0), (True, False, 1), (False, False, 50)]:
        dec.set_exit_conditions(50, psc, pisc)
        bits, it = dec.lut_decode_batch(cha, msg)
        assert (it == want).all() and not bits.any()
    dec.close()
