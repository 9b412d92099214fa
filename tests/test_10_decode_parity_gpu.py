"""Parity proper: the HIP path (through the C-ABI) against the oracle on identical label arrays.
Bit-exact: every decoded bit and the returned iteration code (src/LDPC_Code_LUT.cpp:259-353)."""
import numpy as np
import pytest

from helpers import awgn_labels, compare as _compare, oracle_codec, product_decoder

pytestmark = pytest.mark.gpu


CASES = [
    # name, B, snr_db
    ("n500_q4", 70, 1.8),                  # C1: irregular dv{2,3,9,17}, mix of converging and failing frames
    ("reg36_n1000_q4", 300, 1.6),          # ragged batch > one 256-frame tile
    ("reg36_n1000_mixed", 64, 2.2),        # non-uniform Nq_Msg + reuse_lut
    ("reg36_n1000_q5", 40, 1.9),           # 32 labels: byte rows
    ("reg36_n1000_q3_chklut", 40, 2.5),    # CHKTREE check update (min_lut = false)
    ("reg36_n1000_rootonly", 33, 2.5),     # 3-input root tables
    ("reg36_n1000_high", 33, 2.0),
    ("c5_minlut", 48, 4.0),                # C5: (6,32), 3-bit, file trees, QCHA
    ("c5_chklut", 20, 4.2),                # C5 with the 31-leaf check tree
]


@pytest.mark.parametrize("name,B,snr", CASES)
@pytest.mark.parametrize("psc,pisc", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("resident", ["1", "0"])
def test_lut_decode_matches_oracle(name, B, snr, psc, pisc, resident, monkeypatch):
    """resident = 1: the default for these codes -- one generated kernel keeps the messages in LDS for the whole decode
    (jit_resident.hpp); resident = 0: the streaming kernels (rows in HBM, one launch per pass)."""
    monkeypatch.setenv("LUTLDPC_RESIDENT", resident)
    cd = oracle_codec(name)
    dec = product_decoder(cd)
    assert dec.describe()["resident"] == (int(resident) if name != "c5_chklut" else 0)      # (wide CHKTREE checks: streaming kernels)
    mode = 1 if name.startswith("c5") else 0
    cha, msg, _ = awgn_labels(cd, B, snr, seed=100 + B, mode=mode)
    # frame 0: noise-free all-zero codeword (pisc returns 0), frame 1: all labels identical minimum
    cha[0] = cd.nq_cha - 1; msg[0] = cd.nq_msg[0] - 1
    it = _compare(cd, dec, cha, msg, psc, pisc)
    if pisc:
        assert it[0] == 0
    if psc and not pisc:
        assert it[0] == 1
    dec.close()


@pytest.mark.parametrize("name,B,snr", [("n500_q4", 1100, 1.6), ("reg36_n1000_q4", 1537, 2.0), ("reg36_n1000_mixed", 1025, 2.2), ("dvbs2_q4_i6", 1030, 1.0)])
@pytest.mark.parametrize("env", [{"LUTLDPC_RESIDENT": "0"}])
def test_skewed_pipeline(name, B, snr, env, monkeypatch):
    """Batches of two or more frame groups run as two halves half an iteration out of phase
    (pass_fused_kernel): uneven halves, ragged last group, early termination on and off."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cd = oracle_codec(name)
    dec = product_decoder(cd)
    desc = dec.describe()
    # (n500_q4: variable degrees up to 17, checks up to 10 -> the widest bucket of the fused kernel)
    assert desc["skewed_pipeline"] == (0 if "LUTLDPC_SKEW" in env else 1), desc
    cha, msg, _ = awgn_labels(cd, B, snr, seed=4242)
    _compare(cd, dec, cha, msg, True, True)
    _compare(cd, dec, cha, msg, True, False)
    _compare(cd, dec, cha, msg, False, False)
    dec.close()


@pytest.mark.parametrize("name", ["c5_minlut", "reg36_n1000_rootonly", "reg36_n1000_high", "reg36_n1000_q5", "c5_chklut", "reg36_n1000_q3_chklut"])
def test_generated_kernels_are_used_and_match(name, monkeypatch):
    """jit.hpp: the kernel generated from the node program vs the oracle, and vs the interpreter (LUTLDPC_JIT=0)."""
    monkeypatch.setenv("LUTLDPC_RESIDENT", "0")           # the per-class pass kernels of the streaming path are what is tested here
    cd = oracle_codec(name)
    dec = product_decoder(cd)
    # (degree 3: auto_bin_high over two message leaves IS the balanced shape -> compile-time kernel)
    jit_expected = name not in ("reg36_n1000_high", "reg36_n1000_q3_chklut")     # (the latter: balanced variable trees, CHKTREE checks)
    assert dec.describe()["vn_classes"][0]["kernel"] == ("lutldpc_jit_pass" if jit_expected else "vn_balanced_fast_kernel"), dec.describe()
    if name.endswith("chklut"):
        assert dec.describe()["cn_classes"][0]["kernel"] == "lutldpc_jit_pass", dec.describe()
    mode = 1 if name.startswith("c5") else 0
    cha, msg, _ = awgn_labels(cd, 700, 4.0 if name.startswith("c5") else 2.2, seed=31, mode=mode)
    _compare(cd, dec, cha, msg, True, True)
    _compare(cd, dec, cha, msg, False, False)
    got = dec.lut_decode_batch(cha, msg)
    dec.close()
    monkeypatch.setenv("LUTLDPC_JIT", "0")
    ref = product_decoder(cd)
    assert ref.describe()["vn_classes"][0]["kernel"] == ("tree_pass_kernel<VAR>" if jit_expected else "vn_balanced_fast_kernel")
    if name.endswith("chklut"):
        assert ref.describe()["cn_classes"][0]["kernel"] == "tree_pass_kernel<CHK>"
    ref.set_exit_conditions(cd.max_iters, False, False)
    want = ref.lut_decode_batch(cha, msg)
    assert (got[0] == want[0]).all() and (got[1] == want[1]).all()
    ref.close()


def test_chain_fusion_is_on_for_the_dual_diagonal_code():
    """DVB-S2: 11 of 12 parity nodes (degree 2, checks c and c+1; twelve checks per wave) are updated inside the check pass in
    fixed-work mode."""
    cd = oracle_codec("dvbs2_q4_i6")
    dec = product_decoder(cd)
    assert dec.describe()["chain_nodes"] == 29699, dec.describe()
    cha, msg, _ = awgn_labels(cd, 1030, 1.0, seed=99)          # three frame groups: halves of two and one
    _compare(cd, dec, cha, msg, False, False)
    dec.set_exit_conditions(cd.max_iters, True, False)         # early termination: chain fusion off, same buffers ...
    dec.lut_decode_batch(cha[:600], msg[:600])
    _compare(cd, dec, cha[:520], msg[:520], False, False)      # ... and on again
    dec.close()


@pytest.mark.parametrize("resident", ["1", "0"])
def test_graph_replay_of_repeated_decodes(resident, monkeypatch):
    """From the second decode of a given (batch size, exit conditions) on, the launch sequence is captured
    and replayed as one hipGraph: new labels in the same buffers, changed exit conditions, a batch size
    that forces bigger buffers (captured addresses become stale) and a return to the first size."""
    monkeypatch.setenv("LUTLDPC_RESIDENT", resident)
    cd = oracle_codec("reg36_n1000_q4")
    dec = product_decoder(cd)
    for rep, (B, psc, pisc) in enumerate([(600, True, True), (600, True, True), (600, True, True), (600, False, False), (600, False, False),
                                          (600, False, False), (1300, True, False), (1300, True, False), (1300, True, False), (600, True, True),
                                          (600, True, True), (600, True, True)]):
        cha, msg, _ = awgn_labels(cd, B, 2.0, seed=900 + rep)
        _compare(cd, dec, cha, msg, psc, pisc)
    dec.close()


@pytest.mark.parametrize("B", [1, 3, 255, 256, 257, 511, 512, 513])
@pytest.mark.parametrize("resident", ["1", "0"])
def test_batch_sizes(B, resident, monkeypatch):
    monkeypatch.setenv("LUTLDPC_RESIDENT", resident)
    cd = oracle_codec("n500_q4_i8")
    dec = product_decoder(cd)
    cha, msg, _ = awgn_labels(cd, B, 2.0, seed=B)
    _compare(cd, dec, cha, msg, True, True)
    dec.close()


def test_decode_llr_entry_matches_oracle_quantiser():
    cd = oracle_codec("n500_q4_i8")
    dec = product_decoder(cd)
    cha, msg, llr = awgn_labels(cd, 37, 2.0, seed=5)
    llr[3, :8] = cd.qb_cha[:8]              # values exactly on a boundary: x <= b stops the scan
    cha, msg = __import__("oracle.oracle", fromlist=["x"]).quant_nonlin(llr, cd.qb_cha), __import__("oracle.oracle", fromlist=["x"]).quant_nonlin(llr, cd.qb_msg)
    cd.set_exit_conditions(8, True, True); dec.set_exit_conditions(8, True, True)
    want_bits, want_it = cd.lut_decode_batch(cha, msg)
    got_bits, got_it = dec.decode_llr_batch(llr, cd.qb_cha, cd.qb_msg, mode=0)
    assert (want_it == got_it).all() and (want_bits == got_bits).all()
    # QCHA: initial messages through Nq_Cha_2_Nq_Msg_map (src/LDPC_Code_LUT.cpp:215-217)
    msg_q = cd.cha2msg_map[cha].astype(np.uint8)
    want_bits, want_it = cd.lut_decode_batch(cha, msg_q)
    got_bits, got_it = dec.decode_llr_batch(llr, cd.qb_cha, None, mode=1, cha2msg_map=cd.cha2msg_map)
    assert (want_it == got_it).all() and (want_bits == got_bits).all()
    dec.close()


def test_fewer_iterations_than_designed_is_rejected_unless_decision_set():
    import lut_ldpc_amd as L
    cd = oracle_codec("n500_q4_i8")
    dec = product_decoder(cd)
    with pytest.raises(L.LutLdpcError):
        dec.set_exit_conditions(4)
    dec.close()


@pytest.mark.parametrize("K,M,dv,bucket", [(1600, 400, 3, 1), (3600, 400, 3, 2), (840, 420, 3, 0), (1120, 420, 3, 3)])
def test_chain_fusion_in_every_degree_bucket(tmp_path, K, M, dv, bucket, monkeypatch):
    """(streaming kernels: LUTLDPC_RESIDENT=0 -- these small codes would otherwise be decoded out of LDS)
    Dual-diagonal codes with check degrees 14 (middle bucket), 29 (widest), 8 (first) and 10 (bucket 3): most zigzag nodes are updated
    inside the check pass in every bucket, one frame group and several, fixed work and early termination."""
    from helpers import write_ira_alist
    from oracle import oracle as orc
    monkeypatch.setenv("LUTLDPC_RESIDENT", "0")
    N, _ = write_ira_alist(tmp_path / "ira.alist", K, M, dv, seed=K)
    code = orc.Code(tmp_path / "ira.alist")
    cd = orc.Codec(code, skip_rank=True)
    cd.set_rank(M)
    cd.rate = 1.0 - M / N
    sig = {1: 0.52, 2: 0.40, 0: 0.62, 3: 0.57}[bucket]
    cd.design_luts(sigma2=sig ** 2, max_iters=10, nq_msg=np.full(10, 16, np.int32), nq_cha=16)
    dec = product_decoder(cd)
    desc = dec.describe()
    assert desc["fused_bucket"] == bucket and desc["skewed_pipeline"] == 1, desc
    assert desc["chain_nodes"] >= M // 2, desc                    # at least every second zigzag node (3 of 4 with four checks per wave)
    snr = -10 * np.log10(2 * cd.rate * sig * sig) + 0.3
    for B in (700, 200):
        cha, msg, _ = awgn_labels(cd, B, snr, seed=B)
        it = _compare(cd, dec, cha, msg, True, True)
        assert (it > 0).sum() > 0
        _compare(cd, dec, cha, msg, False, False)
    dec.close()


def test_argument_errors_and_out_of_range_labels():
    """Error codes where the reference aborts (it_assert): an empty batch and a device-less handle are refused.  Labels outside
    the alphabet (undefined in the reference: unchecked table index) are clamped to the largest label when the rows are
    built -- the neighbours in the same 256-byte row decode exactly as without the bad frame."""
    import lut_ldpc_amd as L
    cd = oracle_codec("n500_q4")
    dec = product_decoder(cd)
    cha, msg, _ = awgn_labels(cd, 700, 2.0, seed=31)
    with pytest.raises(L.LutLdpcError):
        dec.lut_decode_batch(cha[:0], msg[:0])
    host_only = product_decoder(cd, device=-1)
    with pytest.raises(L.LutLdpcError):
        host_only.lut_decode_batch(cha[:4], msg[:4])
    host_only.close()
    dec.set_exit_conditions(cd.max_iters, True, True)
    good_bits, good_it = dec.lut_decode_batch(cha, msg)
    bad_c, bad_m = cha.copy(), msg.copy()
    bad_c[5, ::3] = 200                                       # frame 5: labels far outside the 16-label alphabets
    bad_m[5, 1::3] = 255
    bits, it = dec.lut_decode_batch(bad_c, bad_m)
    keep = np.arange(700) != 5
    assert (bits[keep] == good_bits[keep]).all() and (it[keep] == good_it[keep]).all()
    clamp_bits, clamp_it = dec.lut_decode_batch(np.minimum(bad_c, cd.nq_cha - 1), np.minimum(bad_m, cd.nq_msg[0] - 1))
    assert (bits[5] == clamp_bits[5]).all() and it[5] == clamp_it[5]
    dec.close()


FUZZ = [  # N, M, variable degrees, their shares, Nq_Cha, Nq_Msg, iterations, design sigma, B
    (600, 300, [2, 3, 6], [0.4, 0.45, 0.15], 16, 16, 9, 0.80, 777),
    (900, 300, [3, 4], [0.7, 0.3], 16, 16, 7, 0.62, 513),
    (1200, 600, [2, 3, 9], [0.35, 0.5, 0.15], 16, 8, 8, 0.82, 300),
    (800, 400, [3], [1.0], 8, 8, 6, 0.80, 1025),
    (500, 250, [2, 4, 12], [0.45, 0.45, 0.10], 16, 16, 10, 0.85, 64),
    (1000, 200, [3, 5], [0.8, 0.2], 16, 16, 5, 0.45, 1300),
]


@pytest.mark.parametrize("case", range(len(FUZZ)))
@pytest.mark.parametrize("resident", ["1", "0"])
def test_random_irregular_codes(tmp_path, case, resident, monkeypatch):
    """Random irregular graphs the build has never seen (several variable and check degree classes, check degrees that differ
    by one, wide and narrow checks, 3- and 4-bit alphabets): design with the oracle, decode through the default path, every
    bit and iteration code against the oracle in all three exit modes, ragged batch sizes."""
    from helpers import write_random_alist
    from oracle import oracle as orc
    monkeypatch.setenv("LUTLDPC_RESIDENT", resident)
    N, M, dvc, dvp, nqc, nqm, I, sig, B = FUZZ[case]
    dv, dc = write_random_alist(tmp_path / "r.alist", N, M, dvc, dvp, seed=100 + case)
    code = orc.Code(tmp_path / "r.alist")
    assert len(set(dc.tolist())) >= 1 and code.nvar == N
    cd = orc.Codec(code, skip_rank=True)
    cd.set_rank(M)
    cd.rate = 1.0 - M / N
    cd.design_luts(sigma2=sig ** 2, max_iters=I, nq_msg=np.full(I, nqm, np.int32), nq_cha=nqc)
    dec = product_decoder(cd)
    snr = -10 * np.log10(2 * cd.rate * sig * sig) + 0.5
    cha, msg, _ = awgn_labels(cd, B, snr, seed=case)
    it = _compare(cd, dec, cha, msg, True, True)
    _compare(cd, dec, cha, msg, True, False)
    _compare(cd, dec, cha, msg, False, False)
    assert len(set(it.tolist())) >= 2
    dec.close()


@pytest.mark.parametrize("name,level", [("n500_q4_i8", 2), ("n500_q4_i8", 3), ("reg36_n1000_q3_chklut", 3), ("reg36_n1000_mixed", 2)])
def test_output_verbosity_message_dumps_match_the_oracle_text(name, level):
    """output_verbosity 2 / 3 (src/LDPC_Code_LUT.cpp:292-298,311-317,331-337): the initial, check-to-variable (level 3) and
    variable-to-check messages of every iteration, printed frame after frame -- the golden-vector format SURVEY section 7 names.
    The product's text (LDPC_Code_LUT::lut_decode with the dumps taken on the device) equals the oracle's restatement of those
    print statements: frames that pass the test on the channel decisions print nothing, frames that leave through the exit test
    stop before the dump of their last variable update, iteration numbers in upper-case hex (std::hex is sticky)."""
    from test_host_design_parity import product_codec
    cd = oracle_codec(name)
    pcd = product_codec(name, device=0)
    assert pcd.var_trees_txt == cd.var_tree_txt                        # same tables on both sides (design parity)
    cha, msg, _ = awgn_labels(cd, 70, 2.6, seed=level)
    cha[5] = cd.nq_cha - 1; msg[5] = cd.nq_msg[0] - 1                   # passes the test on the channel decisions
    for psc, pisc in ((True, True), (False, False)):
        cd.set_exit_conditions(cd.max_iters, psc, pisc)
        pcd.set_exit_conditions(cd.max_iters, psc, pisc)
        wb, wi, wtxt = cd.lut_decode_dump(cha, msg, level)
        gb, gi, gtxt = pcd.lut_decode_dump(cha, msg, level)
        assert (wi == gi).all() and (wb == gb).all()
        if psc:
            assert (wi == 0).sum() >= 1 and ((wi > 0) & (wi < cd.max_iters)).sum() >= 1      # every kind of return is in the text
        assert len(wtxt) > 1000 and gtxt == wtxt, (len(gtxt), len(wtxt), next((i for i in range(min(len(gtxt), len(wtxt))) if gtxt[i] != wtxt[i]), None))
    # the raw trace of the decoder handle: dump 0 = every edge carries its node's initial message (:284-289)
    dec = product_decoder(cd)
    dec.set_exit_conditions(cd.max_iters, False, False)
    _, _, tr = dec.lut_decode_batch_trace(cha[:9], msg[:9], 3, cd.code.nedges)
    assert tr.shape == (1 + 2 * cd.max_iters, 9, cd.code.nedges)
    assert (tr[0] == np.repeat(msg[:9], cd.code.dv, axis=1)).all()
    dec.close()
    pcd.close()
