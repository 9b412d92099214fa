"""The production as-shipped path -- what BASELINE config 4 runs in every batch -- against the oracle.

`parity_check_iter = true` (the reference's default, src/LDPC_BER_Sim.cpp:71,500 -> psc = pisc = 1,
src/LDPC_Code_LUT.cpp:275-279,327-329,437-452) at production batch sizes combines, in ONE decode: chain fusion (degree-2
zigzag nodes updated inside the check pass), compaction of the surviving frames with the frames that left keeping their
rows, decided bits recovered at the end (`hard_from_frozen_kernel` + `chain_hard_kernel`), and `hard_from_labels_masked_kernel`
for the frames that passed the test on the channel decisions.  Compaction switches on by itself only from
`describe()["compaction_min_groups"]` frame groups on (DVB-S2: 8 groups = 3585+ frames), so the small-batch tests never see
it.  Runs right after the BASELINE configurations (test_00_*), before the knob variants."""
import numpy as np
import pytest

from helpers import awgn_labels, compare, oracle_codec, product_decoder

pytestmark = pytest.mark.gpu


def _ira_codec(tmp_path, K, M, dv, max_iters, sig):
    from helpers import write_ira_alist
    from oracle import oracle as orc
    N, _ = write_ira_alist(tmp_path / "ira.alist", K, M, dv, seed=K + max_iters)
    code = orc.Code(tmp_path / "ira.alist")
    cd = orc.Codec(code, skip_rank=True)
    cd.set_rank(M)
    cd.rate = 1.0 - M / N
    cd.design_luts(sigma2=sig ** 2, max_iters=max_iters, nq_msg=np.full(max_iters, 16, np.int32), nq_cha=16)
    return cd


@pytest.mark.parametrize("keep", ["1", "0"])
@pytest.mark.parametrize("K,M,dv,bucket,sig", [(840, 420, 3, 0, 0.62), (1600, 400, 3, 1, 0.52)])
def test_ira_code_chain_fusion_with_compaction(tmp_path, monkeypatch, K, M, dv, bucket, sig, keep):
    """A dual-diagonal code (check degrees 8 and 14: first and middle bucket of the fused kernel), compaction forced on with a
    check point every second iteration and no cost margin (frames move several times per decode), nine and ten frame groups
    (halves of 5 + 4 / 5 + 5, ragged last group), both row flows (the frames that left keep their rows / drop them), all three
    exit modes, EVERY frame against the oracle.  Some frames are noise-free: they pass the test on the channel decisions and are
    moved by the first permutation like every other finished frame."""
    monkeypatch.setenv("LUTLDPC_RESIDENT", "0")              # the streaming path: this small code would otherwise be decoded out of LDS
    monkeypatch.setenv("LUTLDPC_COMPACT", "1")
    monkeypatch.setenv("LUTLDPC_COMPACT_KEEP", keep)
    monkeypatch.setenv("LUTLDPC_COMPACT_FIRST", "2")
    monkeypatch.setenv("LUTLDPC_COMPACT_EVERY", "2")
    monkeypatch.setenv("LUTLDPC_COMPACT_MARGIN", "0")
    cd = _ira_codec(tmp_path, K, M, dv, 16, sig)
    dec = product_decoder(cd)
    desc = dec.describe()
    assert desc["fused_bucket"] == bucket and desc["skewed_pipeline"] == 1 and desc["compaction"] == 1, desc
    assert desc["chain_nodes"] >= M // 2, desc
    snr = -10 * np.log10(2 * cd.rate * sig * sig) + 0.25
    for B in (512 * 8 + 77, 512 * 10):
        cha, msg, _ = awgn_labels(cd, B, snr, seed=B)
        for f in (0, 700, 2048, B - 1):
            cha[f] = cd.nq_cha - 1
            msg[f] = cd.nq_msg[0] - 1
        it = compare(cd, dec, cha, msg, True, True, flat=True)
        assert (it == 0).sum() == 4 and len(set(it.tolist())) > 5, sorted(set(it.tolist()))      # frames leave at many iterations
        assert (it < 0).sum() > 0                                                                # and some never do
        it = compare(cd, dec, cha, msg, True, False, flat=True)
        assert (it == 1).sum() >= 4
        compare(cd, dec, cha, msg, False, False, flat=True)
    dec.close()


def _sample(it, n):
    """frames for the oracle: every kind of outcome (failed, initial-test exit, early exit spread over the iteration counts, full
    count), from every part of the batch"""
    rng = np.random.default_rng(1)
    groups = [np.flatnonzero(it < 0), np.flatnonzero(it == 0), np.flatnonzero((it > 0) & (it < 36)), np.flatnonzero((it >= 36) & (it < 42)),
              np.flatnonzero((it >= 42) & (it < 50)), np.flatnonzero(it == 50)]
    idx = []
    for g in groups:
        if len(g):
            idx += list(rng.choice(g, size=min(len(g), n // len(groups)), replace=False))
    rest = np.setdiff1d(np.arange(len(it)), idx)
    idx += list(rng.choice(rest, size=n - len(idx), replace=False))
    return np.array(sorted(idx))


def test_dvbs2_as_shipped_at_production_batch_size():
    """DVB-S2 N=64800, 50 iterations, psc = pisc = 1, 4200 frames = nine frame groups: the automatic compaction fires (asserted
    from describe()), chain fusion is on, the decided bits come from the frozen messages / the parity equations / the channel
    rows.  96 frames -- failed, passed on the channel decisions, early exits over the whole range of iteration counts, full
    count; from both halves and every group -- against the oracle, decoded twice more for the hipGraph replay."""
    cd = oracle_codec("dvbs2_q4")
    dec = product_decoder(cd)
    desc = dec.describe()
    B = 4200
    G = (B + desc["tile_frames"] - 1) // desc["tile_frames"]
    assert desc["compaction"] == 2 and 4 <= desc["compaction_min_groups"] <= G, desc           # automatic, and on for this batch
    assert desc["chain_nodes"] == 29699 and desc["skewed_pipeline"] == 1 and desc["fused_bucket"] == 0, desc
    cha, msg, _ = awgn_labels(cd, B, 1.3, seed=777)
    clean = [5, 511, 1400, 2600, 3333, B - 1]
    for f in clean:                                            # pass the test on the channel decisions, in both halves
        cha[f] = cd.nq_cha - 1
        msg[f] = cd.nq_msg[0] - 1
    noisy = np.arange(40, B, 175)                              # 24 frames well below the design point: they fail (negative count)
    cha[noisy], msg[noisy], _ = awgn_labels(cd, len(noisy), 0.6, seed=778)
    dec.set_exit_conditions(50, True, True)
    runs = [dec.lut_decode_batch(cha, msg) for _ in range(3)]  # plain launches, capture + replay, replay
    bits, it = runs[0]
    for b, i in runs[1:]:
        assert (i == it).all() and (b == bits).all()
    assert (it[clean] == 0).all() and (it == 0).sum() == len(clean)
    assert (it[noisy] < 0).sum() > 0 and ((it > 0) & (it < 40)).sum() > 0
    hist = np.bincount(it[it > 0], minlength=51)
    assert (hist > 0).sum() > 8, hist                          # frames leave over many iterations: the compaction has work to do
    assert bits[it >= 0].sum() == 0                            # all-zero codeword sent; a frame reported converged is a codeword
    idx = _sample(it, 96)
    cd.set_exit_conditions(50, True, True)
    wb, wi = cd.lut_decode_batch_flat(cha[idx], msg[idx])
    assert (wi == it[idx]).all(), (idx[wi != it[idx]][:8], wi[wi != it[idx]][:8], it[idx][wi != it[idx]][:8])
    bad = np.argwhere(wb != bits[idx])
    assert bad.size == 0, f"{len(bad)} bit mismatches, first at sample/bit {bad[:4].tolist()} (frames {idx[bad[:4, 0]].tolist()})"
    # the same batch without compaction and without chain fusion is the same decode
    dec.close()


def test_dvbs2_as_shipped_equals_the_path_without_compaction(monkeypatch):
    """Size-independent property at the production batch size: automatic compaction + chain fusion + late decided bits give the
    same bits and iteration codes as the plain pipeline (LUTLDPC_COMPACT=0, LUTLDPC_CHAIN=0, LUTLDPC_LATE_HARD=0) for ALL 4200
    frames -- the plain pipeline is the one the small-batch oracle tests cover."""
    cd = oracle_codec("dvbs2_q4")
    B = 4200
    cha, msg, _ = awgn_labels(cd, B, 1.3, seed=778)
    dec = product_decoder(cd)
    dec.set_exit_conditions(50, True, True)
    bits, it = dec.lut_decode_batch(cha, msg)
    dec.close()
    monkeypatch.setenv("LUTLDPC_COMPACT", "0")
    monkeypatch.setenv("LUTLDPC_CHAIN", "0")
    monkeypatch.setenv("LUTLDPC_LATE_HARD", "0")
    ref = product_decoder(cd)
    assert ref.describe()["chain_nodes"] == 0 and ref.describe()["compaction"] == 0
    ref.set_exit_conditions(50, True, True)
    rbits, rit = ref.lut_decode_batch(cha, msg)
    ref.close()
    assert (rit == it).all(), np.flatnonzero(rit != it)[:8]
    assert (rbits == bits).all()


@pytest.mark.parametrize("nq_cha", [12, 6])
@pytest.mark.parametrize("resident", ["0", "1"])
def test_channel_alphabet_whose_half_is_not_a_power_of_two(nq_cha, resident, monkeypatch):
    """Nq_Cha = 12 / 6: `label < Nq_Cha/2` (src/LDPC_Code_LUT.cpp:275) is not a sign BIT of the label; the test on the channel
    decisions (pisc) must go through the SWAR compare, not the bit trick of the label-row syndrome kernel."""
    from helpers import CODES
    from oracle import oracle as orc
    monkeypatch.setenv("LUTLDPC_RESIDENT", resident)
    code = orc.Code(CODES / "rate0.50_dv03_dc06_N1000.alist")
    cd = orc.Codec(code, skip_rank=True)
    cd.set_rank(500)
    cd.rate = 0.5
    cd.design_luts(sigma2=0.80 ** 2, max_iters=8, nq_msg=np.full(8, 8, np.int32), nq_cha=nq_cha)
    dec = product_decoder(cd)
    for B, snr in ((300, 10.5), (1100, 2.5)):
        cha, msg, _ = awgn_labels(cd, B, snr, seed=B)
        cha[3] = nq_cha - 1; msg[3] = cd.nq_msg[0] - 1
        it = compare(cd, dec, cha, msg, True, True)
        assert it[3] == 0
        if snr > 10:
            assert (it == 0).sum() > 10                           # at 10.5 dB many frames pass on the channel decisions alone
        compare(cd, dec, cha, msg, True, False)
        compare(cd, dec, cha, msg, False, False)
    dec.close()
