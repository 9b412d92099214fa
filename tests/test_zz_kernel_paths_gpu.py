"""Knob variants (run LAST in the GPU suite: the default configuration is covered by test_00_* / test_10_*).

Size-independent properties at BASELINE.json's full configuration (DVB-S2 N=64800, 4-bit labels, all 50
iterations), where the oracle is too slow to decode a whole batch: four independent kernel paths must agree bit for
bit, a batch must decode like its parts, frames must not influence each other, and a few frames are checked against
the oracle at the full iteration count."""
import numpy as np
import pytest

from helpers import awgn_labels, compare as _compare, oracle_codec, product_decoder

pytestmark = pytest.mark.gpu

@pytest.fixture(autouse=True)
def _streaming_kernels(monkeypatch):
    """Everything in this file is about the STREAMING kernels (rows in HBM): the codes that would otherwise be decoded out of LDS
    by the generated resident kernel (the default for N <= ~35000 edges) are pinned to them; the resident path is one of PATHS."""
    monkeypatch.setenv("LUTLDPC_RESIDENT", "0")


PATHS = [{}, {"LUTLDPC_RESIDENT": "1"}, {"LUTLDPC_SKEW": "0"}, {"LUTLDPC_PACK": "1"}, {"LUTLDPC_USE_FAST": "0"}, {"LUTLDPC_GRAPH": "0", "LUTLDPC_VN_EDGES_PER_WAVE": "8"},
         {"LUTLDPC_CHAIN": "0"}, {"LUTLDPC_CN_EDGES_PER_WAVE": "56", "LUTLDPC_PACK": "1"}]


def _decode(cd, cha, msg, psc, env, monkeypatch, repeat=1):
    for k in ("LUTLDPC_SKEW", "LUTLDPC_PACK", "LUTLDPC_USE_FAST", "LUTLDPC_GRAPH", "LUTLDPC_VN_EDGES_PER_WAVE", "LUTLDPC_CHAIN", "LUTLDPC_CN_EDGES_PER_WAVE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("LUTLDPC_RESIDENT", "0")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dec = product_decoder(cd)
    dec.set_exit_conditions(cd.max_iters, psc, psc)
    for _ in range(repeat):
        bits, it = dec.lut_decode_batch(cha, msg)
    dec.close()
    return bits, it


@pytest.mark.parametrize("name,snr", [("dvbs2_q4", 1.3), ("reg36_n10000_q4", 1.9)])
def test_kernel_paths_agree_at_full_iteration_count(name, snr, monkeypatch):
    cd = oracle_codec(name)
    assert cd.max_iters == 50
    B = 1100                                                   # three frame groups: uneven halves, ragged last group
    cha, msg, _ = awgn_labels(cd, B, snr, seed=2026)
    for psc in (False, True):
        # fused two-half pipeline replayed as a graph (third call) is the reference point; on DVB-S2 it runs with chain
        # fusion (degree-2 parity nodes updated inside the check pass), which PATHS switches off / re-partitions
        ref_bits, ref_it = _decode(cd, cha, msg, psc, {}, monkeypatch, repeat=3)
        assert ((ref_it > 0).sum() > 0) and ((np.abs(ref_it) == 50).sum() > 0 or psc)
        for env in PATHS[1:]:
            bits, it = _decode(cd, cha, msg, psc, env, monkeypatch)
            assert (it == ref_it).all(), (env, np.flatnonzero(it != ref_it)[:8])
            assert (bits == ref_bits).all(), env


@pytest.mark.parametrize("name,B,snr", [("n500_q4", 300, 1.8), ("reg36_n1000_mixed", 64, 2.2), ("c5_chklut", 20, 4.2), ("reg36_n1000_high", 33, 2.0)])
@pytest.mark.parametrize("env", [{"LUTLDPC_PACK": "1"}, {"LUTLDPC_USE_FAST": "0"}, {"LUTLDPC_PACK": "1", "LUTLDPC_USE_FAST": "0"}])
def test_kernel_variants(name, B, snr, env, monkeypatch):
    """Byte rows vs nibble rows, specialised vs generic kernels: every combination is bit-exact."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cd = oracle_codec(name)
    dec = product_decoder(cd)
    desc = dec.describe()
    assert desc["pack"] == (1 if "LUTLDPC_PACK" in env else 2) and desc["use_fast"] == (0 if "LUTLDPC_USE_FAST" in env else 1)
    mode = 1 if name.startswith("c5") else 0
    cha, msg, _ = awgn_labels(cd, B, snr, seed=77, mode=mode)
    _compare(cd, dec, cha, msg, True, True)
    _compare(cd, dec, cha, msg, False, False)
    dec.close()


@pytest.mark.parametrize("name,B,snr", [("n500_q4", 1100, 1.6), ("reg36_n1000_q4", 1537, 2.0), ("reg36_n1000_mixed", 1025, 2.2)])
@pytest.mark.parametrize("env", [{"LUTLDPC_PACK": "1"}, {"LUTLDPC_SKEW": "0"}, {"LUTLDPC_VALIDATE": "1"}, {"LUTLDPC_LATE_HARD": "0"}, {"LUTLDPC_FIRST_FROM_NODES": "0"}])
def test_skewed_pipeline_variants(name, B, snr, env, monkeypatch):
    """Byte rows, per-class launches instead of the fused pipeline, the validating debug mode (every role checked against
    the allocation sizes, one stream synchronisation per fused launch), and decided bits stored by every variable pass
    instead of recovered at the end (LUTLDPC_LATE_HARD=0), the initial messages copied to the edge rows by their own kernel
    instead of read by the first check pass (LUTLDPC_FIRST_FROM_NODES=0)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cd = oracle_codec(name)
    dec = product_decoder(cd)
    assert dec.describe()["skewed_pipeline"] == (0 if "LUTLDPC_SKEW" in env else 1)
    cha, msg, _ = awgn_labels(cd, B, snr, seed=4242)
    _compare(cd, dec, cha, msg, True, True)
    _compare(cd, dec, cha, msg, True, False)
    _compare(cd, dec, cha, msg, False, False)
    dec.close()


@pytest.mark.parametrize("name,B,snr", [("reg36_n1000_q4", 2100, 1.9), ("n500_q4", 1300, 2.2)])
def test_compaction_of_surviving_frames(name, B, snr, monkeypatch):
    """LUTLDPC_COMPACT=1: every few iterations the slots of a half are permuted (active frames first) and put back at
    the end; decided bits, iteration codes and pending flags travel with their frame."""
    monkeypatch.setenv("LUTLDPC_COMPACT", "1")
    monkeypatch.setenv("LUTLDPC_COMPACT_FIRST", "3")
    monkeypatch.setenv("LUTLDPC_COMPACT_EVERY", "2")
    monkeypatch.setenv("LUTLDPC_COMPACT_MARGIN", "0")          # permute whenever a group falls idle (the default weighs cost against gain)
    cd = oracle_codec(name)
    dec = product_decoder(cd)
    assert dec.describe()["compaction"] == 1
    cha, msg, _ = awgn_labels(cd, B, snr, seed=515)
    it = _compare(cd, dec, cha, msg, True, True)
    assert 0 < (it > 0).sum() and len(set(it.tolist())) > 4          # frames finish at many different iterations
    _compare(cd, dec, cha, msg, True, False)
    _compare(cd, dec, cha, msg, False, False)
    dec.close()


@pytest.mark.parametrize("env,B", [({}, 512 * 36 + 77), ({"LUTLDPC_PACK": "1"}, 256 * 40 + 5), ({}, 512 * 64 - 3)])
def test_compaction_with_many_groups(env, B, monkeypatch):
    """Halves of 18-20 frame groups (and of 32, the most the row kernel takes: 66 KB of LDS, two groups for every wave): every
    wave of the row-permutation kernel fetches and builds two groups (w and w + 16),
    frames travel across many groups, several permutations per decode.  Oracle: flat-table mode on all cores."""
    monkeypatch.setenv("LUTLDPC_COMPACT", "1")
    monkeypatch.setenv("LUTLDPC_COMPACT_FIRST", "3")
    monkeypatch.setenv("LUTLDPC_COMPACT_EVERY", "3")
    monkeypatch.setenv("LUTLDPC_COMPACT_MARGIN", "0")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cd = oracle_codec("n500_q4")
    dec = product_decoder(cd)
    assert dec.describe()["compaction"] == 1
    cha, msg, _ = awgn_labels(cd, B, 2.0, seed=99)
    for f in (3, 600, 5000, B - 2):                            # noise-free frames: they pass the test on the channel decisions (pisc) and
        cha[f] = cd.nq_cha - 1                                 # are moved by the first permutation like every other finished frame
        msg[f] = cd.nq_msg[0] - 1
    it = _compare(cd, dec, cha, msg, True, True, flat=True)
    assert len(set(it.tolist())) > 8 and (it == 0).sum() == 4
    _compare(cd, dec, cha, msg, True, False, flat=True)
    dec.close()


def test_compaction_that_drops_the_rows_of_finished_frames(monkeypatch):
    """LUTLDPC_COMPACT_KEEP=0: the earlier flow -- decided bits recovered at every permuting check point, only the active frames'
    rows moved, decided-bit rows permuted along."""
    monkeypatch.setenv("LUTLDPC_COMPACT", "1")
    monkeypatch.setenv("LUTLDPC_COMPACT_KEEP", "0")
    monkeypatch.setenv("LUTLDPC_COMPACT_FIRST", "3")
    monkeypatch.setenv("LUTLDPC_COMPACT_EVERY", "2")
    monkeypatch.setenv("LUTLDPC_COMPACT_MARGIN", "0")
    cd = oracle_codec("n500_q4")
    dec = product_decoder(cd)
    cha, msg, _ = awgn_labels(cd, 512 * 9 + 40, 2.0, seed=7)
    cha[11] = cd.nq_cha - 1; msg[11] = cd.nq_msg[0] - 1
    it = _compare(cd, dec, cha, msg, True, True, flat=True)
    assert len(set(it.tolist())) > 8 and (it == 0).sum() == 1
    dec.close()


def test_placement_search_of_the_row_buffers_changes_nothing_but_where_they_are(monkeypatch):
    """A batch whose rows exceed 1 GiB makes the streaming decoder try several allocations of its row buffers and keep the fastest
    (decoder.hip: place_rows; DESIGN.md "The levels are buffer placement"): describe() reports the search, the outputs equal those of a
    decoder that takes the first allocation (LUTLDPC_PLACE=0) in both exit modes, and a sample of frames equals the oracle."""
    cd = oracle_codec("dvbs2_q4_i6")
    B = 5300                                   # 11 frame groups: 1.19 GB of rows
    cha, msg, _ = awgn_labels(cd, B, 1.4, seed=77)
    cha[:3] = cd.nq_cha - 1; msg[:3] = int(cd.nq_msg[0]) - 1         # a few noise-free frames (they pass the test on the channel decisions)
    out = {}
    for place in ("16", "0"):
        monkeypatch.setenv("LUTLDPC_PLACE", place)
        dec = product_decoder(cd)
        for psc in (False, True):
            dec.set_exit_conditions(cd.max_iters, psc, psc)
            out[place, psc] = dec.lut_decode_batch(cha, msg)
        info = dec.describe()["placement"]
        if place == "0":
            assert info is None
        else:
            assert info["candidates"] >= 2 and len(info["probe_ms"]) == info["candidates"] and 0 <= info["chosen"] < info["candidates"], info
            assert min(info["probe_ms"]) == pytest.approx(info["probe_ms"][info["chosen"]])
        dec.close()
    for psc in (False, True):
        assert (out["16", psc][0] == out["0", psc][0]).all() and (out["16", psc][1] == out["0", psc][1]).all()
        cd.set_exit_conditions(cd.max_iters, psc, psc)
        idx = np.r_[0:4, B - 24:B]
        wb, wi = cd.lut_decode_batch(cha[idx], msg[idx])
        assert (wb == out["16", psc][0][idx]).all() and (wi == out["16", psc][1][idx]).all()
