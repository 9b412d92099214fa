"""The LDS-resident decoder (jit_resident.hpp): codes whose edge messages fit the LDS of a compute unit are decoded by ONE generated
kernel -- all iterations of src/LDPC_Code_LUT.cpp:259-353 inside, sets of 8 frames retired one by one.  The default path for the
small BASELINE configurations (C1, C2, C5); here its own shapes: every workgroup geometry (sets per workgroup x threads), batches
that are not a multiple of a set or of a frame group, frames that pass the test on the channel decisions, byte rows, non-uniform
alphabets with reused LUT stages, CHKTREE check updates, table composition on and off.  Bit-exact against the oracle."""
import numpy as np
import pytest

from helpers import awgn_labels, compare, oracle_codec, product_decoder

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,B,snr,geom", [
    ("n500_q4", 1029, 1.8, {"LUTLDPC_RESIDENT_S": "1", "LUTLDPC_RESIDENT_NT": "256"}),
    ("n500_q4", 1029, 1.8, {"LUTLDPC_RESIDENT_S": "3", "LUTLDPC_RESIDENT_NT": "512"}),     # sets per workgroup not dividing 64
    ("n500_q4", 5, 1.8, {"LUTLDPC_RESIDENT_S": "7", "LUTLDPC_RESIDENT_NT": "512"}),        # fewer frames than one set
    ("reg36_n1000_q4", 2100, 1.9, {"LUTLDPC_RESIDENT_S": "5", "LUTLDPC_RESIDENT_NT": "1024"}),
    ("reg36_n1000_q4", 777, 1.9, {"LUTLDPC_RESIDENT_S": "12", "LUTLDPC_RESIDENT_NT": "256"}),
    ("reg36_n1000_mixed", 1025, 2.2, {}),                                                   # 16 -> 8 labels, reused stages: several code variants
    ("reg36_n1000_q5", 300, 1.9, {}),                                                       # 32 labels: byte rows, 4 frames per set
    ("reg36_n1000_q3_chklut", 520, 2.5, {}),                                                # CHKTREE check update
    ("reg36_n1000_rootonly", 130, 2.5, {}),
    ("c5_chklut", 260, 4.2, {"LUTLDPC_RESIDENT": "2"}),                              # 31-leaf check tree, degree 32 (streaming by default: forced here)
    ("c5_minlut", 1500, 4.0, {"LUTLDPC_RESIDENT_S": "1"}),                                  # two-sweep min-sum for degree 32
])
def test_resident_geometries_and_shapes(name, B, snr, geom, monkeypatch):
    for k, v in geom.items():
        monkeypatch.setenv(k, v)
    cd = oracle_codec(name)
    dec = product_decoder(cd)
    assert dec.describe()["resident"] == 1, dec.describe()
    mode = 1 if name.startswith("c5") else 0
    cha, msg, _ = awgn_labels(cd, B, snr, seed=B, mode=mode)
    for f in (0, B // 2, B - 1):                                # noise-free frames: pass the test on the channel decisions
        cha[f] = cd.nq_cha - 1
        msg[f] = cd.nq_msg[0] - 1
    it = compare(cd, dec, cha, msg, True, True)
    assert (it == 0).sum() >= 1
    compare(cd, dec, cha, msg, True, False)
    compare(cd, dec, cha, msg, False, False)
    dec.close()


@pytest.mark.parametrize("name,B,snr", [("n500_q4", 700, 1.9), ("reg36_n10000_q4", 300, 1.9), ("c5_chklut", 100, 4.2)])
def test_resident_with_table_composition(name, B, snr, monkeypatch):
    """LUTLDPC_COMPOSE=1: the node programs of the composed trees (three-input look-ups into 4 KB tables: fewer look-ups, more bank
    conflicts -- measured slower, off by default) -- same bits, same iteration codes."""
    monkeypatch.setenv("LUTLDPC_COMPOSE", "1")
    monkeypatch.setenv("LUTLDPC_RESIDENT", "2")                # (wide CHKTREE checks go through the streaming kernels unless forced)
    cd = oracle_codec(name)
    dec = product_decoder(cd)
    assert dec.describe()["resident"] == 1
    cha, msg, _ = awgn_labels(cd, B, snr, seed=5, mode=1 if name.startswith("c5") else 0)
    compare(cd, dec, cha, msg, True, True)
    compare(cd, dec, cha, msg, False, False)
    dec.close()


def test_resident_large_batch_and_fewer_iterations():
    """40 frame groups in one launch (2560 sets), then the same decoder with a smaller iteration budget whose last tree set is a
    decision set (set_exit_conditions), then a batch that re-uses the larger buffers."""
    cd = oracle_codec("n500_q4")
    dec = product_decoder(cd)
    B = 512 * 40 - 3
    cha, msg, _ = awgn_labels(cd, B, 2.0, seed=40)
    it = compare(cd, dec, cha, msg, True, True, flat=True)
    assert len(set(it.tolist())) > 8
    compare(cd, dec, cha[:900], msg[:900], False, False, flat=True)
    dec.close()


def test_resident_is_refused_for_codes_beyond_the_lds_and_can_be_switched_off(monkeypatch):
    cd = oracle_codec("dvbs2_q4_i6")
    dec = product_decoder(cd)
    assert dec.describe()["resident"] == 0 and dec.describe()["skewed_pipeline"] == 1
    dec.close()
    monkeypatch.setenv("LUTLDPC_RESIDENT", "0")
    cd = oracle_codec("n500_q4")
    dec = product_decoder(cd)
    assert dec.describe()["resident"] == 0
    dec.close()
