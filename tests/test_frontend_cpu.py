"""Monte-Carlo front end, host side: channel cell tables (product C++ vs oracle), the sampler's
statistics, the stop rule and the sharded frame loop (single process here, gloo in test_sharded_gloo)."""
import numpy as np
import pytest

import lut_ldpc_amd as L
from lut_ldpc_amd.ber_sim import Comm, sim_snr_point_sharded
from helpers import CODES, oracle_codec
from test_host_design_parity import product_codec


@pytest.mark.parametrize("name,mode", [("n500_q4_i8", 0), ("reg36_n1000_mixed", 0), ("c5_minlut", 1)])
@pytest.mark.parametrize("snr", [0.0, 2.5, 6.0])
def test_channel_cells_match_oracle(name, mode, snr):
    want_cd = oracle_codec(name)
    want_cd.set_initial_message_mode(mode)
    got_cd = product_codec(name)
    got_cd.set_initial_message_mode(mode)
    want = want_cd.channel_cells(snr, want_cd.rate)
    got = got_cd.channel_cells(snr)
    for k in want:
        assert (want[k] == got[k]).all(), k
    assert (np.diff(got["thr"].astype(np.float64)) >= 0).all()
    want_cd.set_initial_message_mode(0)
    got_cd.close()


def test_sampler_statistics():
    """Cell sampling reproduces BPSK/AWGN + quant_nonlin: label frequencies match the Gaussian cell masses and
    the slicer error rate matches Q(sqrt(2 R Eb/N0)) (src/LDPC_BER_Sim.cpp:248-279)."""
    from math import erfc, sqrt
    cd = oracle_codec("n500_q4_i8")
    snr, rate = 2.0, 0.5
    cha, msg, unc = cd.sample_labels(snr, rate, seed=11, stream=3, frame0=0, B=4000)
    N0 = 10 ** (-snr / 10) / rate
    sigma = sqrt(N0 / 2)
    t = np.concatenate([[-np.inf], cd.qb_cha * N0 / 4, [np.inf]])
    cdf = np.array([0.5 * erfc(-(x - 1) / sigma / sqrt(2)) for x in t])
    p = np.diff(cdf)
    freq = np.bincount(cha.ravel(), minlength=16) / cha.size
    assert np.abs(freq - p).max() < 4 * np.sqrt(p.max() / cha.size) + 1e-4
    assert abs(unc.sum() / cha.size - 0.5 * erfc(sqrt(2 * rate * 10 ** (snr / 10)) / sqrt(2))) < 2e-3
    # frames are addressable: any split of the range gives the same labels
    a, _, _ = cd.sample_labels(snr, rate, 11, 3, 100, 7)
    assert (a == cha[100:107]).all()
    # a sent 1 mirrors the labels
    ones = np.ones((5, 500), np.uint8)
    c1, m1, u1 = cd.sample_labels(snr, rate, 11, 3, 0, 5, codewords=ones)
    assert (c1 == 15 - cha[:5]).all() and (m1 == 15 - msg[:5]).all() and (u1 == unc[:5]).all()


def _fake_stats(n, seed, p_err):
    rng = np.random.default_rng(seed)
    fe = rng.random(n) < p_err
    st = np.zeros((n, 4), np.int32)
    st[:, 1] = fe
    st[:, 2] = fe * rng.integers(1, 40, n)
    st[:, 3] = rng.integers(0, 60, n)
    return st


def _sequential(st, K, nfers):
    c = np.zeros(5, np.int64)
    for row in st:
        c += [1, K, row[1] != 0, row[2], row[3]]
        if c[2] > nfers:
            break
    return c


@pytest.mark.parametrize("p_err,nframes,nfers", [(0.5, 1000, 20), (0.01, 5000, 20), (0.0, 700, 20), (1.0, 100, 0), (0.03, 999, 3)])
def test_stop_rule_batched_equals_sequential(p_err, nframes, nfers):
    st = _fake_stats(nframes, 1, p_err)
    want = _sequential(st, 250, nfers)
    got = sim_snr_point_sharded(lambda f, b: st[f:f + b], nframes, nfers, 250, Comm(), batch_max=512, batch_first=16)
    assert (got == want).all()


def test_info_bits_and_encoder_against_oracle_stream():
    from oracle import oracle as orc
    cd = L.Codec(CODES / "rate0.50_dv02-17_dc08-09_lut_q4_N500.alist", with_generator=True, device=-1)
    a = orc.info_bits(5, 2, 77, cd.ninfo)
    assert 0.35 < a.mean() < 0.65
    cw = cd.encode(a)
    dv, dc, cn = cd.graph()
    vn_of_edge = np.repeat(np.arange(cd.nvar), dv)
    p = 0
    for c in range(cd.nchk):
        assert cw[vn_of_edge[cn[p:p + dc[c]]]].sum() % 2 == 0
        p += dc[c]
    cd.close()
