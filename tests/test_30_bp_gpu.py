"""[BP] comparison decoder on the GPU against the oracle's statement of the same specification (include/lut_ldpc_bp.h):
every decided bit, every output QLLR and every return code, jac-log table and min-sum, all exit modes, check degrees 3 to 32
(the spelled-out orders of degrees 3..6 and the general left/right partial-sum form), a degree-1 variable node (DVB-S2), ragged batches.  PARITY UNPINNED against the
reference's forked IT++ (absent): this pins the two implementations of this repository to each other."""
import numpy as np
import pytest

import lut_ldpc_amd as L
from helpers import CODES, write_ira_alist
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _llr(code, B, snr_db, seed, rate):
    rng = np.random.default_rng(seed)
    N0 = 10 ** (-snr_db / 10) / rate
    return 4 * (1.0 + rng.normal(0.0, np.sqrt(N0 / 2), (B, code.nvar))) / N0


@pytest.mark.parametrize("alist,rate,snr,B,iters", [("rate0.50_dv03_dc06_N1000", 0.5, 1.4, 300, 25),
                                                    ("rate0.50_dv02-17_dc08-09_lut_q4_N500", 0.5, 1.6, 257, 30),
                                                    ("rate0.84_reg_v6c32_N2048", 0.84, 3.6, 64, 12)])
@pytest.mark.parametrize("d", [(12, 300, 7, 28), (12, 0, 7, 28), (8, 64, 4, 16)])
def test_bp_matches_oracle(alist, rate, snr, B, iters, d):
    code = orc.Code(CODES / f"{alist}.alist")
    ref = orc.BP(code, *d)
    dec = L.BPDecoder(code.nvar, code.nchk, code.dv, code.dc, code.cn_msg_idx, *d, device=0)
    assert (dec.logexp_table() == ref.table()).all()
    llr = _llr(code, B, snr, seed=B, rate=rate)
    llr[0] = 9.0                                             # noise-free frame: pisc returns 0
    for psc, pisc in [(True, True), (True, False), (False, False)]:
        ref.set_exit_conditions(iters, psc, pisc)
        dec.set_exit_conditions(iters, psc, pisc)
        wb, wi, wq = ref.decode_llr_batch(llr)
        gb, gi, gq = dec.decode_llr_batch(llr, want_qllr=True)
        assert (gi == wi).all(), (np.flatnonzero(gi != wi)[:8], gi[:8], wi[:8])
        assert (gq == wq).all() and (gb == wb).all()
        if psc:
            assert (wi > 0).sum() > 0 and len(set(wi.tolist())) > 2
    # quantised entry point
    q = np.clip(np.floor(0.5 + llr * 2.0 ** d[0]), -(2 ** (d[3] - 1) - 1), 2 ** (d[3] - 1) - 1).astype(np.int32)
    gb2, gi2 = dec.decode_qllr_batch(q)
    assert (gb2 == gb).all() and (gi2 == gi).all()
    dec.close()


@pytest.mark.parametrize("K,dc", [(100, 3), (200, 4), (300, 5), (400, 6), (500, 7)])
def test_bp_spelled_out_check_degrees(tmp_path, K, dc):
    """Check degrees 3..6 (association orders written out, include/lut_ldpc_bp.h) and 7 (first general one), jac-log table."""
    N, M = write_ira_alist(tmp_path / "c.alist", K, 300, 3, seed=dc)
    code = orc.Code(tmp_path / "c.alist")
    assert set(code.dc.tolist()) == {dc}
    ref = orc.BP(code)
    dec = L.BPDecoder(code.nvar, code.nchk, code.dv, code.dc, code.cn_msg_idx, device=0)
    llr = _llr(code, 130, 2.0 + 0.5 * (7 - dc), seed=dc, rate=K / N)
    for psc in (True, False):
        ref.set_exit_conditions(20, psc, psc); dec.set_exit_conditions(20, psc, psc)
        wb, wi, wq = ref.decode_llr_batch(llr)
        gb, gi, gq = dec.decode_llr_batch(llr, want_qllr=True)
        assert (gi == wi).all() and (gq == wq).all() and (gb == wb).all()
    dec.close()


def test_bp_on_the_dvbs2_code():
    """N = 64800 with its degree-1 variable node and check degrees 6 / 7, fewer frames than a tile."""
    code = orc.Code(CODES / "rate0.50_irreg_dvbs2_N64800.alist")
    ref = orc.BP(code)
    dec = L.BPDecoder(code.nvar, code.nchk, code.dv, code.dc, code.cn_msg_idx, device=0)
    llr = _llr(code, 5, 1.2, seed=8, rate=0.5)
    ref.set_exit_conditions(30, True, True); dec.set_exit_conditions(30, True, True)
    wb, wi, wq = ref.decode_llr_batch(llr)
    gb, gi, gq = dec.decode_llr_batch(llr, want_qllr=True)
    assert (gi == wi).all() and (gq == wq).all() and (gb == wb).all()
    dec.close()


def test_ber_sim_bp_section_matches_oracle_frame_loop(tmp_path):
    """ber_sim on a parameter file with a [BP] section (data/params/ber.ini.bp.example, Nframes reduced): the C++ driver
    (LDPC_BER_Sim_BP: host AWGN front end + device decoder), the sharded Python driver and the oracle's frame-by-frame loop
    (same Philox-addressed noise, same stop rule, src/LDPC_BER_Sim.cpp:246-311) give the same five counters per SNR point."""
    import ctypes as C
    import re
    import shutil
    from helpers import ROOT
    from itfile_reader import itload
    from lut_ldpc_amd._capi import lib, check
    from lut_ldpc_amd import ber_sim
    base = tmp_path / "base"
    (base / "codes").mkdir(parents=True)
    shutil.copy(CODES / "rate0.50_dv03_dc06_N1000.alist", base / "codes")
    txt = (ROOT / "data" / "params" / "ber.ini.bp.example").read_text()
    params = tmp_path / "ber.ini.bp.example"
    params.write_text(re.sub(r"Nframes\s*=\s*2000", "Nframes  = 150", txt))
    snr = (C.c_double * 32)(); cnt = (C.c_int64 * 160)()
    n = lib.lutldpc_ber_sim_run(str(params).encode(), str(base).encode(), 2, b"", 0, 1, 1, snr, cnt, 32)
    check(min(n, 0))
    assert n == 7 and list(snr[:n]) == [0, .5, 1, 1.5, 2, 2.5, 3]
    got = np.array(cnt[:n * 5]).reshape(n, 5)
    # the oracle, frame by frame
    code = orc.Code(CODES / "rate0.50_dv03_dc06_N1000.alist")
    ref = orc.BP(code, 12, 300, 7, 28)
    ref.set_exit_conditions(30, True, True)
    K, stop = 500, False
    for i in range(n):
        if stop:
            assert (got[i] == 0).all()
            continue
        N0 = 10 ** (-snr[i] / 10) / 0.5
        llr, unc = orc.awgn_llr(2, i, 0, 150, code.nvar, N0)
        bits, it, _ = ref.decode_llr_batch(llr)
        be = bits[:, :K].sum(1)
        fe = np.cumsum(be > 0)
        hit = np.flatnonzero(fe > 20)
        m = int(hit[0]) + 1 if hit.size else 150
        want = np.array([m, m * K, (be[:m] > 0).sum(), be[:m].sum(), unc[:m].sum()])
        assert (got[i] == want).all(), (i, got[i], want)
        ber, fer = want[3] / want[1], want[2] / want[0]
        stop = ber < 1e-7 or fer < 1e-5
    assert got[0][2] == 21                                    # 0 dB: BP fails too, stops after Nfers + 1 frame errors
    res = list((base / "results").glob("RES_N1000_R0.5_maxIter30_zcw1_frames150/*_rseed0002.it"))     # base-class naming, :104-115
    assert len(res) == 1 and (itload(res[0])["sim_Nframes"] == got[:, 0]).all()
    pts, _ = ber_sim.run(params, base, seed=2, custom_name="_py", quiet=True)
    assert (np.array([c for _, c in pts]) == got).all()
