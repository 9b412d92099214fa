"""Monte-Carlo driver on the GPU against the oracle's frame-by-frame loop: device sampler, sim_batch,
the C++ LDPC_BER_Sim_LUT (ber_sim) end to end, results file."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

import lut_ldpc_amd as L
from helpers import CODES, ROOT, TREES, oracle_codec
from itfile_reader import itload
from oracle import oracle as orc
from test_host_design_parity import product_codec

pytestmark = pytest.mark.gpu


def test_device_sampler_matches_oracle():
    ocd = oracle_codec("n500_q4_i8")
    pcd = product_codec("n500_q4_i8", device=0)
    for snr, seed, stream, f0, B in [(1.5, 3, 0, 0, 300), (4.0, 2 ** 40 + 5, 7, 2 ** 33, 17)]:
        want_cha, want_msg, _ = ocd.sample_labels(snr, 0.5, seed, stream, f0, B)
        got_cha, got_msg, _ = pcd.sample_labels(snr, seed, stream, f0, B)
        assert (want_cha == got_cha).all() and (want_msg == got_msg).all()
    pcd.close()


def _oracle_for(pcd, min_lut=True, mode=0):
    """Oracle codec on the product's (possibly column-permuted) graph with the product's tables."""
    dv, dc, cn = pcd.graph()
    code = orc.Code(graph=(pcd.nvar, pcd.nchk, dv, dc, cn))
    oc = orc.Codec(code, skip_rank=True)
    oc.set_rank(pcd.rank)
    return code, oc


@pytest.mark.parametrize("zero_codeword", [True, False])
def test_sim_batch_matches_oracle_frame_loop(zero_codeword):
    I = 10
    pcd = L.Codec(CODES / "rate0.50_dv02-17_dc08-09_lut_q4_N500.alist", with_generator=not zero_codeword, device=0)
    pcd.design_luts(sigma2=0.88 ** 2, max_iters=I)
    pcd.set_exit_conditions(I, True, True)
    code, oc = _oracle_for(pcd)
    oc.set_trees_txt(pcd.var_trees_txt, "", I, np.zeros(I, np.uint8), 16, np.full(I, 16, np.int32), True)
    oc.set_exit_conditions(I, True, True)
    # the oracle needs the boundaries for its own cell table: design them on its side (same ensemble)
    ref = orc.Codec(code, skip_rank=True); ref.design_luts(sigma2=0.88 ** 2, max_iters=I)
    assert ref.var_tree_txt == pcd.var_trees_txt
    snr, seed, stream, B = 1.8, 12, 2, 333
    cw = None
    if not zero_codeword:
        _, _, cw = pcd.sample_labels(snr, seed, stream, 0, B, zero_codeword=False)
        assert cw.any()
    want_c, want_per, _ = ref.sim_snr_point(snr, 0.5, pcd.ninfo, seed, stream, B, nfers=10 ** 9, codewords=cw)
    got = pcd.sim_batch(snr, seed, stream, 0, B, zero_codeword=zero_codeword)
    assert (got == want_per).all(), np.argwhere(got != want_per)[:5]
    pcd.close()


def _setup_basedir(tmp_path):
    (tmp_path / "codes").mkdir()
    (tmp_path / "trees").mkdir()
    for f in CODES.glob("*N500.alist"):
        shutil.copy(f, tmp_path / "codes" / f.name)
    shutil.copy(CODES / "rate0.84_reg_v6c32_N2048.alist", tmp_path / "codes")
    for f in TREES.glob("*.ini"):
        shutil.copy(f, tmp_path / "trees" / f.name)
    return tmp_path


def test_ber_sim_irregular_example_matches_oracle(tmp_path):
    """BASELINE config 1: ber_sim -p params/ber.ini.irregular.example (N500, Nframes=100, SNR 0:.5:4, non-zero codewords)."""
    from lut_ldpc_amd._capi import lib, check
    import ctypes as C
    base = _setup_basedir(tmp_path)
    params = ROOT / "data" / "params" / "ber.ini.irregular.example"
    snr = (C.c_double * 32)(); cnt = (C.c_int64 * 160)()
    n = lib.lutldpc_ber_sim_run(str(params).encode(), str(base).encode(), 0, b"", 0, 1, 1, snr, cnt, 32)
    check(min(n, 0))
    assert n == 9 and list(snr[:n]) == [0, .5, 1, 1.5, 2, 2.5, 3, 3.5, 4]
    got = np.array(cnt[:n * 5]).reshape(n, 5)
    # the same run, frame by frame, with the oracle decoder on the same code / tables / codewords
    pcd = L.Codec(CODES / "rate0.50_dv02-17_dc08-09_lut_q4_N500.alist", with_generator=True, device=0)
    pcd.design_luts(sigma2=0.88 ** 2, max_iters=50)
    dv, dc, cn = pcd.graph()
    code = orc.Code(graph=(pcd.nvar, pcd.nchk, dv, dc, cn))
    ref = orc.Codec(code, skip_rank=True); ref.set_rank(250)
    ref.design_luts(sigma2=0.88 ** 2, max_iters=50)
    ref.set_exit_conditions(50, True, True)               # psc AND pisc (src/LDPC_BER_Sim.cpp:500)
    stop = False
    for i in range(n):
        if stop:
            assert (got[i] == 0).all()                     # padded points (src/LDPC_BER_Sim.cpp:142-149)
            continue
        _, _, cw = pcd.sample_labels(snr[i], 0, i, 0, 100, zero_codeword=False)
        want, _, stop = ref.sim_snr_point(snr[i], 0.5, 250, 0, i, 100, nfers=20, codewords=cw)
        assert (got[i] == want).all(), (i, got[i], want)
    assert got[0][0] == 21 and got[0][2] == 21              # at 0 dB every frame fails: stops after Nfers+1 errors
    # results file: name as README.md:114, readable like scripts/itload.m
    res = base / "results" / "RES_N500_R0.5_maxIter50_zcw0_frames100_minLUT" / "RES_N500_R0.5_maxIter50_zcw0_frames100_minLUT_rseed0000.it"
    assert res.exists()
    it = itload(res)
    assert (it["sim_Nframes"] == got[:, 0]).all() and (it["sim_data_bit_errors"] == got[:, 3]).all()
    assert it["ldpc_nvar"][0] == 500 and it["ldpc_nchk"][0] == 250
    assert (res.parent / "lut_codec.it").exists() and (res.parent / "ber.ini.irregular.example").exists()
    pcd.close()


def test_ber_sim_binary_regular_example(tmp_path):
    """BASELINE config 5 through the command line: (6,32) code, file trees, 3-bit, QCHA, output_verbosity=1."""
    base = _setup_basedir(tmp_path)
    exe = ROOT / "lut_ldpc_amd" / "lib" / "ber_sim"
    r = subprocess.run([str(exe), "-p", str(ROOT / "data" / "params" / "ber.ini.regular.example"), "-b", str(base), "-s", "3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Stimuli Pair" in r.stdout and r.stdout.count("SNR = ") >= 1
    out = list((base / "results").glob("RES_N2048_R0.841309_maxIter8_zcw0_frames20_minLUT/*_rseed0003.it"))     # README.md:239
    assert len(out) == 1
    it = itload(out[0])
    assert len(it["sim_SNRdB"]) == 7 and it["sim_Nframes"][0] > 0


def test_python_driver_equals_cpp_driver(tmp_path):
    from lut_ldpc_amd import ber_sim
    base = _setup_basedir(tmp_path)
    params = ROOT / "data" / "params" / "ber.ini.irregular.example"
    pts, path = ber_sim.run(params, base, seed=1, quiet=True)
    from lut_ldpc_amd._capi import lib
    import ctypes as C
    snr = (C.c_double * 32)(); cnt = (C.c_int64 * 160)()
    n = lib.lutldpc_ber_sim_run(str(params).encode(), str(base).encode(), 1, b"_cpp", 0, 0, 1, snr, cnt, 32)
    assert n == len(pts)
    assert (np.array(cnt[:n * 5]).reshape(n, 5) == np.array([c for _, c in pts])).all()
    assert path and itload(path)["sim_SNRdB"][1] == 0.5


def test_config4_sweep_on_one_gpu_at_reduced_nframes(tmp_path):
    """BASELINE config 4 (data/params/ber.ini.dvbs2_sweep) through lut_ldpc_amd.ber_sim.run on one GPU, Nframes reduced:
    the sweep stops and pads like the reference, and frames of a mid-SNR point equal the oracle's on the same Philox frames."""
    import re
    from lut_ldpc_amd import ber_sim
    base = tmp_path / "base"
    (base / "codes").mkdir(parents=True)
    shutil.copy(CODES / "rate0.50_irreg_dvbs2_N64800.alist", base / "codes")
    txt = (ROOT / "data" / "params" / "ber.ini.dvbs2_sweep").read_text()
    params = tmp_path / "ber.ini.dvbs2_sweep"
    params.write_text(re.sub(r"Nframes\s*=\s*1e6", "Nframes  = 3000", txt))
    pts, path = ber_sim.run(params, base, seed=5, quiet=True)
    cnt = np.array([c for _, c in pts])
    assert cnt[0][0] == 21 and cnt[0][2] == 21                     # 0 dB: stops after Nfers + 1 frame errors (:289)
    stop = int(np.flatnonzero(cnt[:, 2] == 0)[0])
    assert cnt[stop][0] == 3000 and (cnt[stop + 1:] == 0).all()    # first error-free point ends the sweep (:307), the rest is padded
    assert itload(path)["sim_Nframes"].tolist() == cnt[:, 0].tolist()
    # the oracle on 24 frames of the point before the stop (a mix of early exits and frames that run all 50 iterations)
    sim = ber_sim.BerSim(params, base, 5, "", 0)
    idx = stop - 1
    got = sim.batch(idx, 100, 24)
    sim.close()
    cd = oracle_codec("dvbs2_q4")
    cd.set_exit_conditions(50, True, True)
    cha, msg, unc = cd.sample_labels(pts[idx][0], 0.5, 5, idx, 100, 24)
    bits, it = cd.lut_decode_batch_flat(cha, msg)
    be = bits[:, :32400].sum(1)
    want = np.stack([it, be > 0, be, unc], 1).astype(np.int32)
    assert (got == want).all(), np.argwhere(got != want)[:5]


def test_config4_two_ranks_sharing_the_gpu_equal_one_rank(tmp_path):
    """The multi-rank ber_sim command line on the GPU: `torch.distributed.run --nproc-per-node 2 -m lut_ldpc_amd.ber_sim` with
    both ranks on this box's one card (LUTLDPC_DIST_BACKEND=gloo; RCCL needs one card per rank) writes the same result file
    as the single-process run -- frames dealt to the ranks, counters exchanged, stop rule applied in global frame order."""
    import re
    base = tmp_path / "base"
    (base / "codes").mkdir(parents=True)
    shutil.copy(CODES / "rate0.50_irreg_dvbs2_N64800.alist", base / "codes")
    txt = (ROOT / "data" / "params" / "ber.ini.dvbs2_sweep").read_text()
    params = tmp_path / "ber.ini.dvbs2_sweep"
    params.write_text(re.sub(r"Nframes\s*=\s*1e6", "Nframes  = 1500", txt))
    env = dict(os.environ, PYTHONPATH=str(ROOT), LUTLDPC_DIST_BACKEND="gloo")
    one = subprocess.run([sys.executable, "-m", "lut_ldpc_amd.ber_sim", "-p", str(params), "-b", str(base), "-s", "3", "-c", "_one"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", "-m", "lut_ldpc_amd.ber_sim", "-p", str(params), "-b", str(base), "-s", "3", "-c", "_two"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    res = sorted((base / "results").rglob("*.it"))
    assert len(res) == 2, res
    a, b = itload(res[0]), itload(res[1])
    for k in ("sim_SNRdB", "sim_Nframes", "sim_Ndatabits", "sim_frame_errors", "sim_data_bit_errors", "sim_uncoded_bit_errors"):
        assert a[k].tolist() == b[k].tolist(), k
    assert sum(a["sim_Nframes"].tolist()) > 1500


def test_cpp_multi_rank_ber_sim_equals_the_single_rank_result_file(tmp_path):
    """The native multi-device ber_sim (ber_sim_multi.cpp) on the one GPU of the box: ranks that SHARE the device (`-d 0,0,0`: three
    ranks, counters combined on the host -- RCCL refuses ranks on one device) and the default two lanes on one device write the
    same result file as the plain single-rank loop: Philox-addressed frames + the stop rule applied in global frame order
    (src/LDPC_BER_Sim.cpp:289).  Nframes is raised so that several rounds of several ranks happen and low-SNR points stop early
    in the middle of a round."""
    base = _setup_basedir(tmp_path)
    ini = (ROOT / "data" / "params" / "ber.ini.irregular.example").read_text().replace("Nframes  = 1e2", "Nframes  = 7000\n   batch_frames = 1024")
    params = tmp_path / "ber_multi.ini"
    params.write_text(ini)
    exe = ROOT / "lut_ldpc_amd" / "lib" / "ber_sim"
    files = {}
    for tag, extra in (("_one", ["-d", "0", "--lanes", "1"]), ("_lanes", ["-d", "0"]), ("_three", ["-d", "0,0,0", "--lanes", "1", "--exchange", "host"])):
        r = subprocess.run([str(exe), "-p", str(params), "-b", str(base), "-s", "4", "-c", tag] + extra, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        out = sorted((base / "results").glob(f"*{tag}/*_rseed0004.it"))
        assert len(out) == 1
        files[tag] = itload(out[0])
        if tag != "_one":
            assert "counters over the host" in r.stdout, r.stdout[-500:]
    a = files["_one"]
    assert a["sim_Nframes"][0] < 7000 and a["sim_frame_errors"][0] == 21             # 0 dB: stops after Nfers + 1 frame errors
    assert a["sim_Nframes"].max() == 7000                                             # a high-SNR point runs to Nframes
    for tag in ("_lanes", "_three"):
        for k in ("sim_SNRdB", "sim_Nframes", "sim_Ndatabits", "sim_frame_errors", "sim_data_bit_errors", "sim_uncoded_bit_errors"):
            assert (np.asarray(files[tag][k]) == np.asarray(a[k])).all(), (tag, k, files[tag][k], a[k])


def test_counter_exchange_through_rccl_proper_with_one_rank(tmp_path):
    """RCCL itself on the one GPU of the box (it refuses two ranks on a card, so ONE rank): the C++ ber_sim with `--exchange rccl`
    (dlopen of librccl, ncclCommInitAll, ncclAllGather / ncclAllReduce of int64 on the device's stream) and the Python driver under
    torch.distributed.run with a one-rank nccl process group (LUTLDPC_DIST_FORCE=1) write the same result file as the plain loop."""
    base = _setup_basedir(tmp_path)
    ini = (ROOT / "data" / "params" / "ber.ini.irregular.example").read_text().replace("Nframes  = 1e2", "Nframes  = 5000\n   batch_frames = 1024")
    params = tmp_path / "ber_rccl.ini"
    params.write_text(ini)
    exe = ROOT / "lut_ldpc_amd" / "lib" / "ber_sim"
    keys = ("sim_SNRdB", "sim_Nframes", "sim_Ndatabits", "sim_frame_errors", "sim_data_bit_errors", "sim_uncoded_bit_errors")
    files = {}
    for tag, extra in (("_plain", ["-d", "0", "--lanes", "1"]), ("_rccl", ["-d", "0", "--lanes", "2", "--exchange", "rccl"])):
        r = subprocess.run([str(exe), "-p", str(params), "-b", str(base), "-s", "6", "-c", tag] + extra, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        if tag == "_rccl":
            assert "1 device(s) x 2 lane(s), counters over RCCL" in r.stdout, r.stdout[-500:]
        out = sorted((base / "results").glob(f"*{tag}/*_rseed0006.it"))
        assert len(out) == 1
        files[tag] = itload(out[0])
    env = dict(os.environ, PYTHONPATH=str(ROOT), LUTLDPC_DIST_FORCE="1")
    env.pop("LUTLDPC_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", "-m", "lut_ldpc_amd.ber_sim", "-p", str(params), "-b", str(base), "-s", "6", "-c", "_pyrccl"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = sorted((base / "results").glob("*_pyrccl/*_rseed0006.it"))
    assert len(out) == 1
    files["_pyrccl"] = itload(out[0])
    assert files["_plain"]["sim_Nframes"].max() == 5000
    for tag in ("_rccl", "_pyrccl"):
        for k in keys:
            assert (np.asarray(files[tag][k]) == np.asarray(files["_plain"][k])).all(), (tag, k)
