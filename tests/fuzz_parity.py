#!/usr/bin/env python3
"""Randomised parity run on the GPU box (test infrastructure: uses the oracle as the checker, like tests/): random irregular
graphs, alphabets, iteration counts, batch sizes and exit modes through the default path, every bit and iteration code
against the oracle.  Usage: python tests/fuzz_parity.py [cases] [seed]"""
import os, pathlib, sys, tempfile
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))   # (this file lives in tests/: only tests may use the oracle)
from helpers import awgn_labels, compare, product_decoder, write_random_alist     # noqa: E402
from oracle import oracle as orc                                                   # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
COMPACT = "--compact" in sys.argv      # many frame groups, compaction of the surviving frames forced on with random check points
BP = "--bp" in sys.argv                # the [BP] comparison decoder instead of the LUT decoder (GPU against the oracle's statement of it)
GEOM = "--resident-geom" in sys.argv   # LDS-resident decoder (the default for these code sizes) with a random workgroup geometry per case
if COMPACT:
    os.environ["LUTLDPC_RESIDENT"] = "0"   # (compaction belongs to the streaming kernels)
bad = 0
for c in range(cases):
    N = int(rng.integers(200, 1600))
    rate = float(rng.choice([0.5, 0.6, 0.75]))
    M = max(40, int(N * (1 - rate)))
    k = int(rng.integers(1, 4))
    dvc = sorted(rng.choice([2, 3, 4, 5, 6, 8, 9, 12], size=k, replace=False).tolist())
    if dvc == [2]:
        dvc = [2, 3]
    p = rng.dirichlet(np.ones(len(dvc)) * 2)
    nqc, nqm = int(rng.choice([8, 16])), int(rng.choice([8, 16]))
    I = int(rng.integers(3, 12))
    B = int(rng.choice([1, 63, 257, 513, 700, 1100, 1537]))
    if COMPACT:
        N = min(N, 700)
        B = int(rng.integers(4, 66)) * 512 - int(rng.integers(0, 400))
        os.environ.update(LUTLDPC_COMPACT="1", LUTLDPC_COMPACT_MARGIN="0", LUTLDPC_COMPACT_FIRST=str(int(rng.integers(1, 5))),
                          LUTLDPC_COMPACT_EVERY=str(int(rng.integers(1, 4))), LUTLDPC_COMPACT_KEEP=str(int(rng.integers(0, 2))))
    free_frames = rng.integers(0, B, 3)
    sig = float(rng.uniform(0.45, 0.9))
    snr_off = float(rng.uniform(-0.3, 1.5))
    draws = rng.integers(0, 4, 3), int(rng.integers(1, max(2, I - 1))), rng.integers(0, 2, I)
    if c < int(os.environ.get("FUZZ_FROM", "0")):
        continue
    d = pathlib.Path(tempfile.mkdtemp())
    try:
        dv, dc = write_random_alist(d / "r.alist", N, M, dvc, p.tolist(), seed=1000 + c)
    except AssertionError:
        print(f"case {c}: graph generation gave up, skipped"); continue
    if dc.max() > 32 or dv.max() > 20:
        print(f"case {c}: degrees outside the compile-time kernels, skipped"); continue
    code = orc.Code(d / "r.alist")
    if BP:
        import lut_ldpc_amd as L
        dpar = [(12, 300, 7, 28), (12, 0, 7, 28), (8, 64, 4, 16), (10, 128, 5, 20)][int(draws[0][0])]
        ref = orc.BP(code, *dpar)
        gpu = L.BPDecoder(code.nvar, code.nchk, code.dv, code.dc, code.cn_msg_idx, *dpar, device=0)
        Bb = min(B, 300)
        r2 = np.random.default_rng(c)
        N0 = 10 ** (-(2.0 + 4 * sig) / 10) / (1.0 - M / N)
        llr = 4 * (1.0 + r2.normal(0.0, np.sqrt(N0 / 2), (Bb, N))) / N0
        llr[0] = 9.0
        okc = True
        codes = None
        for psc, pisc in [(True, True), (True, False), (False, False)]:
            ref.set_exit_conditions(I + 8, psc, pisc); gpu.set_exit_conditions(I + 8, psc, pisc)
            wb, wi, wq = ref.decode_llr_batch(llr)
            gb, gi, gq = gpu.decode_llr_batch(llr, want_qllr=True)
            okc = okc and bool((gi == wi).all() and (gq == wq).all() and (gb == wb).all())
            codes = codes if codes is not None else sorted(set(wi.tolist()))
        gpu.close()
        bad += 0 if okc else 1
        print(f"case {c}: BP N={N} M={M} dv={sorted(set(dv.tolist()))} dc={sorted(set(dc.tolist()))} d={dpar} B={Bb} iters<={I + 8} {'ok' if okc else 'MISMATCH'}, return codes with the syndrome test {codes[:8]}", flush=True)
        continue
    cd = orc.Codec(code, skip_rank=True); cd.set_rank(M); cd.rate = 1.0 - M / N
    # one case in four each: check-node LUT trees instead of min-sum, a message alphabet that shrinks along the iterations,
    # LUT stages reused over several iterations (src/LDPC_Code_LUT.cpp:120-169)
    extra = {}
    nq_vec = np.full(I, nqm, np.int32)
    if draws[0][0] == 0 and dc.max() <= 12:
        extra["min_lut"] = False
    if draws[0][1] == 0 and nqm == 16 and I >= 4:
        nq_vec[min(draws[1], I - 2):] = 8
    if draws[0][2] == 0:
        reuse = draws[2].astype(np.int32); reuse[0] = 0; reuse[-1] = 0     # (first and last iteration are exempt: src/LDPC_Code_LUT.cpp:122)
        for i in range(1, I):                                  # a reused stage keeps the alphabets of the stage it repeats
            if nq_vec[i] != nq_vec[i - 1] or (i + 1 < I and nq_vec[i + 1] != nq_vec[i]):
                reuse[i] = 0
        extra["reuse_vec"] = reuse.tolist()
    cd.design_luts(sigma2=sig ** 2, max_iters=I, nq_msg=nq_vec, nq_cha=nqc, **extra)
    snr = -10 * np.log10(2 * cd.rate * sig * sig) + snr_off
    cha, msg, _ = awgn_labels(cd, B, snr, seed=c)
    if COMPACT:                                                # a few noise-free frames: they pass the test on the channel decisions
        for f in free_frames:
            cha[f] = nqc - 1; msg[f] = nq_vec[0] - 1
    if os.environ.get("FUZZ_ORACLE_ONLY"):                     # (debugging aid: the checker alone, on a machine without a GPU)
        for psc, pisc in [(True, True), (True, False), (False, False)]:
            cd.set_exit_conditions(I, psc, pisc)
            (cd.lut_decode_batch_flat if COMPACT else cd.lut_decode_batch)(cha, msg)
        print(f"case {c}: N={N} M={M} dv={dvc} dc={sorted(set(dc.tolist()))} nq={nqc}/{nq_vec.tolist()} I={I} B={B} {extra}: oracle alone ok", flush=True)
        continue
    if GEOM:
        r3 = np.random.default_rng(7000 + c)
        os.environ["LUTLDPC_RESIDENT_S"] = str(int(r3.integers(1, 9)))
        os.environ["LUTLDPC_RESIDENT_NT"] = str(int(r3.choice([256, 512, 768, 1024])))
        os.environ["LUTLDPC_RESIDENT"] = "2"
        os.environ["LUTLDPC_COMPOSE"] = str(int(r3.integers(0, 2)))
    try:
        dec = product_decoder(cd)
    except Exception as e:
        print(f"case {c}: decoder refused ({str(e)[:120]}), skipped"); continue
    try:
        for psc, pisc in [(True, True), (True, False), (False, False)]:
            it = compare(cd, dec, cha, msg, psc, pisc, flat=COMPACT)
        print(f"case {c}: N={N} M={M} dv={sorted(set(dv.tolist()))} dc={sorted(set(dc.tolist()))} nq={nqc}/{nqm} I={I} B={B} bucket={dec.describe()['fused_bucket']} skew={dec.describe()['skewed_pipeline']} resident={dec.describe()['resident']}{(' geom ' + os.environ['LUTLDPC_RESIDENT_S'] + 'x' + os.environ['LUTLDPC_RESIDENT_NT'] + ' compose ' + os.environ['LUTLDPC_COMPOSE']) if GEOM else ''} {extra if extra else ''} nq_msg={sorted(set(nq_vec.tolist()))} {('compaction ' + str(dec.describe()['compaction']) + ' first/every/keep ' + os.environ['LUTLDPC_COMPACT_FIRST'] + '/' + os.environ['LUTLDPC_COMPACT_EVERY'] + '/' + os.environ['LUTLDPC_COMPACT_KEEP']) if COMPACT else ''} ok, iteration codes {sorted(set(it.tolist()))[:5]}")
    except AssertionError as e:
        bad += 1
        print(f"case {c}: MISMATCH N={N} M={M} dv={dvc} nq={nqc}/{nqm} I={I} B={B}: {str(e)[:200]}")
    dec.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
