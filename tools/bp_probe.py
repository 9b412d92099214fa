#!/usr/bin/env python3
"""Throughput of the [BP] comparison decoder and of its host AWGN front end (for the record in DESIGN.md 7.2)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lut_ldpc_amd as L
from lut_ldpc_amd.bp import awgn_llr

for alist, B, it in [("rate0.50_irreg_dvbs2_N64800", 2048, 30), ("rate0.50_dv03_dc06_N10000", 8192, 30), ("rate0.50_dv03_dc06_N1000", 32768, 30)]:
    cd = L.Codec(ROOT / "data" / "codes" / f"{alist}.alist", known_rank=1, device=-1)       # (the graph arrays only)
    dv, dc, cn = cd.graph()

    class code:
        nvar, nchk = cd.nvar, cd.nchk
    dec = L.BPDecoder(cd.nvar, cd.nchk, dv, dc, cn, device=0)
    dec.set_exit_conditions(it, False, False)
    t0 = time.perf_counter(); llr, _ = awgn_llr(1, 0, 0, min(B, 256), code.nvar, 0.78); t_host = (time.perf_counter() - t0) / min(B, 256)
    llr = np.tile(llr, (B // len(llr) + 1, 1))[:B]
    dec.decode_llr_batch(llr)
    t0 = time.perf_counter(); dec.decode_llr_batch(llr); dt = time.perf_counter() - t0
    print(f"{alist}: B={B} {it} iterations fixed: {B / dt:.0f} codewords/s incl. PCIe ({dt * 1e3:.0f} ms); host AWGN 1 thread {1 / t_host:.0f} frames/s")
    dec.close()
