#!/usr/bin/env python3
"""Fill the design cache (data/design_cache/, LUTLDPC_DESIGN_CACHE) with the LUT designs of bench.py's workloads: host
code only, no GPU.  The files travel to the GPU box with the tree, so a bench / profile run there starts in seconds
instead of re-running the 50-iteration density evolution and the GF(2) rank of the parity-check matrix (about a second per workload on this host).  Usage: tools/warm_design_cache.py [workload...]"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("LUTLDPC_DESIGN_CACHE", str(ROOT / "data" / "design_cache"))
Path(os.environ["LUTLDPC_DESIGN_CACHE"]).mkdir(parents=True, exist_ok=True)

import bench  # noqa: E402
import lut_ldpc_amd as L  # noqa: E402

for wl in (sys.argv[1:] or sorted(bench.WORKLOADS)):
    alist, sigma, max_iter, qc, qm, _, extra, rank = bench.WORKLOADS[wl]
    t0 = time.perf_counter()
    cd = L.Codec(ROOT / "data" / "codes" / f"{alist}.alist", known_rank=rank, device=-1)
    cd.design_luts(sigma2=sigma * sigma, max_iters=max_iter, nq_cha=1 << qc, nq_msg=1 << qm, **extra)
    print(f"{wl}: {'cached' if cd.design_from_cache else 'designed'} in {time.perf_counter() - t0:.1f} s")
    cd.close()
