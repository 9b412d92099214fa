#!/usr/bin/env python3
"""How much group-level work skipping would sorting the frames of a batch by a cheap difficulty predictor buy?  Decodes one
batch in as-shipped mode on the GPU (device sampler: per-frame iteration counts and uncoded error counts) and compares
  ideal   = sum over frames of the iterations each one needs,
  today   = sum over 512-frame groups (in arrival order) of the group's slowest frame,
  sorted  = the same with the frames sorted by the predictor (uncoded bit errors ~ initial syndrome weight)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench, lut_ldpc_amd as L  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "dvbs2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
alist, sigma, max_iter, qc, qm, _, extra, rank = bench.WORKLOADS[wl]
cd = L.Codec(ROOT / "data" / "codes" / f"{alist}.alist", known_rank=rank, device=0)
cd.design_luts(sigma2=sigma * sigma, max_iters=max_iter, nq_cha=1 << qc, nq_msg=1 << qm, **extra)
cd.set_exit_conditions(max_iter, True, True)
base = -10 * np.log10(2 * cd.rate * sigma * sigma)
for d in (0.2, 0.4, 0.8):
    st = cd.sim_batch(base + d, 7, 0, 0, B)
    it, unc = np.abs(st[:, 0]).astype(np.int64), st[:, 3].astype(np.int64)
    rho = np.corrcoef(it, unc)[0, 1]
    grp = lambda x: x.reshape(-1, 512).max(axis=1).sum() * 512
    order = np.argsort(unc, kind="stable")
    print(f"{wl} Eb/N0 {base + d:.2f}: mean it {it.mean():.1f}  corr(iters, uncoded errors) {rho:.3f}  work/frame: ideal {it.mean():.1f}  "
          f"today {grp(it) / B:.1f}  sorted-by-errors {grp(it[order]) / B:.1f}  sorted-by-iters(oracle) {grp(np.sort(it)) / B:.1f}")
