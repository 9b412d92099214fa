#!/usr/bin/env python3
"""Histogram of the iteration at which frames leave the decoder in as-shipped mode (psc = pisc = 1), per workload and Eb/N0:
the input for every decision about compaction / early-exit work skipping.  Run on the GPU box."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import lut_ldpc_amd as L  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "dvbs2"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    alist, sigma, max_iter, qc, qm, _, extra, rank = bench.WORKLOADS[wl]
    cd = L.Codec(ROOT / "data" / "codes" / f"{alist}.alist", known_rank=rank, device=0)
    cd.design_luts(sigma2=sigma * sigma, max_iters=max_iter, nq_cha=1 << qc, nq_msg=1 << qm, **extra)
    cd.set_exit_conditions(max_iter, True, True)
    dec = cd.decoder()
    base = -10 * np.log10(2 * cd.rate * sigma * sigma)
    for d_snr in (0.2, 0.4, 0.8):
        cha, msg = bench.make_labels(cd, B, base + d_snr, seed=7)
        bits, it = dec.lut_decode_batch(cha, msg)
        h = np.bincount(np.abs(it), minlength=max_iter + 1)
        cum = np.cumsum(h) / B
        print(f"{wl} Eb/N0 {base + d_snr:.2f} dB: mean {np.abs(it).mean():.2f}, failed {int((it < 0).sum())}/{B}")
        print("  finished by iteration: " + " ".join(f"{i}:{cum[i]:.2f}" for i in range(0, max_iter + 1, 2) if cum[i] > 0.005))
        # how many 512-frame groups are completely done by iteration i (what the kernels can skip today)
        g = np.abs(it).reshape(-1, 512).max(axis=1) if B % 512 == 0 else None
        if g is not None:
            print("  groups done by iteration: " + " ".join(f"{i}:{(g <= i).mean():.2f}" for i in range(30, max_iter + 1, 2)))


if __name__ == "__main__":
    main()
