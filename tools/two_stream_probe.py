#!/usr/bin/env python3
"""Does a second decoder on its own stream fill the tails of the first one's launches?  DVB-S2, fixed work: ONE decoder at B frames per
step against TWO decoders (own handles, own streams, host threads) at B/2 frames each, same total work.  Run on the GPU box.
Usage: tools/two_stream_probe.py [B]"""
import json, sys, threading, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import bench
import lut_ldpc_amd as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
alist, sigma, max_iter, qc, qm, B_default, extra, known_rank = bench.WORKLOADS["dvbs2"]

def make():
    cd = L.Codec(ROOT / "data" / "codes" / f"{alist}.alist", known_rank=known_rank, device=0)
    cd.alist, cd.nq_cha, cd.nq_msg = alist, 1 << qc, 1 << qm
    cd.design_luts(sigma2=sigma * sigma, max_iters=max_iter, nq_cha=1 << qc, nq_msg=1 << qm, **extra)
    cd.set_exit_conditions(max_iter, False, False)
    return cd, cd.decoder()

snr = -10 * np.log10(2 * 0.5 * sigma * sigma)
def run(n_lanes, steps=6):
    lanes = []
    for i in range(n_lanes):
        cd, dec = make()
        Bl = B // n_lanes
        cha, msg = bench.make_labels_device(cd, Bl, snr, seed=77 + i)
        ob = torch.empty((Bl, cd.nvar), dtype=torch.uint8, device="cuda"); oi = torch.empty(Bl, dtype=torch.int32, device="cuda")
        lanes.append((cd, dec, cha, msg, ob, oi, Bl))
    def work(l, k):
        cd, dec, cha, msg, ob, oi, Bl = l
        for j in range(k):
            dec.lut_decode_batch_device(cha.data_ptr(), msg.data_ptr(), Bl, ob.data_ptr(), oi.data_ptr(), sync=(j == k - 1))
    for l in lanes: work(l, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(l, steps)) for l in lanes]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for l in lanes: l[0].close()
    return B * steps / dt

for rep in range(2):
    for n in (1, 2, 4):
        print(json.dumps({"lanes": n, "frames_per_lane": B // n, "k_cw_per_s": round(run(n) / 1e3, 1)}), flush=True)
