#!/usr/bin/env python3
"""Does the decode time depend on where this process' buffers landed?  Re-creates the decoder (fresh
hipMallocs) several times inside ONE process and times the same decode; prints the buffer addresses."""
import sys, time, pathlib, subprocess
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import lut_ldpc_amd as L

B = 4096
cd = L.Codec(ROOT / "data" / "codes" / "rate0.50_irreg_dvbs2_N64800.alist", known_rank=32400, device=0)
cd.design_luts(sigma2=0.88 ** 2, max_iters=50, nq_cha=16, nq_msg=16, allow_degree_one=True)
cd.set_exit_conditions(50, False, False)
N = cd.nvar
rng = np.random.default_rng(1)
cha = torch.from_numpy(rng.integers(0, 16, (B, N), dtype=np.uint8)).cuda()
msg = cha.clone()
out = torch.empty((B, N), dtype=torch.uint8, device="cuda")
it = torch.empty(B, dtype=torch.int32, device="cuda")
# AWGN-like labels (the levels show with the bench's data, uniformly random labels give one steady level)
from bench import make_labels
cd.alist = "rate0.50_irreg_dvbs2_N64800"
cha_h, msg_h = make_labels(cd, B, 1.11, seed=1234)
cha = torch.from_numpy(cha_h).cuda(); msg = torch.from_numpy(msg_h).cuda()
alive = []
for rep in range(5):
    vt = cd.var_trees_txt
    dv, dc, cn = cd.graph()
    dec = L.Decoder(N, cd.nchk, dv, dc, cn, 16, np.full(50, 16, np.int32), np.zeros(50, np.uint8), 50, True, vt, "", device=0)
    dec.set_exit_conditions(50, False, False)
    alive.append(dec)                                   # earlier decoders stay alive: every new one gets other memory
    for d in alive:
        for _ in range(3):
            d.lut_decode_batch_device(cha.data_ptr(), msg.data_ptr(), B, out.data_ptr(), it.data_ptr(), sync=True)
        t0 = time.perf_counter()
        for _ in range(5):
            d.lut_decode_batch_device(cha.data_ptr(), msg.data_ptr(), B, out.data_ptr(), it.data_ptr(), sync=False)
        d.lut_decode_batch_device(cha.data_ptr(), msg.data_ptr(), B, out.data_ptr(), it.data_ptr(), sync=True)
        dt = (time.perf_counter() - t0) / 6
        print(f"round {rep}: decoder #{alive.index(d)}: {dt*1e3:.2f} ms/step  {B/dt:.0f} cw/s", flush=True)
