#!/usr/bin/env python3
"""Do the two throughput levels of the DVB-S2 bench inside one box (253 k / 262 k codewords/s, per process) follow where the buffers
of a process landed?  ONE process: the decoder is created (fresh hipMallocs), timed, destroyed, with a dummy allocation of varying
size in between so that the next one lands elsewhere.  Usage (GPU box): tools/level_probe_realloc.py [reps]"""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import bench
import lut_ldpc_amd as L
import ctypes, io, os, re, tempfile
hip = ctypes.CDLL("libamdhip64.so")

def copy_gbps(ptr, nbytes):
    """device-to-device copy of the first half of [ptr, ptr + nbytes) onto the second half, GB/s of traffic (read + write)"""
    half = (nbytes // 2) & ~0xFFFFF
    best = 0.0
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hip.hipMemcpy(ctypes.c_void_p(ptr + half), ctypes.c_void_p(ptr), ctypes.c_size_t(half), 3)
        hip.hipDeviceSynchronize()
        best = max(best, 2 * half / (time.perf_counter() - t0) / 1e9)
    return best

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
KEEP = "--keep" in sys.argv          # never free a decoder: every one gets memory nobody has used in this process
alist, sigma, max_iter, qc, qm, B, extra, known_rank = bench.WORKLOADS["dvbs2"]
cd = L.Codec(ROOT / "data" / "codes" / f"{alist}.alist", known_rank=known_rank, device=0)
cd.alist, cd.nq_cha, cd.nq_msg = alist, 1 << qc, 1 << qm
cd.design_luts(sigma2=sigma * sigma, max_iters=max_iter, nq_cha=1 << qc, nq_msg=1 << qm, **extra)
cd.set_exit_conditions(max_iter, False, False)
snr = -10 * np.log10(2 * cd.rate * sigma * sigma)
cha, msg = bench.make_labels_device(cd, B, snr, seed=1234)
ob = torch.empty((B, cd.nvar), dtype=torch.uint8, device="cuda"); oi = torch.empty(B, dtype=torch.int32, device="cuda")
dummies = []
def make_codec():
    c = L.Codec(ROOT / "data" / "codes" / f"{alist}.alist", known_rank=known_rank, device=0)
    c.alist, c.nq_cha, c.nq_msg = alist, 1 << qc, 1 << qm
    c.design_luts(sigma2=sigma * sigma, max_iters=max_iter, nq_cha=1 << qc, nq_msg=1 << qm, **extra)
    c.set_exit_conditions(max_iter, False, False)
    return c

for r in range(reps):
    cdr = make_codec()          # a fresh codec owns a fresh decoder handle: fresh hipMallocs
    dec = cdr.decoder()
    tf = tempfile.TemporaryFile(); saved = os.dup(2); os.dup2(tf.fileno(), 2)
    dec.lut_decode_batch_device(cha.data_ptr(), msg.data_ptr(), B, ob.data_ptr(), oi.data_ptr(), sync=True)
    os.dup2(saved, 2); os.close(saved); tf.seek(0); txt = tf.read().decode(); tf.close()
    m = re.search(r"msgs (0x[0-9a-f]+) .*\((\d+) MB", txt)
    mp, mb = (int(m.group(1), 16), int(m.group(2)) << 20) if m else (0, 0)
    for _ in range(2):
        dec.lut_decode_batch_device(cha.data_ptr(), msg.data_ptr(), B, ob.data_ptr(), oi.data_ptr(), sync=True)
    ts = []
    for blk in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(5):
            dec.lut_decode_batch_device(cha.data_ptr(), msg.data_ptr(), B, ob.data_ptr(), oi.data_ptr(), sync=(k == 4))
        ts.append(B * 5 / (time.perf_counter() - t0) / 1e3)
    print(json.dumps({"rep": r, "k_cw_per_s_blocks": [round(t, 1) for t in ts], "keep": KEEP, "msgs": hex(mp), "placement": dec.describe().get("placement")}), flush=True)
    if KEEP: dummies.append(cdr)
    else: cdr.close()
    # shift the next allocation: keep a dummy of odd size alive
    if not KEEP: dummies.append(torch.empty(int((37 + 61 * r) * 2**20 + 4096 * (r + 1)), dtype=torch.uint8, device="cuda"))
