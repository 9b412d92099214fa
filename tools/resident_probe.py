#!/usr/bin/env python3
"""Resident decoder vs streaming kernels on the small BASELINE workloads: throughput on device-resident labels (bench.py's
workload table), both exit modes, oracle check of a sample.  Usage: tools/resident_probe.py [workload ...]"""
import json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import bench
import lut_ldpc_amd as L

def run(wl, B, resident, env=None):
    os.environ["LUTLDPC_RESIDENT"] = "1" if resident else "0"
    for k, v in (env or {}).items():
        os.environ[k] = v
    alist, sigma, max_iter, qc, qm, B_default, extra, known_rank = bench.WORKLOADS[wl]
    B = B or B_default
    cd = L.Codec(ROOT / "data" / "codes" / f"{alist}.alist", known_rank=known_rank, device=0)
    cd.alist, cd.nq_cha, cd.nq_msg = alist, 1 << qc, 1 << qm
    cd.design_luts(sigma2=sigma * sigma, max_iters=max_iter, nq_cha=1 << qc, nq_msg=1 << qm, **extra)
    cd.min_lut = bool(extra.get("min_lut", True))
    qmap = None
    if wl.startswith("c5"):
        cd.set_initial_message_mode(1); qmap = np.asarray(cd.cha2msg_map, np.uint8)
    out = {"workload": wl, "B": B, "resident": resident, "env": env or {}}
    for psc in ((False, True) if os.environ.get("PROBE_MODES", "both") == "both" else (False,)):
        snr = -10 * np.log10(2 * cd.rate * sigma * sigma) + (0.4 if psc else 0.0)
        cd.set_exit_conditions(max_iter, psc, psc)
        dec = cd.decoder()
        cha, msg = bench.make_labels_device(cd, B, snr, seed=1234, qcha_map=qmap)      # (synchronises: the decoder runs on its own stream)
        ob = torch.empty((B, cd.nvar), dtype=torch.uint8, device="cuda"); oi = torch.empty(B, dtype=torch.int32, device="cuda")
        t0 = time.perf_counter()
        for _ in range(3):
            dec.lut_decode_batch_device(cha.data_ptr(), msg.data_ptr(), B, ob.data_ptr(), oi.data_ptr(), sync=True)
        t_first = time.perf_counter() - t0
        K = 5
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(K):
            dec.lut_decode_batch_device(cha.data_ptr(), msg.data_ptr(), B, ob.data_ptr(), oi.data_ptr(), sync=(k == K - 1))
        dt = (time.perf_counter() - t0) / K
        dec.set_profiling(True); dec.reset_profile()
        dec.lut_decode_batch_device(cha.data_ptr(), msg.data_ptr(), B, ob.data_ptr(), oi.data_ptr(), sync=True)
        prof = {k: round(v["ms"], 3) for k, v in dec.profile().items() if v["launches"]}
        dec.set_profiling(False)
        it = oi.cpu().numpy()
        n = min(B, 48)
        oc = bench.oracle_codec_for(cd, max_iter, psc, psc)
        wb, wi = oc.lut_decode_batch_flat(cha[:n].cpu().numpy(), msg[:n].cpu().numpy(), threads=bench.usable_cores())
        same = bool((wi == it[:n]).all() and (wb == ob[:n].cpu().numpy()).all())
        out["shipped" if psc else "fixed"] = {"cw_per_s": B / dt, "ms": dt * 1e3, "warm3_s": round(t_first, 2), "mean_iters": float(np.abs(it).mean()), "kernel_ms": prof, "oracle_match_48": same}
        out["describe_resident"] = dec.describe().get("resident")
        if resident:
            out["S_NT_lds"] = list(dec.resident_source((B + dec.describe()["tile_frames"] - 1) // dec.describe()["tile_frames"])[1])
    print(json.dumps(out), flush=True)
    cd.close()

if __name__ == "__main__":
    # usage: resident_probe.py [workload ...] [-- KEY=VALUE,KEY=VALUE ...]   (each env set after -- is one more resident run)
    argv = sys.argv[1:]
    envs = []
    if "--" in argv:
        i = argv.index("--")
        envs = [dict(kv.split("=") for kv in e.split(",") if kv) for e in argv[i + 1:]]
        argv = argv[:i]
    wls = argv or ["c2", "c1", "c5", "c5chk"]
    for wl in wls:
        B = 4096 if wl == "c2" else 0
        if not envs:
            run(wl, B, False)
        for env in (envs or [{}]):
            for k in list(os.environ):
                if k.startswith("LUTLDPC_") and k not in ("LUTLDPC_DESIGN_CACHE",):
                    del os.environ[k]
            run(wl, B, True, env)
