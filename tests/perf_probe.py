"""Quick kernel-time probe (test infrastructure; uses oracle-designed tables).  Not the bench."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import numpy as np
from helpers import awgn_labels, oracle_codec, product_decoder


def probe(name, B, snr, psc, reps=3):
    cd = oracle_codec(name)
    dec = product_decoder(cd)
    cha, msg, _ = awgn_labels(cd, min(B, 512), snr, seed=1)
    reps_b = (B + len(cha) - 1) // len(cha)
    cha = np.tile(cha, (reps_b, 1))[:B]; msg = np.tile(msg, (reps_b, 1))[:B]
    dec.set_exit_conditions(cd.max_iters, psc, psc)
    dec.lut_decode_batch(cha, msg)
    dec.set_profiling(True); dec.reset_profile()
    t = time.time()
    for _ in range(reps):
        bits, it = dec.lut_decode_batch(cha, msg)
    wall = (time.time() - t) / reps
    prof = dec.profile()
    N, E, I = cd.code.nvar, cd.code.nedges, cd.max_iters
    out = {"name": name, "B": B, "psc": psc, "wall_s_incl_pcie": wall, "iters_mean": float(np.abs(it).mean())}
    tot = 0
    for k, v in prof.items():
        if v["launches"]:
            out[k] = {"ms_per_launch": v["ms"] / v["launches"], "launches": v["launches"] // reps, "ms_per_decode": v["ms"] / reps}
            tot += v["ms"] / reps
    out["kernel_ms_per_decode"] = tot
    out["cw_per_s_kernels"] = B / (tot * 1e-3)
    cn_bytes = 2 * E * B; vn_bytes = (2 * E + N) * B
    out["cn_GBps"] = cn_bytes / (out["cn_pass"]["ms_per_launch"] * 1e-3) / 1e9
    out["vn_GBps"] = vn_bytes / (out["vn_pass"]["ms_per_launch"] * 1e-3) / 1e9
    alg = (4 * I * E + (I + 2) * N + N / 8) * B
    out["alg_GBps_overall"] = alg / (tot * 1e-3) / 1e9
    print(json.dumps(out))
    dec.close()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "c2"):
        probe("reg36_n10000_q4", 4096, 1.8, False)
        probe("reg36_n10000_q4", 4096, 1.8, True)
    if which in ("all", "dvbs2"):
        probe("dvbs2_q4", 1024, 1.2, False)
