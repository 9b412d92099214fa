"""Worker for test_config4_sweep_gloo.py: lut_ldpc_amd.ber_sim.run on (a reduced-Nframes copy of) data/params/ber.ini.dvbs2_sweep,
single process or as a rank of a gloo group.  The device batch is replaced by the oracle (flat-table mode) on the SAME
Philox-addressed frames, so the test checks the sharding, the stop rules, the padding of the sweep and the result file."""
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from lut_ldpc_amd import ber_sim                                  # noqa: E402
from helpers import oracle_codec                                  # noqa: E402


def main():
    out_path, params, base = sys.argv[1], sys.argv[2], sys.argv[3]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    comm = ber_sim.Comm()
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")
        comm = ber_sim.Comm(dist, torch.device("cpu"))
    cd = oracle_codec("dvbs2_q4")
    cd.set_exit_conditions(50, True, True)                         # parity_check_iter = true: psc AND pisc (src/LDPC_BER_Sim.cpp:500)
    touched = []

    def batch(sim, idx, f0, B):
        touched.append((idx, f0, B))
        K = sim.ninfo
        cha, msg, unc = cd.sample_labels(sim.snr_db[idx], 0.5, 5, idx, f0, B)      # seed 5, stream = SNR index
        bits, it = cd.lut_decode_batch_flat(cha, msg, threads=4)
        be = bits[:, :K].sum(1)
        return np.stack([it, be > 0, be, unc], 1).astype(np.int32)

    pts, path = ber_sim.run(params, base, seed=5, custom_name=f"_w{world}", comm=comm, quiet=True, batch_override=batch)
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, touched)
    else:
        gathered = [touched]
    if comm.rank == 0:
        json.dump({"points": [[s, c.tolist()] for s, c in pts], "path": path, "touched": gathered}, open(out_path, "w"))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
