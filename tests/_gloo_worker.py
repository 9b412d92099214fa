"""Worker for test_sharded_gloo.py: world_size-2 gloo run of the sharded frame loop on CPU.  The per-frame
results come from the oracle (every rank computes the same table; a rank only READS its own shard)."""
import json
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from lut_ldpc_amd.ber_sim import Comm, sim_snr_point_sharded   # noqa: E402
from helpers import oracle_codec                                 # noqa: E402


def main():
    out_path, snr, nframes, nfers = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dist.init_process_group("gloo")
    comm = Comm(dist, torch.device("cpu"))
    cd = oracle_codec("n500_q4_i8")
    cd.set_exit_conditions(8, True, True)
    K = cd.code.nvar - 250
    touched = []

    def batch_fn(f0, b):
        touched.append((f0, b))
        # each rank simulates only its own frames (oracle = stand-in for the GPU batch)
        cha, msg, unc = cd.sample_labels(snr, 0.5, 9, 1, f0, b)
        bits, it = cd.lut_decode_batch(cha, msg)
        be = bits[:, :K].sum(1)
        return np.stack([it, be > 0, be, unc], 1).astype(np.int32)

    c = sim_snr_point_sharded(batch_fn, nframes, nfers, K, comm, batch_max=64, batch_first=8)
    gathered = [None] * comm.world
    dist.all_gather_object(gathered, touched)
    if comm.rank == 0:
        json.dump({"counters": c.tolist(), "touched": gathered}, open(out_path, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
