"""The N>1 path on CPU: two gloo processes shard the frames of an SNR point and exchange only counters;
the result must equal the single-process, frame-by-frame oracle loop (src/LDPC_BER_Sim.cpp:246-311)."""
import json
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT, oracle_codec


@pytest.mark.parametrize("snr,nframes,nfers", [(1.0, 200, 5), (3.5, 150, 20)])
def test_two_rank_gloo_equals_sequential(tmp_path, snr, nframes, nfers):
    out = tmp_path / "out.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + int(snr * 10) % 50), str(ROOT / "tests" / "_gloo_worker.py"), str(out), str(snr), str(nframes), str(nfers)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.load(open(out))
    cd = oracle_codec("n500_q4_i8")
    cd.set_exit_conditions(8, True, True)
    want, per, _ = cd.sim_snr_point(snr, 0.5, 250, 9, 1, nframes, nfers)
    assert got["counters"] == want.tolist()
    # the two ranks simulated disjoint frame ranges
    r0 = {f for f0, b in got["touched"][0] for f in range(f0, f0 + b)}
    r1 = {f for f0, b in got["touched"][1] for f in range(f0, f0 + b)}
    assert r0 and r1 and not (r0 & r1)
