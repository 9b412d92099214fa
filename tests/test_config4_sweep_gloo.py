"""BASELINE.json config 4 as a runnable artefact: data/params/ber.ini.dvbs2_sweep (DVB-S2 N=64800, Nframes = 1e6 per point,
sweep 0:.5:4) through lut_ldpc_amd.ber_sim.run, single process and as two gloo ranks on CPU, at reduced Nframes.  Counters
must be identical (frames are Philox-addressed, the stop rule runs in global frame order), the sweep must stop and pad like
src/LDPC_BER_Sim.cpp:142-149,307, and the result file must carry the fields scripts/aggregate_results.m:73-84 sums."""
import json
import re
import shutil
import subprocess
import sys

import numpy as np

from helpers import CODES, ROOT
from itfile_reader import itload

FIELDS = ["sim_SNRdB", "sim_Nframes", "sim_Ndatabits", "sim_frame_errors", "sim_data_bit_errors", "sim_uncoded_bit_errors",
          "ldpc_nvar", "ldpc_nchk", "ldpc_code_rate", "runtime"]          # src/LDPC_BER_Sim.cpp:347-358


def test_config4_params_file_is_the_baseline_config():
    txt = (ROOT / "data" / "params" / "ber.ini.dvbs2_sweep").read_text()
    assert re.search(r"parity_filename\s*=\s*rate0\.50_irreg_dvbs2_N64800", txt)
    assert re.search(r"Nframes\s*=\s*1e6", txt) and re.search(r"SNRdB\s*=\s*0:\.5:4", txt)
    assert re.search(r"max_iter\s*=\s*50", txt) and re.search(r"qbits_message_uniform\s*=\s*4", txt) and re.search(r"qbits_channel\s*=\s*4", txt)


def test_two_gloo_ranks_equal_one_process_on_the_config4_sweep(tmp_path):
    base = tmp_path / "base"
    (base / "codes").mkdir(parents=True)
    shutil.copy(CODES / "rate0.50_irreg_dvbs2_N64800.alist", base / "codes")
    params = tmp_path / "ber.ini.dvbs2_sweep"
    # the committed file with Nframes reduced (12 frames per point) and small device batches, everything else untouched
    txt = (ROOT / "data" / "params" / "ber.ini.dvbs2_sweep").read_text()
    txt = re.sub(r"Nframes\s*=\s*1e6", "Nframes  = 12", txt)
    txt = re.sub(r"batch_frames\s*=\s*32768", "batch_frames = 4", txt)
    params.write_text(txt)
    worker = str(ROOT / "tests" / "_gloo_worker_c4.py")
    one, two = tmp_path / "one.json", tmp_path / "two.json"
    r = subprocess.run([sys.executable, worker, str(one), str(params), str(base)], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29571", worker, str(two), str(params), str(base)], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = json.load(open(one)), json.load(open(two))
    assert a["points"] == b["points"]
    snr = [p[0] for p in a["points"]]
    cnt = np.array([p[1] for p in a["points"]])
    assert snr == [0, .5, 1, 1.5, 2, 2.5, 3, 3.5, 4]
    assert (cnt[0] == [12, 12 * 32400, 12, cnt[0][3], cnt[0][4]]).all() and cnt[0][3] > 0          # 0 dB: every frame fails
    stop = int(np.flatnonzero(cnt[:, 2] == 0)[0])                     # first error-free point: FER 0 < fer_min ends the sweep (:307)
    assert 1 <= stop <= 5 and (cnt[stop + 1:] == 0).all() and cnt[stop][0] == 12
    # both ranks worked, on disjoint frames of every simulated point
    t0 = {(i, f) for i, f0, n in b["touched"][0] for f in range(f0, f0 + n)}
    t1 = {(i, f) for i, f0, n in b["touched"][1] for f in range(f0, f0 + n)}
    assert t0 and t1 and not (t0 & t1)
    # result files: the field set the reference's aggregation script reads, equal counters in both
    fa, fb = itload(a["path"]), itload(b["path"])
    for k in FIELDS:
        assert k in fa and k in fb, k
    for k in FIELDS[:9]:
        assert (np.asarray(fa[k]) == np.asarray(fb[k])).all(), k
    assert (fa["sim_Nframes"] == cnt[:, 0]).all() and (fa["sim_frame_errors"] == cnt[:, 2]).all()
    assert fa["ldpc_nvar"][0] == 64800 and fa["ldpc_nchk"][0] == 32400 and abs(fa["ldpc_code_rate"][0] - 0.5) < 1e-12
    assert "frames12" in a["path"] and "_N64800_R0.5_maxIter50_zcw1_" in a["path"]
