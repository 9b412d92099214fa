#!/usr/bin/env python3
"""Writes tests/golden/decode_*.npz: inputs and outputs of the ORACLE's lut_decode for a few frames of small
configurations (seeded labels).  The reference ships no decode-level vectors and cannot be built here,
so these are regression anchors of this repository's own oracle -- they pin the oracle against accidental change
and give the GPU tests a second, oracle-build-independent target -- not outputs of the reference.
Re-create with:  python tests/golden/make_decode_vectors.py"""
import pathlib
import sys

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE.parent))
from helpers import awgn_labels, oracle_codec  # noqa: E402

CASES = [("n500_q4", 24, 1.8, 0), ("reg36_n1000_mixed", 16, 2.2, 0), ("c5_minlut", 24, 4.0, 1), ("c5_chklut", 12, 4.2, 1)]

for name, B, snr, mode in CASES:
    cd = oracle_codec(name)
    cha, msg, _ = awgn_labels(cd, B, snr, seed=20261004, mode=mode)
    out = {"cha": cha, "msg": msg, "trees_sha": np.frombuffer(__import__("hashlib").sha256(cd.var_tree_txt.encode()).digest(), np.uint8)}
    for psc, pisc, tag in [(True, True, "shipped"), (False, False, "fixed")]:
        cd.set_exit_conditions(cd.max_iters, psc, pisc)
        bits, iters = cd.lut_decode_batch(cha, msg)
        out[f"bits_{tag}"] = np.packbits(bits, axis=1)
        out[f"iters_{tag}"] = iters
    np.savez_compressed(HERE / f"decode_{name}.npz", **out)
    print(name, {k: v.shape for k, v in out.items()})
