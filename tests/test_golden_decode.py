"""Committed decode vectors (tests/golden/decode_*.npz, written by tests/golden/make_decode_vectors.py from the
oracle): the oracle must keep reproducing them (CPU), and the HIP path must reproduce them through the C-ABI (GPU).
They are regression anchors of this repository's oracle, not outputs of the reference (which ships no decode-level
vectors and cannot be built here: SURVEY 8c)."""
import hashlib
import pathlib

import numpy as np
import pytest

from helpers import oracle_codec, product_decoder

GOLD = pathlib.Path(__file__).resolve().parent / "golden"
NAMES = ["n500_q4", "reg36_n1000_mixed", "c5_minlut", "c5_chklut"]


def _load(name):
    z = np.load(GOLD / f"decode_{name}.npz", allow_pickle=False)
    return z


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_the_committed_vectors(name):
    z = _load(name)
    cd = oracle_codec(name)
    # the design itself is part of the anchor: same trees as when the vectors were written
    assert (np.frombuffer(hashlib.sha256(cd.var_tree_txt.encode()).digest(), np.uint8) == z["trees_sha"]).all()
    for psc, pisc, tag in [(True, True, "shipped"), (False, False, "fixed")]:
        cd.set_exit_conditions(cd.max_iters, psc, pisc)
        bits, iters = cd.lut_decode_batch(z["cha"], z["msg"])
        assert (iters == z[f"iters_{tag}"]).all()
        assert (np.packbits(bits, axis=1) == z[f"bits_{tag}"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_path_reproduces_the_committed_vectors(name):
    z = _load(name)
    cd = oracle_codec(name)             # only for the graph and the designed tables handed to the C-ABI
    dec = product_decoder(cd)
    for psc, pisc, tag in [(True, True, "shipped"), (False, False, "fixed")]:
        dec.set_exit_conditions(cd.max_iters, psc, pisc)
        bits, iters = dec.lut_decode_batch(z["cha"], z["msg"])
        assert (iters == z[f"iters_{tag}"]).all()
        assert (np.packbits(bits, axis=1) == z[f"bits_{tag}"]).all()
    dec.close()
