"""The oracle against everything the reference itself publishes for this path."""
import numpy as np
import pytest

from oracle import oracle as orc
from helpers import CODES, ROOT, oracle_codec


def test_tree_known_answer():
    """trees/README.md:24-85 -- 7 LUT nodes x 128 entries designed from two Gaussian pmfs."""
    got = orc.readme_tree().split()
    want = (ROOT / "tests" / "golden" / "readme_tree_t8.txt").read_text().split()
    assert len(want) == 958
    assert got == want


@pytest.mark.parametrize("alist,rank,rate_str", [
    ("rate0.50_dv02-17_dc08-09_lut_q4_N500", 250, "0.5"),          # README.md:114  RES_N500_R0.5_...
    ("rate0.84_reg_v6c32_N2048", 325, "0.841309"),                  # README.md:239  RES_N2048_R0.841309_...
])
def test_rank_from_result_folder_names(alist, rank, rate_str):
    code = orc.Code(CODES / f"{alist}.alist")
    assert code.rank() == rank
    assert f"{1.0 - rank / code.nvar:g}" == rate_str   # operator<< of a double prints 6 significant digits


def test_graph_indexing_is_a_permutation():
    code = orc.Code(CODES / "rate0.50_dv02-17_dc08-09_lut_q4_N500.alist")
    assert (code.nvar, code.nchk, code.nedges) == (500, 250, 2149)
    assert sorted(code.cn_msg_idx.tolist()) == list(range(code.nedges))
    # edge ids inside one check ascend with the VN index (src/LDPC_Code_LUT.cpp:513-527)
    vn_of_edge = np.repeat(np.arange(code.nvar), code.dv)
    p = 0
    for c in range(code.nchk):
        vns = vn_of_edge[code.cn_msg_idx[p:p + code.dc[c]]]
        assert (np.diff(vns) > 0).all()
        assert (vns == code.row_idx[p:p + code.dc[c]]).all()
        p += code.dc[c]


@pytest.mark.slow
def test_de_threshold_readme():
    """README.md:138-178: sigma* = 0.929193 after 20 bisections for rate0.50_dv02-17_dc08-09_lut_q4.ens."""
    thr, it = orc.de_threshold([2, 3, 9, 17], [0.138045, 0.401038, 0.026586, 0.434331], [8, 9], [0.323376, 0.676624])
    assert it == 20
    assert f"{thr:g}" == "0.929193"


def test_design_symmetry_and_shapes():
    cd = oracle_codec("n500_q4_i8")
    qb = cd.qb_cha
    assert len(qb) == 15 and qb[7] == 0 and np.allclose(qb, -qb[::-1])
    assert cd.n_sets() == 8
    # last set is the decision tree set: dv+1 leaves (src/LDPC_DE.cpp:1257-1259)
    for cls, dv in enumerate([2, 3, 9, 17]):
        assert cd.tree_info(0, 0, cls) == (0, dv)
        assert cd.tree_info(0, 7, cls) == (2, dv + 1)


def test_degree_one_aborts_like_the_reference():
    """the DVB-S2 alist has a degree-1 VN; auto trees assert num_leaves >= 2 (src/LUT_Tree.cpp:202)."""
    code = orc.Code(CODES / "rate0.50_irreg_dvbs2_N64800.alist")
    assert np.bincount(code.dv)[1] == 1
    cd = orc.Codec(code)
    with pytest.raises(RuntimeError):
        cd.design_luts(max_iters=3, nq_msg=np.full(3, 16, np.int32))


def test_decoder_corrects_errors():
    cd = oracle_codec("n500_q4")
    from helpers import awgn_labels
    cha, msg, _ = awgn_labels(cd, 24, 3.0, seed=7)
    cd.set_exit_conditions(50, True, True)
    bits, iters = cd.lut_decode_batch(cha, msg)
    assert (cha < 8).mean() > 0.05          # plenty of channel errors
    assert bits.sum() == 0 and (iters > 0).all() and (iters < 50).all()
