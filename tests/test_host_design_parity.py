"""The product's C++ host mirror (alist, rank, LUT design, codec files) against the oracle.
Bit-exact: the serialised trees must be the same text, the boundaries the same doubles."""
import numpy as np
import pytest

import lut_ldpc_amd as L
from helpers import CODES, CONFIGS, oracle_codec


def product_codec(name, device=-1):
    alist, kw = CONFIGS[name]
    kw = dict(kw)
    rank = kw.pop("rank", None)
    cd = L.Codec(CODES / f"{alist}.alist", known_rank=rank or (32400 if "64800" in alist else 0), device=device)
    snr = kw.pop("design_snr_db", None)
    sigma2 = kw.pop("sigma2")
    if sigma2 is None:
        sigma2 = 10 ** (-snr / 10) / (2 * cd.rate)
    if "allow_deg1" in kw:
        kw["allow_degree_one"] = kw.pop("allow_deg1")
    cd.design_luts(sigma2=sigma2, **kw)
    return cd


@pytest.mark.parametrize("name", ["n500_q4_i8", "reg36_n1000_mixed", "reg36_n1000_q3_chklut", "reg36_n1000_rootonly",
                                  "reg36_n1000_high", "c5_minlut", "c5_chklut", "dvbs2_q4_i6"])
def test_design_matches_oracle(name):
    want = oracle_codec(name)
    got = product_codec(name)
    assert got.var_trees_txt == want.var_tree_txt
    if not want.min_lut:
        assert got.chk_trees_txt == want.chk_tree_txt
    assert (got.qb_cha == want.qb_cha).all() and (got.qb_msg == want.qb_msg).all()
    assert (got.cha2msg_map == want.cha2msg_map).all()
    dv, dc, cn = got.graph()
    assert (dv == want.code.dv).all() and (dc == want.code.dc).all() and (cn == want.code.cn_msg_idx).all()
    got.close()


def test_rank_and_rate():
    cd = L.Codec(CODES / "rate0.84_reg_v6c32_N2048.alist", device=-1)
    assert cd.rank == 325 and f"{cd.rate:g}" == "0.841309"        # README.md:239
    cd.close()
    cd = L.Codec(CODES / "rate0.50_dv02-17_dc08-09_lut_q4_N500.alist", device=-1)
    assert cd.rank == 250 and f"{cd.rate:g}" == "0.5"             # README.md:114
    cd.close()


def test_degree_one_needs_the_extension():
    cd = L.Codec(CODES / "rate0.50_irreg_dvbs2_N64800.alist", known_rank=32400, device=-1)
    with pytest.raises(L.LutLdpcError):       # the reference asserts here (src/LUT_Tree.cpp:202)
        cd.design_luts(max_iters=3)
    cd.close()


def test_generator_makes_codewords():
    cd = L.Codec(CODES / "rate0.84_reg_v6c32_N2048.alist", with_generator=True, device=-1)
    assert cd.ninfo == 2048 - 325
    rng = np.random.default_rng(3)
    dv, dc, cn = cd.graph()
    vn_of_edge = np.repeat(np.arange(cd.nvar), dv)      # H rebuilt from the (column-permuted) graph
    for _ in range(5):
        info = rng.integers(0, 2, cd.ninfo).astype(np.uint8)
        cw = cd.encode(info)
        assert (cw[:cd.ninfo] == info).all()              # systematic, src/LDPC_Code_LUT.cpp:225
        p = 0
        for c in range(cd.nchk):
            assert cw[vn_of_edge[cn[p:p + dc[c]]]].sum() % 2 == 0
            p += dc[c]
    cd.close()


def test_codec_file_roundtrip(tmp_path):
    cd = product_codec("n500_q4_i8")
    path = tmp_path / "lut_codec.it"
    cd.save(path)
    raw = path.read_bytes()
    assert raw[:5] == b"IT++\x03" and b"var_tree_string\x00string\x00" in raw      # scripts/itload.m:48-63
    cd2 = L.Codec(codec_path=path, device=-1)
    assert cd2.var_trees_txt == cd.var_trees_txt and (cd2.qb_cha == cd.qb_cha).all()
    assert (cd2.nvar, cd2.nchk, cd2.rank) == (cd.nvar, cd.nchk, cd.rank)
    cd.close()
    cd2.close()


@pytest.mark.slow
def test_de_threshold_readme_product():
    thr, it = L.de_threshold([2, 3, 9, 17], [0.138045, 0.401038, 0.026586, 0.434331], [8, 9], [0.323376, 0.676624])
    assert it == 20 and f"{thr:g}" == "0.929193"      # README.md:173-176


def test_design_cache_roundtrip(tmp_path, monkeypatch):
    """LUTLDPC_DESIGN_CACHE: the second design with equal inputs is read back (trees in the reference's text serialisation,
    boundaries as hex floats) and is identical to a fresh design; other inputs miss; a damaged file is ignored."""
    name = "reg36_n1000_mixed"
    fresh = product_codec(name)
    assert not fresh.design_from_cache
    monkeypatch.setenv("LUTLDPC_DESIGN_CACHE", str(tmp_path))
    a = product_codec(name)
    assert not a.design_from_cache and len(list(tmp_path.glob("*.lutdesign"))) == 1
    b = product_codec(name)
    assert b.design_from_cache
    assert b.var_trees_txt == fresh.var_trees_txt and (b.qb_cha == fresh.qb_cha).all() and (b.qb_msg == fresh.qb_msg).all()
    assert (b.cha2msg_map == fresh.cha2msg_map).all()
    # a cache hit leaves the same object behind as a fresh design: the codec files are byte-identical (min_lut = true here:
    # the check templates a fresh design keeps are restored too)
    fresh.save(tmp_path / "fresh.it"); b.save(tmp_path / "hit.it")
    assert (tmp_path / "fresh.it").read_bytes() == (tmp_path / "hit.it").read_bytes()
    assert b.chk_trees_txt == fresh.chk_trees_txt
    c = product_codec("reg36_n1000_q3_chklut")            # other alphabets, CHKTREE designs
    assert not c.design_from_cache and len(list(tmp_path.glob("*.lutdesign"))) == 2
    d = product_codec("reg36_n1000_q3_chklut")
    assert d.design_from_cache and d.chk_trees_txt == c.chk_trees_txt and d.var_trees_txt == c.var_trees_txt
    for f in tmp_path.glob("*.lutdesign"):                 # truncated files are not designs
        f.write_bytes(f.read_bytes()[:200])
    e = product_codec(name)
    assert not e.design_from_cache and e.var_trees_txt == fresh.var_trees_txt
    for x in (fresh, a, b, c, d, e):
        x.close()


@pytest.mark.parametrize("reuse", [[0, 1, 0, 1], [1, 0, 0, 0], [0, 1, 0, 0]])
def test_reuse_in_the_first_or_last_iteration_is_refused(reuse):
    """src/LDPC_DE.cpp:199 and src/LDPC_Code_LUT.cpp:122 stop on a reuse vector that reuses in the first or last iteration: the
    product returns an error, the oracle too (found by tests/fuzz_parity.py: the oracle used to run into the two-label decision
    stage with a reused stage's alphabet).  Third case: a reused stage across a change of the message alphabet (16 -> 8 after
    iteration 0) -- the reference would add probability vectors of different lengths."""
    import lut_ldpc_amd as L
    from helpers import CODES
    from oracle import oracle as orc
    alist = CODES / "rate0.50_dv03_dc06_N1000.alist"
    c = L.Codec(alist, known_rank=500, device=-1)
    with pytest.raises(L.LutLdpcError):
        c.design_luts(sigma2=0.7, max_iters=4, nq_cha=16, nq_msg=[16, 8, 8, 8], reuse_vec=reuse)
    c.design_luts(sigma2=0.7, max_iters=4, nq_cha=16, nq_msg=[16, 8, 8, 8], reuse_vec=[0, 0, 1, 0])     # a valid one still works
    cd = orc.Codec(orc.Code(alist), skip_rank=True)
    cd.set_rank(500)
    with pytest.raises(RuntimeError):
        cd.design_luts(sigma2=0.7, max_iters=4, nq_msg=np.array([16, 8, 8, 8], np.int32), nq_cha=16, reuse_vec=reuse)
