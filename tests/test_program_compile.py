"""The product's tree compiler (value sharing, slot allocation, full-table expansion) against the
oracle's faithful queue/recursion evaluation, without a GPU: lutldpc_selftest_program_eval."""
import numpy as np
import pytest

from helpers import oracle_codec, product_decoder

VAR, CHK, DEC = 0, 1, 2


def _check(cd, dec, kind, tree_set, cls, deg, n_trials, rng, k_in, k_cha):
    n_in = deg if kind == CHK else deg + 1
    n_out = 1 if kind == DEC else deg
    for _ in range(n_trials):
        x = rng.integers(0, k_in, n_in).astype(np.int32)
        if kind != CHK:
            x[-1] = rng.integers(0, k_cha)
        want = cd.tree_eval(kind, tree_set, cls, x, n_out)
        got = dec.program_eval(kind, tree_set, cls, x, n_out)
        assert (want == got).all(), (kind, tree_set, cls, x, want, got)


@pytest.mark.parametrize("name", ["n500_q4_i8", "reg36_n1000_mixed", "reg36_n1000_q3_chklut", "reg36_n1000_rootonly",
                                  "reg36_n1000_high", "c5_chklut"])
def test_programs_match_oracle_tree_walk(name):
    cd = oracle_codec(name)
    dec = product_decoder(cd, device=-1)      # host-only handle: no GPU needed for the compile step
    rng = np.random.default_rng(11)
    degs_v = sorted(set(cd.code.dv.tolist()))
    degs_c = sorted(set(cd.code.dc.tolist()))
    n_sets = cd.n_sets()
    set_iters = np.flatnonzero(np.asarray(cd.reuse_vec) == 0)      # first iteration using each tree set
    for s in range(n_sets):
        it = int(set_iters[s])
        k_in = int(cd.nq_msg[it])
        kind = DEC if s == n_sets - 1 else VAR
        for cls, dv in enumerate(degs_v):
            _check(cd, dec, kind, s, cls, dv, 40, rng, k_in, cd.nq_cha)
        if not cd.min_lut:
            for cls, dc in enumerate(degs_c):
                _check(cd, dec, CHK, s, cls, dc, 25, rng, k_in, cd.nq_cha)
    dec.close()


def test_value_sharing_counts():
    """a subtree over L leaves has only L+1 distinct input tuples over all outputs."""
    cd = oracle_codec("n500_q4_i8")
    dec = product_decoder(cd, device=-1)
    stats = {dv: dec.program_stats(VAR, 0, cls) for cls, dv in enumerate([2, 3, 9, 17])}
    assert stats[17]["ops_naive"] == 17 * 16 and stats[17]["ops"] == 96
    assert stats[9]["ops_naive"] == 9 * 8 and stats[9]["ops"] == 40
    assert stats[3]["ops"] == 6 and stats[2]["ops"] == 2
    dec.close()


def test_create_rejects_bad_input():
    import lut_ldpc_amd as L
    cd = oracle_codec("n500_q4_i8")
    c = cd.code
    bad = c.cn_msg_idx.copy(); bad[0] = bad[1]
    with pytest.raises(L.LutLdpcError):
        L.Decoder(c.nvar, c.nchk, c.dv, c.dc, bad, 16, cd.nq_msg, cd.reuse_vec, 8, True, cd.var_tree_txt, "", device=-1)
    with pytest.raises(L.LutLdpcError):
        L.Decoder(c.nvar, c.nchk, c.dv, c.dc, c.cn_msg_idx, 16, cd.nq_msg, cd.reuse_vec, 8, True, "3\n1\n0 2", "", device=-1)
    reuse = np.array(cd.reuse_vec); reuse[0] = 1      # src/LDPC_Code_LUT.cpp:122
    with pytest.raises(L.LutLdpcError):
        L.Decoder(c.nvar, c.nchk, c.dv, c.dc, c.cn_msg_idx, 16, cd.nq_msg, reuse, 8, True, cd.var_tree_txt, "", device=-1)
    d = product_decoder(cd, device=-1)
    with pytest.raises(L.LutLdpcError) as e:       # no device -> loud failure, never a CPU decode
        d.lut_decode_batch(np.zeros((1, c.nvar), np.uint8), np.zeros((1, c.nvar), np.uint8))
    assert e.value.code == -5
    with pytest.raises(L.LutLdpcError):            # iteration 3 is not a decision-tree set
        d.set_exit_conditions(3)
    d.close()


@pytest.mark.parametrize("name", ["c5_minlut", "reg36_n1000_rootonly", "reg36_n1000_high", "reg36_n1000_q5"])
def test_jit_source_compiles_without_a_gpu(name):
    """Tree shapes outside the compile-time path (file trees, root_only, auto_bin_high, 5-bit alphabets) get a
    kernel generated from their node program; hiprtc cross-compiles it for gfx950 here, it runs in the gpu tests."""
    cd = oracle_codec(name)
    dec = product_decoder(cd, device=-1)
    src = dec.jit_source(0, 0, 0, compile=True)
    stats = dec.program_stats(0, 0, 0)
    assert src.count("= tab[") == stats["ops"]                  # one statement per look-up of the shared program
    assert "lutldpc_jit_pass" in src and "pipeline_entry_fence" in src
    last = max(s for s in range(cd.max_iters) if _has_dec(dec, s))
    assert "hardw = lshl_or(" in dec.jit_source(2, last, 0, compile=True)
    dec.close()


@pytest.mark.parametrize("name", ["c5_chklut", "reg36_n1000_q3_chklut"])
def test_jit_check_tree_source_compiles(name):
    """CHKTREE check update (min_lut = false): sign/magnitude look-ups, one statement per shared look-up."""
    cd = oracle_codec(name)
    dec = product_decoder(cd, device=-1)
    src = dec.jit_source(1, 0, 0, compile=True)
    assert src.count("= tab[") == dec.program_stats(1, 0, 0)["ops"]
    dec.close()


def _has_dec(dec, s):
    try:
        dec.program_stats(2, s, 0)
        return True
    except Exception:
        return False


COMPOSED = 16      # kind + 16: the program of the same tree after exact table composition (lut_program.hpp: compose_tree)


@pytest.mark.parametrize("name", ["n500_q4_i8", "reg36_n1000_mixed", "reg36_n1000_q3_chklut", "reg36_n1000_rootonly", "reg36_n1000_high",
                                  "c5_chklut", "dvbs2_q4_i6"])
def test_composed_programs_match_oracle_tree_walk(name, monkeypatch):
    """(LUTLDPC_COMPOSE=1: off by default -- exact, but the 4 KB tables cost more LDS bank conflicts than the saved look-ups are worth.)
    Table composition folds a LUT node into its parent (T'[...] = T_P[..., T_X[...], ...]): fewer look-ups, the same function.
    Every composed program against the oracle's queue / recursion walk of the ORIGINAL tree on random inputs."""
    monkeypatch.setenv("LUTLDPC_COMPOSE", "1")
    cd = oracle_codec(name)
    dec = product_decoder(cd, device=-1)
    rng = np.random.default_rng(12)
    degs_v = sorted(set(cd.code.dv.tolist()))
    degs_c = sorted(set(cd.code.dc.tolist()))
    n_sets = cd.n_sets()
    set_iters = np.flatnonzero(np.asarray(cd.reuse_vec) == 0)
    fewer = 0
    for s in range(n_sets):
        k_in = int(cd.nq_msg[int(set_iters[s])])
        kind = DEC if s == n_sets - 1 else VAR
        for cls, dv in enumerate(degs_v):
            _check_composed(cd, dec, kind, s, cls, dv, 60, rng, k_in, cd.nq_cha)
            fewer += dec.program_stats(kind, s, cls)["ops"] - dec.program_stats(kind + COMPOSED, s, cls)["ops"]
        if not cd.min_lut:
            for cls, dc in enumerate(degs_c):
                _check_composed(cd, dec, CHK, s, cls, dc, 40, rng, k_in, cd.nq_cha)
                fewer += dec.program_stats(CHK, s, cls)["ops"] - dec.program_stats(CHK + COMPOSED, s, cls)["ops"]
    assert fewer > 0 or name == "reg36_n1000_rootonly"          # (root_only: one 3-input table already, nothing to fold)
    dec.close()


def _check_composed(cd, dec, kind, tree_set, cls, deg, n_trials, rng, k_in, k_cha):
    n_in = deg if kind == CHK else deg + 1
    n_out = 1 if kind == DEC else deg
    for _ in range(n_trials):
        x = rng.integers(0, k_in, n_in).astype(np.int32)
        if kind != CHK:
            x[-1] = rng.integers(0, k_cha)
        want = cd.tree_eval(kind, tree_set, cls, x, n_out)
        got = dec.program_eval(kind + COMPOSED, tree_set, cls, x, n_out)
        assert (want == got).all(), (kind, tree_set, cls, x, want, got)


FULL_LABELS = 32   # CHK + 32: the check program over the children's full labels (lut_program.hpp: chk_full_label_program)


@pytest.mark.parametrize("name", ["reg36_n1000_q3_chklut", "c5_chklut"])
def test_full_label_check_programs_match_oracle_tree_walk(name):
    """The generated check kernels look a CHKTREE node up by its children's LABELS (one table of prod(K) entries, one instruction per
    look-up) instead of by (sign parity, magnitudes) as src/LUT_Tree.cpp:420-445 walks it: the same map, checked here against the
    oracle's recursion over the original tree on random inputs, every degree class of every tree set."""
    cd = oracle_codec(name)
    dec = product_decoder(cd, device=-1)
    rng = np.random.default_rng(21)
    degs_c = sorted(set(cd.code.dc.tolist()))
    set_iters = np.flatnonzero(np.asarray(cd.reuse_vec) == 0)
    for s in range(cd.n_sets()):
        k_in = int(cd.nq_msg[int(set_iters[s])])
        for cls, dc in enumerate(degs_c):
            assert dec.program_stats(CHK + FULL_LABELS, s, cls)["ops"] == dec.program_stats(CHK, s, cls)["ops"]
            for _ in range(60):
                x = rng.integers(0, k_in, dc).astype(np.int32)
                want = cd.tree_eval(CHK, s, cls, x, dc)
                got = dec.program_eval(CHK + FULL_LABELS, s, cls, x, dc)
                assert (want == got).all(), (s, cls, x, want, got)
    dec.close()


def test_composition_look_up_counts(monkeypatch):
    """Balanced trees: degree 3 -> three 3-input look-ups instead of six; degree 8 -> 20 instead of 34."""
    monkeypatch.setenv("LUTLDPC_COMPOSE", "1")
    cd = oracle_codec("dvbs2_q4_i6")
    dec = product_decoder(cd, device=-1)
    degs = sorted(set(cd.code.dv.tolist()))
    ops = {dv: (dec.program_stats(VAR, 0, c)["ops"], dec.program_stats(VAR + COMPOSED, 0, c)["ops"]) for c, dv in enumerate(degs)}
    assert ops[3] == (6, 3) and ops[8] == (34, 20) and ops[2] == (2, 2), ops
    dec.close()


@pytest.mark.parametrize("name,G", [("n500_q4_i8", 3), ("reg36_n1000_mixed", 1), ("c5_chklut", 2), ("c5_minlut", 64), ("reg36_n10000_q4", 8), ("reg36_n1000_q5", 2)])
def test_resident_source_compiles_without_a_gpu(name, G):
    """The LDS-resident decode kernel (jit_resident.hpp) is generated per code and batch shape; hiprtc cross-compiles it here."""
    cd = oracle_codec(name)
    dec = product_decoder(cd, device=-1)
    src, (S, NT, lds) = dec.resident_source(G, compile=True)
    assert "lutldpc_jit_pass" in src and "__shared__" in src and S >= 1 and NT in (256, 512, 1024) and lds <= 160 * 1024 - 2048
    assert S * cd.code.nedges * 4 <= lds
    dec.close()


def test_resident_refuses_codes_that_do_not_fit_the_lds():
    import lut_ldpc_amd as L
    cd = oracle_codec("dvbs2_q4_i6")
    dec = product_decoder(cd, device=-1)
    assert dec.describe()["resident"] == 0
    with pytest.raises(L.LutLdpcError):
        dec.resident_source(4)
    dec.close()


def test_describe_names_the_device_code_by_a_hash_of_its_sources():
    """bench.py replays a counter set from profiles/ only when it was taken on the running device code: describe() carries a hash of
    everything under csrc/hip/ (Makefile: kernel_src_hash.inc), stable across rebuilds of unchanged sources."""
    import hashlib
    from pathlib import Path
    dec = product_decoder(oracle_codec("n500_q4_i8"), device=-1)
    h = dec.describe()["kernel_sources"]
    hip = Path(__file__).resolve().parent.parent / "lut_ldpc_amd" / "csrc" / "hip"
    files = sorted([*hip.glob("*.hpp"), *hip.glob("*.hip")], key=lambda p: "hip/" + p.name)
    assert h == hashlib.sha256(b"".join(p.read_bytes() for p in files)).hexdigest()[:16]
    dec.close()
