"""Shared helpers for the parity tests.  The ORACLE (oracle/) is the checker; the product is
lut_ldpc_amd.  Both sides always receive the *same* label arrays (never rely on
cross-platform float RNG equality)."""
from __future__ import annotations

import functools
from pathlib import Path

import numpy as np

from oracle import oracle as orc

ROOT = Path(__file__).resolve().parent.parent
CODES = ROOT / "data" / "codes"
TREES = ROOT / "data" / "trees"

# name -> (alist, design kwargs)
CONFIGS = {
    # C1 of BASELINE.json: params/ber.ini.irregular.example
    "n500_q4": ("rate0.50_dv02-17_dc08-09_lut_q4_N500", dict(sigma2=0.88 ** 2, max_iters=50, nq_cha=16, nq_msg=16)),
    "n500_q4_i8": ("rate0.50_dv02-17_dc08-09_lut_q4_N500", dict(sigma2=0.88 ** 2, max_iters=8, nq_cha=16, nq_msg=16)),
    "reg36_n1000_q4": ("rate0.50_dv03_dc06_N1000", dict(sigma2=0.88 ** 2, max_iters=20, nq_cha=16, nq_msg=16)),
    # non-uniform message resolution + LUT reuse (SURVEY step 5)
    "reg36_n1000_mixed": ("rate0.50_dv03_dc06_N1000", dict(sigma2=0.82 ** 2, max_iters=12, nq_cha=16,
                                                           nq_msg=[16, 16, 16, 8, 8, 8, 8, 8, 8, 8, 8, 8],
                                                           reuse_vec=[0, 1, 0, 0, 1, 1, 0, 1, 0, 0, 1, 0])),
    # 5-bit labels: more than 16 labels -> byte rows (PACK = 1), 1024-entry tables -> generic kernels
    "reg36_n1000_q5": ("rate0.50_dv03_dc06_N1000", dict(sigma2=0.84 ** 2, max_iters=8, nq_cha=32, nq_msg=32)),
    "reg36_n1000_q3_chklut": ("rate0.50_dv03_dc06_N1000", dict(sigma2=0.80 ** 2, max_iters=10, nq_cha=16, nq_msg=8, min_lut=False)),
    "reg36_n1000_rootonly": ("rate0.50_dv03_dc06_N1000", dict(sigma2=0.85 ** 2, max_iters=6, nq_cha=8, nq_msg=8, tree_method="root_only")),
    "reg36_n1000_high": ("rate0.50_dv03_dc06_N1000", dict(sigma2=0.85 ** 2, max_iters=6, nq_cha=16, nq_msg=16, tree_method="auto_bin_high")),
    # C5 of BASELINE.json: params/ber.ini.regular.example (design_SNRdB 3.9, R = 1 - 325/2048)
    "c5_minlut": ("rate0.84_reg_v6c32_N2048", dict(sigma2=None, design_snr_db=3.9, max_iters=8, nq_cha=16, nq_msg=8,
                                                   tree_method="filename=" + str(TREES / "6_32_wide.ini"), rank=325)),
    "c5_chklut": ("rate0.84_reg_v6c32_N2048", dict(sigma2=None, design_snr_db=3.9, max_iters=8, nq_cha=16, nq_msg=8, min_lut=False,
                                                   tree_method="filename=" + str(TREES / "6_32_wide.ini"), rank=325)),
    # C2 / C3 of BASELINE.json
    "reg36_n10000_q4": ("rate0.50_dv03_dc06_N10000", dict(sigma2=0.84 ** 2, max_iters=50, nq_cha=16, nq_msg=16)),
    "dvbs2_q4": ("rate0.50_irreg_dvbs2_N64800", dict(sigma2=0.88 ** 2, max_iters=50, nq_cha=16, nq_msg=16, allow_deg1=True)),
    "dvbs2_q4_i6": ("rate0.50_irreg_dvbs2_N64800", dict(sigma2=0.88 ** 2, max_iters=6, nq_cha=16, nq_msg=16, allow_deg1=True)),
    "twin64800_q4_i6": ("rate0.50_dv02-08_dc07-08_lut_q4_N64800", dict(sigma2=0.88 ** 2, max_iters=6, nq_cha=16, nq_msg=16)),
}


@functools.lru_cache(maxsize=None)
def oracle_codec(name: str):
    """Load the alist and design the LUTs with the oracle (cached per test session)."""
    alist, kw = CONFIGS[name]
    kw = dict(kw)
    code = orc.Code(CODES / f"{alist}.alist")
    cd = orc.Codec(code, skip_rank=True)
    rank = kw.pop("rank", code.nchk)
    cd.set_rank(rank)
    max_iters = kw.pop("max_iters")
    nq = kw.pop("nq_msg")
    nq_msg = np.full(max_iters, nq, np.int32) if np.isscalar(nq) else np.asarray(nq, np.int32)
    snr = kw.pop("design_snr_db", None)
    sigma2 = kw.pop("sigma2")
    if sigma2 is None:   # src/LDPC_BER_Sim.cpp:482
        rate = 1.0 - rank / code.nvar
        sigma2 = 10 ** (-snr / 10) / (2 * rate)
    cd.design_luts(sigma2=sigma2, max_iters=max_iters, nq_msg=nq_msg, **kw)
    cd.rate = 1.0 - rank / code.nvar
    return cd


def product_decoder(cd, device=0):
    """Build the product's decoder from the oracle-designed tables (the reference's own tree text)."""
    import lut_ldpc_amd as L
    c = cd.code
    chk = "" if cd.min_lut else cd.chk_tree_txt
    return L.Decoder(c.nvar, c.nchk, c.dv, c.dc, c.cn_msg_idx, cd.nq_cha, cd.nq_msg, cd.reuse_vec, cd.max_iters, cd.min_lut,
                     cd.var_tree_txt, chk, device=device)


def awgn_labels(cd, B, snr_db, seed, mode=0):
    """All-zero codeword over BPSK/AWGN, LLR = 4x/N0 (src/LDPC_BER_Sim.cpp:248-278), quantised with the
    designed boundaries (src/LDPC_Code_LUT.cpp:207-221).  Returns (cha, msg0) uint8 [B, nvar]."""
    rng = np.random.default_rng(seed)
    N0 = 10 ** (-snr_db / 10) / cd.rate
    x = 1.0 + rng.normal(0.0, np.sqrt(N0 / 2), (B, cd.code.nvar))
    llr = 4 * x / N0
    cha = orc.quant_nonlin(llr, cd.qb_cha)
    msg = orc.quant_nonlin(llr, cd.qb_msg) if mode == 0 else cd.cha2msg_map[cha].astype(np.uint8)
    return cha, msg, llr


def compare(cd, dec, cha, msg, psc, pisc, max_iters=None, flat=False):
    """Decode the same labels with the oracle and through the C-ABI: every decided bit and every returned iteration
    code (src/LDPC_Code_LUT.cpp:259-353) must be equal.  Returns the iteration codes."""
    I = max_iters or cd.max_iters
    cd.set_exit_conditions(I, psc, pisc)
    dec.set_exit_conditions(I, psc, pisc)
    # long codes: the oracle's flat-table mode on all cores (bit-identical to its faithful mode: tests/test_oracle_flat.py)
    want_bits, want_it = (cd.lut_decode_batch_flat if (flat or cd.code.nvar >= 10000) else cd.lut_decode_batch)(cha, msg)
    got_bits, got_it = dec.lut_decode_batch(cha, msg)
    assert (want_it == got_it).all(), (np.flatnonzero(want_it != got_it)[:8], want_it[:8], got_it[:8])
    bad = np.argwhere(want_bits != got_bits)
    assert bad.size == 0, f"{len(bad)} bit mismatches, first at frame/bit {bad[:4].tolist()}"
    return want_it


def write_ira_alist(path, K, M, dv_info, seed=0):
    """A dual-diagonal (IRA / DVB-S2 style) code for the tests: K information nodes of degree dv_info spread evenly over M checks
    plus M parity nodes of degree 2 in a (tail-biting) zigzag, parity node j joining checks j and j+1.  Check degree
    K*dv_info/M + 2.  Returns (N, M)."""
    rng = np.random.default_rng(seed)
    assert (K * dv_info) % M == 0
    per = K * dv_info // M
    sockets = np.repeat(np.arange(M), per)
    rng.shuffle(sockets)
    cols = sockets.reshape(K, dv_info).copy()
    for _ in range(100):                                   # repair columns that got the same check twice by swapping sockets
        bad = [v for v in range(K) if len(set(cols[v])) < dv_info]
        if not bad:
            break
        for v in bad:
            for k in range(1, dv_info):
                if cols[v][k] in cols[v][:k]:
                    w = int(rng.integers(K))
                    j = int(rng.integers(dv_info))
                    if cols[w][j] not in cols[v] and cols[v][k] not in cols[w]:
                        cols[v][k], cols[w][j] = cols[w][j], cols[v][k]
    assert all(len(set(c)) == dv_info for c in cols)
    col_rows = [sorted(int(r) for r in c) for c in cols] + [sorted({j, (j + 1) % M}) for j in range(M)]
    N = K + M
    row_cols = [[] for _ in range(M)]
    for v, rows in enumerate(col_rows):
        for r in rows:
            row_cols[r].append(v)
    with open(path, "w") as f:
        f.write(f"{N} {M}\n{max(len(c) for c in col_rows)} {max(len(r) for r in row_cols)}\n")
        f.write(" ".join(str(len(c)) for c in col_rows) + "\n" + " ".join(str(len(r)) for r in row_cols) + "\n")
        for c in col_rows:
            f.write(" ".join(str(r + 1) for r in c) + "\n")
        for r in row_cols:
            f.write(" ".join(str(v + 1) for v in sorted(r)) + "\n")
    return N, M


def write_random_alist(path, N, M, dv_choices, dv_probs, seed=0):
    """A random irregular code for the fuzz tests: variable degrees drawn from dv_choices with dv_probs, edges dealt to the M
    checks as evenly as possible (check degrees differ by at most one), double edges repaired by swapping sockets.
    Returns (dv, dc) as arrays."""
    rng = np.random.default_rng(seed)
    dv = rng.choice(dv_choices, size=N, p=dv_probs).astype(int)
    E = int(dv.sum())
    sockets = np.arange(E) % M                                  # check of every socket: degrees E // M or E // M + 1
    rng.shuffle(sockets)
    ptr = np.concatenate([[0], np.cumsum(dv)])
    cols = [list(sockets[ptr[v]:ptr[v + 1]]) for v in range(N)]
    for _ in range(200):
        bad = [v for v in range(N) if len(set(cols[v])) < len(cols[v])]
        if not bad:
            break
        for v in bad:
            for k in range(1, len(cols[v])):
                if cols[v][k] in cols[v][:k]:
                    w = int(rng.integers(N)); j = int(rng.integers(len(cols[w])))
                    if cols[w][j] not in cols[v] and cols[v][k] not in cols[w]:
                        cols[v][k], cols[w][j] = cols[w][j], cols[v][k]
    assert all(len(set(c)) == len(c) for c in cols)
    col_rows = [sorted(int(r) for r in c) for c in cols]
    row_cols = [[] for _ in range(M)]
    for v, rows in enumerate(col_rows):
        for r in rows:
            row_cols[r].append(v)
    assert min(len(r) for r in row_cols) >= 2
    with open(path, "w") as f:
        f.write(f"{N} {M}\n{max(len(c) for c in col_rows)} {max(len(r) for r in row_cols)}\n")
        f.write(" ".join(str(len(c)) for c in col_rows) + "\n" + " ".join(str(len(r)) for r in row_cols) + "\n")
        for c in col_rows:
            f.write(" ".join(str(r + 1) for r in c) + "\n")
        for r in row_cols:
            f.write(" ".join(str(v + 1) for v in sorted(r)) + "\n")
    return dv, np.array([len(r) for r in row_cols])
