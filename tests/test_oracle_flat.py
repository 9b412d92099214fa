"""The oracle's flat-table, multi-threaded mode (oracle/or_flat.c, the CPU baseline leg (ii) of SURVEY 8d) against its
faithful mode (per-output queue copy + recursive walk): every bit and every iteration code, all exit modes, every tree
shape the test configurations have (balanced, root_only, auto_bin_high, file trees, CHKTREE checks, mixed alphabets)."""
import numpy as np
import pytest

from helpers import awgn_labels, oracle_codec


@pytest.mark.parametrize("name,B,snr", [("n500_q4", 40, 1.8), ("reg36_n1000_mixed", 48, 2.2), ("reg36_n1000_q5", 24, 1.9),
                                        ("reg36_n1000_q3_chklut", 24, 2.5), ("reg36_n1000_rootonly", 24, 2.5), ("reg36_n1000_high", 24, 2.0),
                                        ("c5_minlut", 32, 4.0), ("c5_chklut", 12, 4.2), ("dvbs2_q4_i6", 6, 1.0)])
def test_flat_mode_equals_faithful_mode(name, B, snr):
    cd = oracle_codec(name)
    mode = 1 if name.startswith("c5") else 0
    cha, msg, _ = awgn_labels(cd, B, snr, seed=11, mode=mode)
    cha[0] = cd.nq_cha - 1; msg[0] = cd.nq_msg[0] - 1          # noise-free frame: pisc returns 0
    for psc, pisc in [(False, False), (True, False), (True, True)]:
        cd.set_exit_conditions(cd.max_iters, psc, pisc)
        wb, wi = cd.lut_decode_batch(cha, msg)
        for threads in (1, 4):
            gb, gi = cd.lut_decode_batch_flat(cha, msg, threads=threads)
            assert (gi == wi).all(), (psc, pisc, threads, gi[:8], wi[:8])
            assert (gb == wb).all()
