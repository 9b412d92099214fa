"""TEST INFRASTRUCTURE: replay the call sequences of the GPU test-suite against the sanitizer build of the library
with the do-nothing HIP runtime (tests/fakehip/fake_hip.c).  Kernels do not run, so results are not compared;
what is checked is everything the HOST half does around them -- buffer sizing, copies, role / item tables, graph
capture, handle life cycle -- under AddressSanitizer + UBSan.  The oracle runs for real (sanitized too), because in
the parity tests it shares the process with the product.

Run through tests/test_host_dryrun_asan.py (which sets LD_PRELOAD / LUTLDPC_LIB / LUTLDPC_ORACLE_LIB), or by hand:
    make -C tests/fakehip && python tests/fakehip/replay.py --launch
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
ASAN_RT = "/opt/rocm/lib/llvm/lib/clang/{ver}/lib/linux/libclang_rt.asan-x86_64.so"


def sanitizer_env():
    import glob
    rt = sorted(glob.glob(ASAN_RT.format(ver="*")))
    if not rt:
        raise RuntimeError("clang's shared ASan runtime not found under /opt/rocm/lib/llvm")
    env = dict(os.environ)
    env.update({
        "LD_PRELOAD": rt[-1],
        "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1:strict_string_checks=1:detect_stack_use_after_return=1",
        "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1",
        "LUTLDPC_LIB": str(HERE / "_build" / "liblut_ldpc_amd_asan.so"),
        "LUTLDPC_ORACLE_LIB": str(HERE / "_build" / "liblut_ldpc_oracle_asan.so"),
    })
    return env


def launch(args):
    subprocess.run(["make", "-s", "-j8", "-C", str(HERE)], check=True)
    return subprocess.run([sys.executable, str(Path(__file__).resolve())] + args, env=sanitizer_env(), cwd=str(ROOT))


def scenarios(which):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import ctypes as C
    import numpy as np
    from helpers import awgn_labels, oracle_codec, product_decoder
    import lut_ldpc_amd as L

    fake = C.CDLL(str(HERE / "_build" / "libfakehip.so"))
    fake.fakehip_launches_of.argtypes = [C.c_char_p]
    fake.fakehip_launches_of.restype = C.c_long
    for f in ("fakehip_launches", "fakehip_graph_launches", "fakehip_captures", "fakehip_live_allocations", "fakehip_live_bytes"):
        getattr(fake, f).restype = C.c_long

    def run(name, B, snr, env, modes, repeats=1, with_oracle=True):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            cd = oracle_codec(name)
            dec = product_decoder(cd)
            mode = 1 if name.startswith("c5") else 0
            cha, msg, _ = awgn_labels(cd, B, snr, seed=77, mode=mode)
            for psc, pisc in modes:
                cd.set_exit_conditions(cd.max_iters, psc, pisc)
                dec.set_exit_conditions(cd.max_iters, psc, pisc)
                if with_oracle and cd.code.nvar <= 2048 and B <= 400:
                    cd.lut_decode_batch(cha, msg)
                for _ in range(repeats):
                    dec.lut_decode_batch(cha, msg)
            desc = dec.describe()
            dec.close()
            return desc
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    all3 = [(False, False), (True, False), (True, True)]
    done = []
    if which == "quick":
        # the CPU suite's share: the decode the round-1 driver run died in (first fused launch of the widest bucket, byte rows,
        # ragged second group), the graph path, compaction, a JIT configuration and DVB-S2 with chain fusion
        run("n500_q4_i8", 300, 1.8, {"LUTLDPC_PACK": "1"}, [(True, True), (False, False)])
        run("n500_q4_i8", 1100, 1.8, {}, all3, repeats=3, with_oracle=False)
        run("n500_q4_i8", 1100, 1.8, {"LUTLDPC_VALIDATE": "1"}, all3, with_oracle=False)
        run("reg36_n1000_mixed", 1025, 2.2, {"LUTLDPC_COMPACT": "1", "LUTLDPC_COMPACT_FIRST": "2", "LUTLDPC_COMPACT_EVERY": "1", "LUTLDPC_COMPACT_MARGIN": "0"}, all3, repeats=3, with_oracle=False)
        run("c5_chklut", 20, 4.2, {}, [(True, True)])
        run("dvbs2_q4_i6", 1030, 1.0, {}, [(True, True), (False, False)], repeats=3, with_oracle=False)
        assert fake.fakehip_captures() > 0 and fake.fakehip_graph_launches() > 0 and fake.fakehip_launches_of(b"pass_fused_kernel") > 0
        # the [BP] comparison decoder: growing and shrinking batches, both entry points
        from oracle import oracle as orc
        code = orc.Code(ROOT / "data" / "codes" / "rate0.50_dv02-17_dc08-09_lut_q4_N500.alist")
        bp = L.BPDecoder(code.nvar, code.nchk, code.dv, code.dc, code.cn_msg_idx, device=0)
        for B in (3, 700, 40):
            bp.set_exit_conditions(5, True, True)
            bp.decode_llr_batch(np.ones((B, code.nvar)), want_qllr=True)
            bp.decode_qllr_batch(np.ones((B, code.nvar), np.int32))
        bp.close()
        assert fake.fakehip_launches_of(b"bp_cn_kernel") > 0
        done.append("quick")
    if which in ("all", "parity"):
        # tests/test_decode_parity_gpu.py, in collection order: the 27 oracle-parity cases, then the knob variants
        for name, B, snr in [("n500_q4", 70, 1.8), ("reg36_n1000_q4", 300, 1.6), ("reg36_n1000_mixed", 64, 2.2), ("reg36_n1000_q5", 40, 1.9),
                             ("reg36_n1000_q3_chklut", 40, 2.5), ("reg36_n1000_rootonly", 33, 2.5), ("reg36_n1000_high", 33, 2.0),
                             ("c5_minlut", 48, 4.0), ("c5_chklut", 20, 4.2)]:
            for m in all3:
                run(name, B, snr, {}, [m])
            done.append(name)
        for env in [{"LUTLDPC_PACK": "1"}, {"LUTLDPC_USE_FAST": "0"}, {"LUTLDPC_PACK": "1", "LUTLDPC_USE_FAST": "0"}]:
            for name, B, snr in [("n500_q4", 300, 1.8), ("reg36_n1000_mixed", 64, 2.2), ("c5_chklut", 20, 4.2), ("reg36_n1000_high", 33, 2.0)]:
                run(name, B, snr, env, [(True, True), (False, False)])
        done.append("kernel_variants")
    if which in ("all", "skew"):
        # skewed pipeline, graph replay (three calls with one key: plain, capture, replay), compaction, no chain
        for env in [{}, {"LUTLDPC_PACK": "1"}, {"LUTLDPC_SKEW": "0"}, {"LUTLDPC_COMPACT": "1", "LUTLDPC_COMPACT_FIRST": "2", "LUTLDPC_COMPACT_EVERY": "1", "LUTLDPC_COMPACT_MARGIN": "0"}, {"LUTLDPC_CHAIN": "0"},
                    {"LUTLDPC_GRAPH": "0"}]:
            for name, B, snr in [("n500_q4", 1100, 1.6), ("reg36_n1000_q4", 1537, 2.0), ("reg36_n1000_mixed", 1025, 2.2)]:
                run(name, B, snr, env, [(True, True), (True, False), (False, False)], repeats=3, with_oracle=False)
        assert fake.fakehip_captures() > 0 and fake.fakehip_graph_launches() > 0, "the graph path was not exercised"
        assert fake.fakehip_launches_of(b"pass_fused_kernel") > 0, "the fused kernel was never launched"
        done.append("skew")
    if which in ("all", "big"):
        # the N=64800 codes: chain fusion (first bucket) and the middle bucket, batches of several groups, growing and shrinking
        for name in ("dvbs2_q4_i6", "twin64800_q4_i6"):
            cd = oracle_codec(name)
            dec = product_decoder(cd)
            for B in (1030, 513, 2100, 96):
                cha, msg, _ = awgn_labels(cd, B, 1.0, seed=B)
                for psc, pisc in all3:
                    dec.set_exit_conditions(cd.max_iters, psc, pisc)
                    dec.lut_decode_batch(cha, msg)
                    dec.lut_decode_batch(cha, msg)
            dec.close()
        done.append("big")
    live = fake.fakehip_live_allocations()
    print(f"replay ok: {done}; launches {fake.fakehip_launches()}, graph launches {fake.fakehip_graph_launches()}, device allocations still live {live}")
    assert live == 0, f"{live} device allocations ({fake.fakehip_live_bytes()} bytes) were never freed"


if __name__ == "__main__":
    if "--launch" in sys.argv:
        sys.exit(launch([a for a in sys.argv[1:] if a != "--launch"]).returncode)
    scenarios(sys.argv[1] if len(sys.argv) > 1 else "all")
