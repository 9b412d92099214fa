"""ber_sim, sharded over the GPUs of one node.

    python -m lut_ldpc_amd.ber_sim -p <params.ini> [-b <basedir>] [-s <seed>] [-c <custom-name>]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m lut_ldpc_amd.ber_sim -p ...

Same command line and parameter file as the reference's prog/ber_sim.cpp; the simulation object is
the C++ LDPC_BER_Sim_LUT of lut_ldpc_amd/csrc/host (through include/lut_ldpc_host.h).  With more than
one process, one rank drives one GPU: the frames of every SNR point are dealt to the ranks in
contiguous batches, and the only data exchanged are the BER/FER counters -- the reference's
multi-host recipe (independent `-s` seeds summed by scripts/aggregate_results.m:73-84) turned into
one RCCL exchange per round.

The stop rule of sim_snr_point (src/LDPC_BER_Sim.cpp:289: stop after the frame that makes the
frame-error count EXCEED Nfers) is applied in global frame order, so the counters are identical to a
single-process, frame-by-frame run over the same Philox-addressed frames, for any number of ranks.
Per round: (1) all-gather of the per-rank batch totals, (2) all-reduce of each rank's contribution --
its whole batch if it lies before the stopping frame, a truncated prefix if it contains it, nothing
after.  Both payloads are a few int64.
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys
import time
from typing import Callable, Optional

import numpy as np

# ------------------------------------------------------------------------------------------------
# sharded frame loop (pure logic: the batch function may be the GPU, or the oracle in the CPU tests)
# ------------------------------------------------------------------------------------------------


class Comm:
    """Minimal collective interface: single process, or torch.distributed (RCCL on GPUs, gloo on CPU)."""

    def __init__(self, dist=None, device=None):
        self.dist, self.device = dist, device
        self.rank = dist.get_rank() if dist else 0
        self.world = dist.get_world_size() if dist else 1

    def all_gather_i64(self, vec: np.ndarray) -> np.ndarray:
        if not self.dist:
            return vec[None, :].copy()
        import torch
        t = torch.from_numpy(np.ascontiguousarray(vec, np.int64)).to(self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.stack([o.cpu().numpy() for o in out])

    def all_reduce_sum_i64(self, vec: np.ndarray) -> np.ndarray:
        if not self.dist:
            return vec.copy()
        import torch
        t = torch.from_numpy(np.ascontiguousarray(vec, np.int64)).to(self.device)
        self.dist.all_reduce(t)
        return t.cpu().numpy()


def _prefix_until_stop(stats: np.ndarray, K: int, nfers: int, ferr_before: int):
    """Counters of the frames of `stats` (in order) up to and including the one whose frame error makes
    the running count exceed nfers.  Returns (counters[5], stopped)."""
    fe = (stats[:, 1] != 0).astype(np.int64)
    run = ferr_before + np.cumsum(fe)
    hit = np.flatnonzero(run > nfers)
    n = int(hit[0]) + 1 if hit.size else len(stats)
    s = stats[:n]
    return np.array([n, n * K, fe[:n].sum(), s[:, 2].sum(), s[:, 3].sum()], np.int64), bool(hit.size)


def sim_snr_point_sharded(batch_fn: Callable[[int, int], np.ndarray], nframes: int, nfers: int, K: int, comm: Comm,
                          batch_max: int = 4096, batch_first: int = 512):
    """Frame loop of sim_snr_point (src/LDPC_BER_Sim.cpp:260-291) over `comm.world` ranks.

    batch_fn(frame0, B) -> int array [B, 4] = {iters, frame error, data bit errors, uncoded errors} of
    frames frame0..frame0+B-1.  Returns counters[5] = {frames, data bits, frame errors, data bit errors,
    uncoded bit errors}, identical on every rank."""
    total = np.zeros(5, np.int64)
    f0, batch = 0, min(batch_first, batch_max)
    while f0 < nframes:
        lo = min(nframes, f0 + comm.rank * batch)
        hi = min(nframes, lo + batch)
        stats = batch_fn(lo, hi - lo) if hi > lo else np.zeros((0, 4), np.int32)
        stats = np.asarray(stats).reshape(-1, 4)
        mine = np.array([len(stats), (stats[:, 1] != 0).sum()], np.int64)
        allr = comm.all_gather_i64(mine)                          # (1) totals of every rank's batch
        ferr_before = int(total[2]) + int(allr[:comm.rank, 1].sum())
        first_stop = None                                         # first rank whose batch crosses the limit
        run = int(total[2])
        for r in range(comm.world):
            run += int(allr[r, 1])
            if run > nfers:
                first_stop = r
                break
        if first_stop is None or comm.rank < first_stop:
            contrib, _ = _prefix_until_stop(stats, K, 1 << 62, 0)
        elif comm.rank == first_stop:
            contrib, _ = _prefix_until_stop(stats, K, nfers, ferr_before)
        else:
            contrib = np.zeros(5, np.int64)
        total += comm.all_reduce_sum_i64(contrib)                 # (2) the only payload: five counters
        if first_stop is not None:
            break
        f0 += comm.world * batch
        batch = min(batch * 4, batch_max)
    return total


# ------------------------------------------------------------------------------------------------
# the simulation object (C++ LDPC_BER_Sim_LUT) and the command line
# ------------------------------------------------------------------------------------------------


class BerSim:
    """LDPC_BER_Sim_LUT (constructor + load()) driven batch by batch."""

    def __init__(self, params_path, base_dir, seed=0, custom_name="", device=0):
        from ._capi import lib, check
        self._lib, self._check = lib, check
        self._h = C.c_void_p()
        check(lib.lutldpc_bersim_create(str(params_path).encode(), str(base_dir).encode(), int(seed), custom_name.encode(), int(device),
                                        C.byref(self._h)))
        info = (C.c_int64 * 8)()
        limits = (C.c_double * 2)()
        snr = (C.c_double * 256)()
        check(lib.lutldpc_bersim_info(self._h, info, limits, snr, 256))
        self.n_snr, self.nframes, self.nfers, self.nvar, self.ninfo, self.max_iter = (int(info[i]) for i in range(6))
        self.zero_codeword, self.batch_frames = bool(info[6]), int(info[7])
        self.ber_min, self.fer_min = limits[0], limits[1]
        self.snr_db = [snr[i] for i in range(self.n_snr)]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lutldpc_bersim_destroy(self._h)
            self._h = None

    __del__ = close

    def batch(self, snr_index: int, frame0: int, B: int) -> np.ndarray:
        stats = np.empty((B, 4), np.int32)
        self._check(self._lib.lutldpc_bersim_batch(self._h, int(snr_index), int(frame0), int(B), stats.ctypes.data_as(C.POINTER(C.c_int32))))
        return stats

    def add_point(self, snr: float, counters):
        c = np.ascontiguousarray(counters, np.int64)
        self._check(self._lib.lutldpc_bersim_add_point(self._h, float(snr), c.ctypes.data_as(C.POINTER(C.c_int64))))

    def save(self, runtime_s: float):
        self._check(self._lib.lutldpc_bersim_save(self._h, float(runtime_s)))

    def results_path(self) -> str:
        n = self._lib.lutldpc_bersim_results_path(self._h, None, 0)
        buf = C.create_string_buffer(int(n))
        self._lib.lutldpc_bersim_results_path(self._h, buf, n)
        return buf.value.decode()


def _placement_policy(params, world: int):
    """The same rule as the C++ driver (ber_sim_driver.cpp, ber_sim_main): the placement search of the row buffers (0.1-0.4 s per
    batch size for +2-6 % of the streaming kernels' rate) is left to runs in which a rank sees at least 64 full batches of an
    SNR point; LUTLDPC_PLACE set by the user wins (the library reads it when the decoder is created)."""
    import re
    try:
        txt = open(params).read()
    except OSError:
        return False
    def key(name, default):
        m = re.search(r"^[ \t]*" + name + r"[ \t]*=[ \t]*([-+0-9.eE]+)", txt, re.M)
        try:
            return float(m.group(1)) if m else default
        except ValueError:
            return default
    if "LUTLDPC_PLACE" not in os.environ and key("Nframes", 1e2) / (max(world, 1) * max(1.0, key("batch_frames", 32768.0))) < 64.0:
        os.environ["LUTLDPC_PLACE"] = "0"
        return True
    return False


def _run(params, base_dir, seed=0, custom_name="", comm: Optional[Comm] = None, device=0, save=True, quiet=False, batch_override=None):
    """LDPC_BER_Sim::run (src/LDPC_BER_Sim.cpp:121-155) + save(), sharded over comm.

    batch_override(sim, snr_index, frame0, B) -> [B, 4] replaces the device batch (sampler + decode + counting); the CPU
    tests of the multi-rank path pass the oracle there, with the simulation object created host-only (device = -1)."""
    comm = comm or Comm()
    sim = BerSim(params, base_dir, seed, custom_name, -1 if batch_override else device)
    batch = (lambda i, f, b: batch_override(sim, i, f, b)) if batch_override else sim.batch
    t0 = time.perf_counter()
    points = []
    stop_sweep = False
    for idx, snr in enumerate(sim.snr_db):
        if stop_sweep:
            c = np.zeros(5, np.int64)                              # remaining points are padded, :142-149
        else:
            c = sim_snr_point_sharded(lambda f, b: batch(idx, f, b), sim.nframes, sim.nfers, sim.ninfo, comm, sim.batch_frames)
            ber = c[3] / c[1] if c[1] else 0.0
            fer = c[2] / c[0] if c[0] else 0.0
            if comm.rank == 0 and not quiet:
                print(f"SNR = {snr:g}  Simulated {c[0]} frames and {c[1]} data bits. Obtained {c[3]} data bit errors.  "
                      f"Data BER: {ber:g} Uncoded BER: {c[4] / (c[0] * sim.nvar) if c[0] else 0:g} FER: {fer:g}", flush=True)
            stop_sweep = ber < sim.ber_min or fer < sim.fer_min    # :307
        points.append((snr, c))
        sim.add_point(snr, c)
    runtime = time.perf_counter() - t0
    path = None
    if comm.rank == 0:
        if save:
            sim.save(runtime)
            path = sim.results_path()
        if not quiet:
            print(f"Done simulating. Runtime = {runtime:g} seconds", flush=True)
    sim.close()
    return points, path


def run(params, base_dir, seed=0, custom_name="", comm: Optional[Comm] = None, device=0, save=True, quiet=False, batch_override=None):
    """`_run` under the placement policy of this run (the decoder is created at the first batch, so the variable stays set for the
    whole run and is removed afterwards: a caller's later decoders decide for themselves)."""
    comm = comm or Comm()
    policy_set = _placement_policy(params, comm.world)
    try:
        return _run(params, base_dir, seed, custom_name, comm, device, save, quiet, batch_override)
    finally:
        if policy_set:
            os.environ.pop("LUTLDPC_PLACE", None)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="ber_sim", description="LUT-LDPC BER simulation on MI355X (drop-in for the reference's ber_sim)")
    ap.add_argument("-b", "--basedir", default=os.getcwd(), help="paths in params files are relative to this directory")
    ap.add_argument("-c", "--custom-name", default="", help="append this string at the end of the results file name")
    ap.add_argument("-p", "--params", help="input parameter file")
    ap.add_argument("-s", "--seed", type=int, default=0, help="random seed")
    args = ap.parse_args(argv)
    if not args.params:
        print("No input parameters specified. To learn more, use the --help option.")
        return 0
    if not os.path.isabs(args.basedir):
        print("Base directory must be specified as absolut path")
        return 0
    params = args.params if os.path.isabs(args.params) else os.path.join(args.basedir, args.params)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    comm = Comm()
    # LUTLDPC_DIST_FORCE=1: a process group even for ONE rank -- on a one-GPU box this is the only way to send the counter exchange
    # through RCCL proper (all_gather / all_reduce of int64 on the device)
    grouped = world > 1 or bool(os.environ.get("LUTLDPC_DIST_FORCE"))
    if grouped:
        import torch
        import torch.distributed as dist
        use_gpu = torch.cuda.is_available()
        # LUTLDPC_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share cards,
        # counters travel over gloo); the default on GPUs is RCCL, one card per rank
        backend = os.environ.get("LUTLDPC_DIST_BACKEND", "nccl" if use_gpu else "gloo")
        if use_gpu and backend != "nccl":
            local %= torch.cuda.device_count()
        if use_gpu:
            torch.cuda.set_device(local)
        dist.init_process_group(backend)
        comm = Comm(dist, torch.device("cuda", local) if backend == "nccl" else torch.device("cpu"))
    run(params, args.basedir, args.seed, args.custom_name, comm, device=local)
    if grouped:
        comm.dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
