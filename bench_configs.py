#!/usr/bin/env python3
"""Secondary measurements: the BASELINE.json configs other than the headline batch (which is
bench.py): latency of one KS-PBS, FheString::eq 64/64, FheString::contains 256/8 and FheUint32 mul,
each through the host op-graph layer (include/tfhe_hip_strings.h) on ONE GPU.  Synthetic data (random
keys / ciphertexts: work is data-independent; correctness is covered by tests/test_gpu_ops.py).
Prints one JSON line per config.  Usage: python bench_configs.py [--reps 5]

BASELINE.json config 4 AS STATED -- `FheString::contains` (256-char haystack, 8-char needle) sharded across N GPUs:
  python bench_configs.py --gpus N [--reps 5]
times (i) the windows sharded over N ranks, one rank per GPU under torch.distributed (backend nccl = RCCL; the per-window
booleans -- one big LWE each -- are all-gathered, the OR tree runs on every rank: strings.contains_sharded, the fan-out the
reference gets from rayon in fhe_strings/server_key/contains.rs:77-91 / radix_parallel/comparison.rs:22-28) and
(ii) the same op on ONE process with an N-device C-ABI context (tfhe_hip_context_create(out, device_ids, N): every level of
the op-graph is split over the devices).  Like bench.py the ranks are started by the driver (WORLD_SIZE set) or by this
process itself -- as CHILD processes, before torch is imported or HIP touched, never by re-exec.  One JSON line per mode on
rank 0 with n_ranks_seen, the window range of every rank and the gather time.  `--share-gpu` puts every rank / replica on
GPU 0 (rehearsal on a one-GPU box); `--no-device` runs launch, rendezvous, gather and reduction over gloo without kernels
(value null: NOT a measurement).  Unmeasured on distinct GPUs until the driver's 8-GPU node runs it.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench_configs.py --gpus N`, no launcher above us: start the N ranks as fresh child processes (nothing in THIS
    process has imported torch or touched HIP), relay their JSON lines, exit with their return code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = 0
    for out in proc.stdout:
        s = out.strip()
        if s.startswith("{") and '"config"' in s:
            print(s, flush=True)
            lines += 1
        elif s:
            print(s, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc == 0 and lines == 0:
        print("bench_configs.py: the ranks exited without printing a result line", file=sys.stderr)
        rc = 1
    sys.exit(rc)


class NoDeviceKey:
    """--no-device: stands in for StringServerKey in contains_sharded -- the windows and the tree are NOT computed (zeros), the
    sharding, the gather and the reduction run."""

    def __init__(self, params):
        self.params = params

    def window_count(self, s, pattern):
        return 0 if len(pattern) > len(s) else len(s) - len(pattern) + 1

    def contains_windows(self, s, pattern, b, e):
        return np.full((e - b, self.params.big_size), b, dtype=np.uint64)  # row values name the shard: the gather is checked

    def contains(self, s, pattern):
        return np.zeros(self.params.big_size, dtype=np.uint64)

    def boolean_any(self, blocks):
        self.last_blocks = blocks
        return np.zeros(self.params.big_size, dtype=np.uint64)


def config4_sharded(args):
    """FheString::contains 256 / 8 over args.gpus ranks (this process is one of them) + over an N-device context on rank 0"""
    import torch
    import torch.distributed as dist

    import tfhe_rs_string_amd as T
    from tfhe_rs_string_amd.strings import contains_sharded, shard_range

    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU")
    backend = "gloo" if args.no_device else args.backend
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    else:
        dist.init_process_group(backend)
    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
    rng = np.random.default_rng(7)  # the same keys and strings on every rank (replicated server key, shared operands)
    device = None
    if args.no_device:
        key = NoDeviceKey(p)
        hay, needle = np.zeros((256, 4, p.big_size), dtype=np.uint64), np.zeros((8, 4, p.big_size), dtype=np.uint64)
    else:
        bsk = rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64)
        ksk = rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64)
        ctx = T.Context(local_rank)
        sk = T.ServerKey(ctx, p, bsk, ksk)
        key = T.StringServerKey(sk.handle, p)
        hay = rng.integers(0, 2**64, (256, 4, p.big_size), dtype=np.uint64)
        needle = rng.integers(0, 2**64, (8, 4, p.big_size), dtype=np.uint64)
        if backend == "nccl":
            device = torch.device(f"cuda:{local_rank}")
    nw = key.window_count(hay, needle)
    contains_sharded(key, hay, needle, dist, device)  # warm-up
    if not args.no_device:
        key.stats(reset=True)
    ts, gather, windows_s = [], [], []
    tm = {}
    for _ in range(args.reps):
        dist.barrier()
        t0 = time.perf_counter()
        contains_sharded(key, hay, needle, dist, device, timings=tm)
        dist.barrier()
        ts.append(time.perf_counter() - t0)
        gather.append(tm["gather_s"])
        windows_s.append(tm["windows_s"])
    if args.no_device and rank == 0:  # what the rehearsal can check: every rank's rows arrived in window order
        got = key.last_blocks
        want = np.concatenate([np.full(shard_range(nw, r, world)[1] - shard_range(nw, r, world)[0], shard_range(nw, r, world)[0]) for r in range(world)])
        assert got.shape[0] == nw and np.array_equal(got[:, 0], want.astype(np.uint64)), "gather delivered wrong rows"
    mine = torch.tensor([float(np.median(windows_s)), float(np.median(gather))], dtype=torch.float64)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    if backend == "nccl":
        mine = mine.to(device)
        per_rank = [x.to(device) for x in per_rank]
    dist.all_gather(per_rank, mine)
    pbs = launches = None
    if not args.no_device:
        st = key.stats()
        pbs, launches = st["pbs"] // args.reps, st["launches"] // args.reps
    if rank == 0:
        ms = float(np.median(ts)) * 1e3
        print(json.dumps({
            "config": "FheString::contains, 256-char haystack / 8-char needle, windows sharded over the ranks (BASELINE.json config 4)",
            "ms": None if args.no_device else round(ms, 3), "n_gpus": world, "n_ranks_seen": dist.get_world_size(),
            "windows": nw, "windows_per_rank": [list(shard_range(nw, r, world)) for r in range(world)],
            "windows_ms_per_rank": None if args.no_device else [round(float(x[0]) * 1e3, 3) for x in per_rank],
            "gather_ms_per_rank": [round(float(x[1]) * 1e3, 3) for x in per_rank],
            "pbs_rank0": pbs, "launches_rank0": launches,
            "exchange": f"all_gather of ceil({nw} / {world}) big LWEs of {p.big_size * 8} B per rank ({backend}), then the OR tree on every rank",
            "distinct_gpus": bool(not args.share_gpu and not args.no_device),
            "rehearsal": ("no device: launch / rendezvous / gather / reduction plumbing only, kernels not run" if args.no_device
                          else ("every rank on GPU 0" if args.share_gpu else None)),
            "reference": "256 sequential ORs + 8-deep AND chains, >= 4.5 s critical path (fhe_strings/server_key/contains.rs:77-91)",
            "data": "synthetic"}), flush=True)
    dist.barrier()
    # (ii) one process, N-device C-ABI context: rank 0 only, the other ranks' GPUs are idle by now
    if rank == 0 and not args.no_device:
        devs = [0] * world if args.share_gpu else list(range(world))
        try:
            ctx_n = T.Context(devs)
            sk_n = T.ServerKey(ctx_n, p, bsk, ksk)
            key_n = T.StringServerKey(sk_n.handle, p)
            key_n.contains(hay, needle)
            key_n.stats(reset=True)
            tn = []
            for _ in range(args.reps):
                t0 = time.perf_counter()
                key_n.contains(hay, needle)
                tn.append(time.perf_counter() - t0)
            st = key_n.stats()
            print(json.dumps({
                "config": "FheString::contains, 256 / 8, ONE process over an N-device C-ABI context (tfhe_hip_context_create)",
                "ms": round(float(np.median(tn)) * 1e3, 3), "n_gpus": world, "device_ids": devs, "pbs": st["pbs"] // args.reps,
                "launches": st["launches"] // args.reps, "distinct_gpus": bool(not args.share_gpu),
                "exchange": "every level of the op-graph split contiguously over the devices, rows moved by hipMemcpyPeerAsync (xGMI)",
                "data": "synthetic"}), flush=True)
        except Exception as e:  # a missing peer GPU must not lose the line above
            print(json.dumps({"config": "FheString::contains, 256 / 8, N-device C-ABI context", "error": str(e)}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--gpus", type=int, default=1, help="N > 1: BASELINE.json config 4 as stated (contains 256 / 8 sharded over N GPUs)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo to rehearse N > 1 on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank / replica on GPU 0")
    ap.add_argument("--no-device", action="store_true", help="plumbing rehearsal without a GPU (gloo); NOT a measurement")
    args = ap.parse_args()
    if args.gpus > 1:
        if "WORLD_SIZE" not in os.environ:
            self_launch(args)
        return config4_sharded(args)
    import tfhe_rs_string_amd as T

    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
    rng = np.random.default_rng(7)
    ctx = T.Context(0)
    sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
    key = T.StringServerKey(sk.handle, p)
    lut, _ = T.generate_lookup_table(p, lambda x: x % 4)

    def rnd(*shape):
        return rng.integers(0, 2**64, shape + (p.big_size,), dtype=np.uint64)

    def timed(fn):
        fn()  # warm-up
        key.stats(reset=True)
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            r = fn()  # kept until the clock is read: freeing a 100 MB result is not part of the operation
            ts.append(time.perf_counter() - t0)
            del r
        st = key.stats()
        return float(np.median(ts)) * 1e3, st["pbs"] // args.reps, st["launches"] // args.reps

    def emit(name, ms, pbs, launches, ref):
        print(json.dumps({"config": name, "ms": round(ms, 3), "pbs": pbs, "launches": launches, "n_gpus": 1,
                          "pbs_per_s": round(pbs / ms * 1e3, 1) if pbs else None, "reference": ref, "data": "synthetic"}), flush=True)

    one = rnd(1)
    sk.ks_pbs_batch(one, lut, 0)  # warm-up: the first call allocates the page-locked staging buffer
    ts = []
    for _ in range(max(20, args.reps)):
        t0 = time.perf_counter()
        sk.ks_pbs_batch(one, lut, 0)
        ts.append(time.perf_counter() - t0)
    emit("single shortint KS-PBS (latency, host buffers; median of %d)" % len(ts), float(np.median(ts)) * 1e3, 1, 1,
         "16.6 ms single thread m6i.metal (tfhe/docs/getting_started/benchmarks.md:42)")
    s1, s2 = rnd(64, 4), rnd(64, 4)
    emit("FheString::eq, two 64-char strings", *timed(lambda: key.eq(s1, s2)),
         "64-step sequential AND chain, >= 1.1 s critical path (fhe_strings/server_key/comparisons.rs:184-212)")
    hay, needle = rnd(256, 4), rnd(8, 4)
    emit("FheString::contains, 256-char haystack / 8-char needle", *timed(lambda: key.contains(hay, needle)),
         "256 sequential ORs + 8-deep AND chains, >= 4.5 s critical path (fhe_strings/server_key/contains.rs:77-91)")
    a, b = rnd(16), rnd(16)
    emit("FheUint32 mul (16 blocks)", *timed(lambda: key.mul_parallelized(a, b)),
         "333 ms on m6i.metal, all cores (tfhe/docs/getting_started/benchmarks.md:17)")
    # batch mode: independent operations staged together share their launches (throughput regime)
    col = [rnd(64, 4) for _ in range(64)]

    def eq_column():
        with key.batch():
            return [key.eq(c, s1) for c in col]

    ms, pbs, launches = timed(eq_column)
    emit("64 x FheString::eq (64-char strings) in one batch", ms, pbs, launches, "9.5 ms each when issued one by one")
    col2 = [rnd(32, 4) for _ in range(32)]

    def contains_column():
        with key.batch():
            return [key.contains(c, needle[:4]) for c in col2]

    ms, pbs, launches = timed(contains_column)
    emit("32 x FheString::contains (32-char strings, 4-char needle) in one batch", ms, pbs, launches, None)
    # beyond BASELINE.json: a few of the other string operations on the same key
    s40, pat = rnd(40, 4), rnd(3, 4)
    emit("FheString::le, two 40-char strings", *timed(lambda: key.le(s40, rnd(40, 4))), None)
    emit("FheString::find, 40-char string / 3-char pattern", *timed(lambda: key.find(s40, pat)), None)
    emit("FheString::to_uppercase, 40 chars", *timed(lambda: key.to_uppercase(s40)), None)
    emit("FheString::trim, 40 chars", *timed(lambda: key.trim(s40)), None)
    emit("FheString::push_padding_to_end, 40 chars", *timed(lambda: key.push_padding_to_end(s40)), None)
    emit("FheString::split, 40 chars by a clear char", *timed(lambda: key.split(s40, ",")),
         "41 sequential find + substring rounds (fhe_strings/server_key/split.rs:314-345)")
    emit("FheString::splitn(4), 40 chars by a clear 2-char pattern", *timed(lambda: key.splitn(s40, 4, ", ")), None)
    emit("FheString::split_ascii_whitespace, 40 chars", *timed(lambda: key.split_ascii_whitespace(s40)), None)
    emit("FheString::replace, 40 chars, clear 2-char pattern -> clear 3 chars", *timed(lambda: key.replace(s40, ", ", " - ")),
         "split + re-join of 41 parts (fhe_strings/server_key/replace.rs:20-52)")
    emit("FheString::substring with encrypted bounds, 40 chars", *timed(lambda: key.substring(s40, rnd(4), rnd(4))), None)


if __name__ == "__main__":
    main()
