#!/usr/bin/env python3
"""bench.py -- PBS/s of the batched shortint KS-PBS at PARAM_MESSAGE_2_CARRY_2_KS_PBS.

One "step" = one pass of the hot path (keyswitch + blind rotation + sample extract) over a batch
of 1024 independent big-LWE ciphertexts per GPU, inputs/keys/LUTs already resident in HBM
(BASELINE.json configs[1]).  Synthetic data: uniformly random keys and ciphertexts (the kernels'
work is data-independent; real keys are uniform-looking u64 as well).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
N > 1: launched by torch.distributed.run, one rank per GPU, every rank bootstraps its own shard of
B ciphertexts (weak scaling, keys replicated) and the output LWEs are all-gathered over RCCL.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# SURVEY.md section 8(d) / BASELINE.md section 2: algorithmic bytes per PBS
A_BSK = 742 * 1 * 2 * 2 * 1024 * 16          # 48 627 712  Fourier bootstrap key
A_KSK = 2048 * 5 * 743 * 8                   # 60 866 560  keyswitch key
A_BIG = 2049 * 8                             # 16 392      big LWE (in / out)
A_SMALL = 743 * 8                            # 5 944       small LWE (keyswitch out / blind-rotate in)
A_LUT = 2 * 2048 * 8                         # 32 768      accumulator
A_PATH = A_BSK + A_KSK + A_BIG + A_BIG + A_LUT          # 109 559 824 B per KS-PBS (whole path)
A_BLIND_ROTATE = A_BSK + A_SMALL + A_LUT + A_BIG        # 48 682 816 B per PBS (dominant kernel)
HBM_PEAK_GBPS = 8000.0                                  # MI355X_MICROARCH.md: 8 TB/s spec


def cgroup_cpu_quota():
    """CPUs this process may use according to its cgroup (v2 cpu.max / v1 cfs quota); None if unlimited."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, int(int(quota) / int(period) + 0.5))
    except Exception:
        pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0:
            return max(1, int(quota / period + 0.5))
    except Exception:
        pass
    return None


def cpu_baseline(params_n, seed=1):
    """Times the CPU oracle (a port of the reference algorithm, `oracle/`) on this box's host cores
    on a bounded sample of the same workload."""
    from oracle import pyoracle as O

    p = O.param_message_2_carry_2(lwe_dim=params_n)
    rng = np.random.default_rng(seed)
    ks = O.KeySet.__new__(O.KeySet)
    ks.p = p
    ks.bsk = rng.integers(0, 2**64, O.lib().orc_bsk_len(p), dtype=np.uint64)
    ks.ksk = rng.integers(0, 2**64, O.lib().orc_ksk_len(p), dtype=np.uint64)
    ks.fft = O.Fft(p.poly_size)
    ks.bsk_re = ks.bsk_im = None
    ks.to_fourier()
    cores = min(os.cpu_count() or 1, O.lib().orc_max_threads())
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    cores = min(cores, cgroup_cpu_quota() or cores)  # a container's CPU share, not the host's core count
    lut, _ = ks.lut([(x * x) % 16 for x in range(16)])
    warm = rng.integers(0, 2**64, (2 * cores, p.big_size), dtype=np.uint64)
    ks.ks_pbs_batch(warm[:cores], lut, np.zeros(cores, dtype=np.uint32), threads=cores)  # warm-up
    t0 = time.perf_counter()
    ks.ks_pbs_batch(warm, lut, np.zeros(2 * cores, dtype=np.uint32), threads=cores)  # calibration
    per_round = (time.perf_counter() - t0) / 2
    # bounded sample: about 10 s of wall time on all usable cores (10 s x cores of CPU work)
    count = int(min(8192, max(4 * cores, cores * round(10.0 / max(per_round, 1e-3)))))
    cts = rng.integers(0, 2**64, (count, p.big_size), dtype=np.uint64)
    idx = np.zeros(count, dtype=np.uint32)
    t0 = time.perf_counter()
    ks.ks_pbs_batch(cts, lut, idx, threads=cores)
    dt = time.perf_counter() - t0
    return {
        "value": count / dt,
        "unit": "PBS/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{count} KS-PBS (n={params_n}, random keys) in one OpenMP batch over {cores} threads, "
                  f"{dt:.2f} s wall; C oracle, -O3 -march=x86-64-v3, scalar radix-4 f64 FFT",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="ciphertexts per GPU per step")
    ap.add_argument("--lwe-dim", type=int, default=742)
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    import torch

    import tfhe_rs_string_amd as T

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(args.backend)
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=args.lwe_dim)
    B = args.batch
    rng = np.random.default_rng(1234)  # same keys on every rank (replicated server key)
    bsk = rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64)
    ksk = rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64)
    ctx = T.Context(local_rank)
    sk = T.ServerKey(ctx, p, bsk, ksk)
    del bsk, ksk
    if args.variant is not None:
        sk.set_variant(args.variant)
    rng = np.random.default_rng(99 + rank)  # every rank its own shard of ciphertexts
    cts = rng.integers(0, 2**64, (B, p.big_size), dtype=np.uint64)
    funcs = [lambda x: x % 4, lambda x: x // 4, lambda x: int(x // 4 == x % 4), lambda x: (x // 4) & (x % 4)]
    luts = np.stack([T.generate_lookup_table(p, f)[0] for f in funcs])
    idx = (np.arange(B) % 4).astype(np.int32)
    d_in = torch.from_numpy(cts.view(np.int64)).to(dev)
    d_luts = torch.from_numpy(luts.view(np.int64)).to(dev)
    d_idx = torch.from_numpy(idx).to(dev)
    d_outs = [torch.empty_like(d_in) for _ in range(2)]
    on_host = world > 1 and args.backend != "nccl"  # gloo rehearsal: stage the results through host memory
    gather = T.ResultGather(dist, d_outs[0].cpu() if on_host else d_outs[0]) if world > 1 else None
    step_no = [0]

    def step():
        d_out = d_outs[step_no[0] & 1]
        step_no[0] += 1
        if gather is not None:
            # the gather that last read this buffer must be done before the kernels rewrite it
            while len(gather.pending) >= 2:
                gather.pending.pop(0).wait()
        sk.ks_pbs_batch_device(d_in, d_luts, d_idx, d_out)
        if gather is not None:
            # the path's only exchange: gather the result LWEs (16 392 B each) on rank 0 over
            # RCCL/xGMI, overlapped with the next batch's bootstrap (double-buffered outputs)
            gather.submit(d_out.cpu() if on_host else d_out)

    for _ in range(args.warmup):
        step()
    if gather is not None:
        gather.drain()
    torch.cuda.synchronize()
    sk.profiling(True)
    sk.profiling_read(0, reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if gather is not None:
        gather.drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    br_ms, br_n = sk.profiling_read(1)
    ks_ms, ks_n = sk.profiling_read(0)
    sk.profiling(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if on_host else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        pbs_per_s = world * B * args.steps / dt
        br_avg_ms = br_ms / max(br_n, 1)
        ks_avg_ms = ks_ms / max(ks_n, 1)
        a_br = A_BSK * args.lwe_dim // 742 + (args.lwe_dim + 1) * 8 + A_LUT + A_BIG
        br_gbps = a_br * B / (br_avg_ms * 1e-3) / 1e9 if br_avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("blind_rotate_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "PBS/sec (PARAM_MESSAGE_2_CARRY_2) at 1/2/4/8 GPUs; % HBM roofline",
            "value": pbs_per_s,
            "unit": "PBS/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 torus + f64 FFT",
            "data": "synthetic",
            "config": {
                "workload": f"batch of {B} independent shortint KS-PBS per GPU, PARAM_MESSAGE_2_CARRY_2_KS_PBS "
                            f"(n={args.lwe_dim}, k=1, N=2048, pbs 23x1, ks 3x5), 4 LUTs, inputs resident in HBM",
                "batch_per_gpu": B,
                "sharding": "ciphertexts sharded across ranks, keys replicated, async RCCL gather of the output LWEs on rank 0" if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "blind_rotate (+fused sample extract)",
                "achieved": br_gbps,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": br_gbps / HBM_PEAK_GBPS,
                "traffic": traffic,
                "bytes_per_unit": a_br,
                "units_per_launch": B,
                "avg_launch_ms": br_avg_ms,
                "keyswitch_avg_launch_ms": ks_avg_ms,
                "whole_path": {
                    "bytes_per_unit": A_PATH,
                    "achieved": pbs_per_s / world * A_PATH / 1e9,
                    "frac": pbs_per_s / world * A_PATH / 1e9 / HBM_PEAK_GBPS,
                },
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.lwe_dim)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
