#!/usr/bin/env python3
"""bench.py -- PBS/s of the batched shortint KS-PBS at PARAM_MESSAGE_2_CARRY_2_KS_PBS.

One "step" = one pass of the hot path (keyswitch + blind rotation + sample extract) over a batch
of 1024 independent big-LWE ciphertexts per GPU, inputs/keys/LUTs already resident in HBM
(BASELINE.json configs[1]).  Synthetic data: uniformly random keys and ciphertexts (the kernels'
work is data-independent; real keys are uniform-looking u64 as well).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

N > 1: one rank per GPU under torch.distributed (backend nccl = RCCL).  Started EITHER by the
driver (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: WORLD_SIZE is
set) OR plainly as `python bench.py --gpus N`: then this process -- before it imports torch or
touches HIP -- starts the N ranks itself as a child `torch.distributed.run`, relays rank 0's JSON
line and exits with the children's return code (the fan-out the reference gets from rayon's
par_iter over blocks, tfhe/src/integer/server_key/radix_parallel/comparison.rs:22-28).
Every rank bootstraps its own shard of B ciphertexts (weak scaling, keys replicated); the output
LWEs are gathered on rank 0 over RCCL, asynchronously.  Prints ONE JSON line on rank 0.
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

# SURVEY.md section 8(d) / BASELINE.md section 2: algorithmic bytes per PBS
A_BSK = 742 * 1 * 2 * 2 * 1024 * 16          # 48 627 712  Fourier bootstrap key
A_KSK = 2048 * 5 * 743 * 8                   # 60 866 560  keyswitch key
A_BIG = 2049 * 8                             # 16 392      big LWE (in / out)
A_SMALL = 743 * 8                            # 5 944       small LWE (keyswitch out / blind-rotate in)
A_LUT = 2 * 2048 * 8                         # 32 768      accumulator
A_PATH = A_BSK + A_KSK + A_BIG + A_BIG + A_LUT          # 109 559 824 B per KS-PBS (whole path)
HBM_PEAK_GBPS = 8000.0                                  # MI355X_MICROARCH.md: 8 TB/s spec
# vector FP64 (the blind rotation uses no MFMA): AMD's public MI355X specification, 78.6 TFLOP/s = 256 CUs x 4 SIMDs x 16 lanes x
# 2 flop (fma) x 2.4 GHz -- half the vector FP32 rate MI355X_MICROARCH.md lists (157.3); the guide itself has no FP64 row
FP64_VECTOR_PEAK_TFLOPS = 78.6
INT8_MFMA_PEAK_TOPS = 5000.0                            # MI355X_MICROARCH.md: I8 = 2 x BF16 per clock, dense (~2.5 PF bf16)


def schedule_flops_per_cmux(fn=1024, glwe_size=2, levels=1):
    """f64 floating-point operations of ONE CMUX step (one external product), counted from the
    arithmetic specification of DESIGN.md section 4 ("FFT schedule v1"), fma = 2 flops:
      complex product y*w / x*conj(w): 2 mul + 2 fma = 6 flops; multiply-add: 4 fma = 8 flops;
      radix-4 butterfly: 16 additions; a stage multiplies 3 outputs of every butterfly with j != 0."""
    stages = 0
    while (1 << (2 * stages)) < fn:
        stages += 1
    fft = 0
    for s in range(stages):
        L = fn >> (2 * s)
        q = L // 4
        blocks = fn // L
        fft += (fn // 4) * 16                     # butterflies
        fft += blocks * (q - 1) * 3 * 6           # twiddle products (none for j = 0)
    twist = fn * 6
    polys_fwd = glwe_size * levels                # decomposed polynomials transformed forward
    pointwise = glwe_size * fn * (6 + 8 * (glwe_size * levels - 1))  # first row mul, later rows mul_add
    return polys_fwd * (twist + fft) + glwe_size * (fft + twist) + pointwise


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


def usable_cores():
    cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return min(cores, cgroup_cpu_quota() or cores)  # a container's CPU share, not the host's core count


def cpu_baseline(params_n, cores, seed=1):
    """Times the tuned CPU port of the reference algorithm (`oracle/tfhe_oracle_fast.c`, -march=native bench build) on this
    box's host cores on a bounded sample of the same workload: REAL keys (oracle key generation), fresh
    encryptions, (i) one thread, one ciphertext at a time -- beside the reference's published 16.6 ms --
    and (ii) all usable cores, threads pinned, one OpenMP batch."""
    from oracle import pyoracle as O

    O.use_bench_build()                      # oracle/liboracle_bench.so: same source, -O3 -march=native
    L = O.lib()
    p = O.param_message_2_carry_2(lwe_dim=params_n)
    cores = max(1, min(cores, L.orc_max_threads()))
    t0 = time.perf_counter()
    ks = O.KeySet(p, seed=0x5EED0001 + seed)  # real bootstrap / keyswitch keys (binary secrets, Gaussian noise)
    t_keys = time.perf_counter() - t0
    msgs = np.arange(4 * cores) % 16
    lut, _ = ks.lut([(x * x) % 16 for x in range(16)])
    cts = ks.encrypt(msgs)
    one = np.zeros(1, dtype=np.uint32)
    ks.ks_pbs_batch(cts[:1], lut, one, threads=1, fast=True)  # warm-up
    # (i) single thread: about 5 s of work
    t0 = time.perf_counter()
    out1 = ks.ks_pbs_batch(cts[:1], lut, one, threads=1, fast=True)
    t_first = time.perf_counter() - t0
    n_single = int(min(64, max(2, round(5.0 / max(t_first, 1e-3)))))
    reps = np.resize(cts, (n_single, p.big_size))
    t0 = time.perf_counter()
    for i in range(n_single):
        ks.ks_pbs_batch(reps[i:i + 1], lut, one, threads=1, fast=True)
    single_ms = (time.perf_counter() - t0) / n_single * 1e3
    assert int(ks.decrypt(out1)[0]) == int((msgs[0] * msgs[0]) % 16), "CPU baseline decrypts wrongly"
    # (ii) all cores: about 10 s of wall time
    warm = np.resize(cts, (2 * cores, p.big_size))
    t0 = time.perf_counter()
    ks.ks_pbs_batch(warm, lut, np.zeros(2 * cores, dtype=np.uint32), threads=cores, fast=True)
    per_round = (time.perf_counter() - t0) / 2
    count = int(min(8192, max(4 * cores, cores * round(10.0 / max(per_round, 1e-3)))))
    batch = np.resize(cts, (count, p.big_size))
    t0 = time.perf_counter()
    got = ks.ks_pbs_batch(batch, lut, np.zeros(count, dtype=np.uint32), threads=cores, fast=True)
    dt = time.perf_counter() - t0
    assert np.array_equal(ks.decrypt(got[:16]).astype(np.int64), (np.resize(msgs, count)[:16] ** 2) % 16)
    return {
        "value": count / dt,
        "unit": "PBS/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{count} KS-PBS (n={params_n}, real keys generated by the oracle in {t_keys:.1f} s, fresh encryptions) in one "
                  f"OpenMP batch over {cores} pinned threads, {dt:.2f} s wall; tuned C port of the reference algorithm "
                  f"(oracle/tfhe_oracle_fast.c: vectorised radix-4 f64 FFT, bit-identical to the oracle; bench build: {O.bench_build_flags()})",
        "cpu_single": {
            "ms_per_pbs": single_ms,
            "sample": f"{n_single} KS-PBS one at a time on 1 thread",
            "reference_published_ms": 16.6,
            "reference_published_source": "tfhe/docs/getting_started/benchmarks.md:42 (m6i.metal, AVX-512, 1 thread)",
        },
        "per_core_pbs_per_s": count / dt / cores,
        "reference_published_per_core_pbs_per_s": 1000.0 / 16.6,
        "reference_derived_same_core_count_pbs_per_s": cores * 1000.0 / 16.6,
    }


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # the first ~4 launches after the key upload run 2-20 % slower (clock ramp, cold L2 / Infinity Cache): the default
    # warm-up covers them, the timed steps are the steady state (profiles/r03_kernel_trace_pair_calls.txt)
    # 100 timed steps = 0.8 s of GPU work (a 20-step run is thin against the utilisation sampler)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=1024, help="ciphertexts per GPU per step")
    ap.add_argument("--lwe-dim", type=int, default=742)
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--real-keys", action="store_true",
                    help="keys and ciphertexts from the oracle's key generation (binary secrets, Gaussian noise, fresh encryptions of "
                         "i mod 16) instead of uniformly random words; the outputs are decrypt-checked after the timed region")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--no-device", action="store_true",
                    help="plumbing rehearsal without a GPU: launch, rendezvous, barriers, gather and reduction run (gloo, "
                         "CPU tensors), the kernels do not; the line carries value null and is NOT a measurement")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher above us: start the N ranks as fresh child
    processes.  Nothing in THIS process has imported torch or touched HIP (a process that has initialised the
    GPU must never exec / fork workers); the children are started with subprocess, never exec'd into."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:               # relay: the JSON line to stdout, anything else to stderr
        s = out.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        elif s:
            print(s, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("bench.py: the ranks exited without printing a result line", file=sys.stderr)
        rc = 1
    sys.exit(rc)


def static_counters():
    """PMC-derived figures of the blind-rotation kernel (separate rocprofv3 --pmc passes; cannot be collected
    inside a timing run).  STATIC: read from profiles/pmc_derived.json, which names the summary it was derived from."""
    path = os.path.join(ROOT, "profiles", "pmc_derived.json")
    try:
        return json.load(open(path))
    except Exception:
        return {}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    # cpu_baseline leg: count the usable cores now (binding the OpenMP threads below narrows this thread's own
    # affinity mask to its place), then pin: the variables are read when libgomp initialises, i.e. before torch is imported
    host_cores = usable_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(host_cores))
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "threads")

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU")
    if args.no_device:
        args.backend = "gloo"
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            local_rank = 0
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(args.backend)
    n_ranks_seen = dist.get_world_size() if dist is not None else 1
    B = args.batch

    if args.no_device:
        # Plumbing rehearsal (CPU only): same control flow, no kernels.  Not a measurement.
        big = 2049
        outs = [torch.zeros((B, big), dtype=torch.int64) for _ in range(2)]
        sk = None
        real = None
        dev = torch.device("cpu")

        def run_kernels(d_out):
            d_out.fill_(rank + 1)
    else:
        import tfhe_rs_string_amd as T

        dev = torch.device(f"cuda:{local_rank}")
        torch.cuda.set_device(dev)
        p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=args.lwe_dim)
        real = None
        if args.real_keys:
            # data generation only (client side: key generation and encryption are out of scope of the product)
            from oracle import pyoracle as O

            real = O.KeySet(O.param_message_2_carry_2(lwe_dim=args.lwe_dim), seed=0x5EED0001)  # same keys on every rank
            bsk, ksk = real.bsk, real.ksk
        else:
            rng = np.random.default_rng(1234)  # same keys on every rank (replicated server key)
            bsk = rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64)
            ksk = rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64)
        ctx = T.Context(local_rank)
        sk = T.ServerKey(ctx, p, bsk, ksk)
        del bsk, ksk
        if args.variant is not None:
            sk.set_variant(args.variant)
        rng = np.random.default_rng(99 + rank)  # every rank its own shard of ciphertexts
        if real is not None:
            real_msgs = (np.arange(B) + rank) % 16
            cts = real.encrypt(real_msgs)
        else:
            cts = rng.integers(0, 2**64, (B, p.big_size), dtype=np.uint64)
        funcs = [lambda x: x % 4, lambda x: x // 4, lambda x: int(x // 4 == x % 4), lambda x: (x // 4) & (x % 4)]
        luts = np.stack([T.generate_lookup_table(p, f)[0] for f in funcs])
        idx = (np.arange(B) % 4).astype(np.int32)
        d_in = torch.from_numpy(cts.view(np.int64)).to(dev)
        d_luts = torch.from_numpy(luts.view(np.int64)).to(dev)
        d_idx = torch.from_numpy(idx).to(dev)
        outs = [torch.empty_like(d_in) for _ in range(2)]

        def run_kernels(d_out):
            sk.ks_pbs_batch_device(d_in, d_luts, d_idx, d_out)

    on_host = world > 1 and args.backend != "nccl" and not args.no_device  # gloo rehearsal: stage results through host memory
    gather = None
    if world > 1:
        from tfhe_rs_string_amd.strings import ResultGather

        gather = ResultGather(dist, outs[0].cpu() if on_host else outs[0])
    step_no = [0]

    def sync():
        if not args.no_device:
            torch.cuda.synchronize()

    def step():
        d_out = outs[step_no[0] & 1]
        step_no[0] += 1
        if gather is not None:
            # the gather that last read this buffer must be done before the kernels rewrite it
            while len(gather.pending) >= 2:
                gather.pending.pop(0).wait()
        run_kernels(d_out)
        if gather is not None:
            # the path's only exchange: gather the result LWEs (16 392 B each) on rank 0 over
            # RCCL/xGMI, overlapped with the next batch's bootstrap (double-buffered outputs)
            gather.submit(d_out.cpu() if on_host else d_out)

    for _ in range(args.warmup):
        step()
    if gather is not None:
        gather.drain()
    sync()
    if sk is not None:
        sk.profiling(True)
        sk.profiling_read(0, reset=True)
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if gather is not None:
        gather.drain()
    sync()
    dt_own = time.perf_counter() - t0     # this rank's own time for its K steps
    if args.no_device and gather is not None and rank == 0:
        for slot in gather.recv:          # the rehearsal checks what it can: every rank's rows arrived in rank order
            for r, got in enumerate(slot):
                assert bool((got == r + 1).all()), f"gather delivered wrong rows for rank {r}"
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    br_ms = br_n = ks_ms = ks_n = 0
    if sk is not None and real is not None:
        # size-independent check on the timed data: decrypt(PBS_f(enc(m))) == f(m) for every ciphertext of this rank
        dec = real.decrypt(outs[(step_no[0] - 1) & 1].cpu().numpy().view(np.uint64)).astype(np.int64)
        want = np.array([funcs[i % 4](int(m)) for i, m in enumerate(real_msgs)])
        assert np.array_equal(dec, want), "real-key outputs decrypt wrongly"
    if sk is not None:
        br_ms, br_n = sk.profiling_read(1)
        ks_ms, ks_n = sk.profiling_read(0)
        sk.profiling(False)
    per_rank = [B * args.steps / dt_own]
    if world > 1:
        cdev = "cpu" if (on_host or args.no_device) else dev
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor([dt_own], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [B * args.steps / float(x.item()) for x in allr]

    if rank == 0:
        pbs_per_s = world * B * args.steps / dt
        br_avg_ms = br_ms / max(br_n, 1)
        ks_avg_ms = ks_ms / max(ks_n, 1)
        n = args.lwe_dim
        a_br = A_BSK * n // 742 + (n + 1) * 8 + A_LUT + A_BIG     # 48 682 816 B at n = 742 (dominant kernel)
        br_s = br_avg_ms * 1e-3
        br_gbps = a_br * B / br_s / 1e9 if br_s > 0 else 0.0
        flop_cmux = schedule_flops_per_cmux()
        flop_pbs = flop_cmux * n
        nominal_pbs = n * (4 * 5 * 1024 * 10 + 4 * 1024 * 6 + 2 * 1024 * 14)  # 5 N log2 N per transform (0.19 GFLOP)
        tflops = flop_pbs * B / br_s / 1e12 if br_s > 0 else 0.0
        pmc = static_counters()
        traffic = pmc.get("hbm_bytes_per_launch")
        out = {
            "metric": "PBS/sec (PARAM_MESSAGE_2_CARRY_2) at 1/2/4/8 GPUs; % HBM roofline",
            "value": None if args.no_device else pbs_per_s,
            "unit": "PBS/s",
            "n_gpus": world,
            "n_ranks_seen": n_ranks_seen,
            "per_rank_pbs_per_s": None if args.no_device else per_rank,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 torus + f64 FFT",
            "data": "real keys and fresh encryptions from the oracle's key generation (outputs decrypt-checked)" if args.real_keys else "synthetic",
            "config": {
                "workload": f"batch of {B} independent shortint KS-PBS per GPU, PARAM_MESSAGE_2_CARRY_2_KS_PBS "
                            f"(n={args.lwe_dim}, k=1, N=2048, pbs 23x1, ks 3x5), 4 LUTs, inputs resident in HBM",
                "batch_per_gpu": B,
                "sharding": "ciphertexts sharded across ranks, keys replicated, async RCCL gather of the output LWEs on rank 0" if world > 1 else "single GPU",
                "launched_by": "torch.distributed.run" if world > 1 else "python",
            },
        }
        if args.no_device:
            out["rehearsal"] = "no device: launch / rendezvous / gather / reduction plumbing only, kernels not run"
        else:
            out["roofline"] = {
                # what the counters say binds the dominant kernel: vector-f64 issue (VALU ~90 % busy), not HBM, not MFMA
                "bound": "valu_f64",
                "kernel": "blind_rotate (+fused sample extract)",
                "achieved": tflops,
                "peak": FP64_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": tflops / FP64_VECTOR_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_source": pmc.get("source", None) and f"static: {pmc.get('source')}",
                "units_per_launch": B,
                "avg_launch_ms": br_avg_ms,
                "keyswitch_avg_launch_ms": ks_avg_ms,
                "compute": {
                    "bound": "valu_f64",
                    "flop_per_unit": flop_pbs,
                    "flop_per_unit_how": f"{flop_cmux} f64 flop per CMUX step x n = {n} steps, counted from FFT schedule v1 "
                                         "(bench.py schedule_flops_per_cmux: butterflies 16 adds, complex product 6, multiply-add 8; fma = 2)",
                    "flop_per_unit_5nlog2n_convention": nominal_pbs,
                    "achieved_tflops": tflops,
                    "achieved_tflops_5nlog2n_convention": nominal_pbs * B / br_s / 1e12 if br_s > 0 else 0.0,
                    "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                    "frac": tflops / FP64_VECTOR_PEAK_TFLOPS,
                    "valu_busy": pmc.get("valu_busy"),
                    "instr_per_step": pmc.get("valu_insts_per_step_wave"),
                    "f64_instr_per_step": pmc.get("f64_insts_per_step_wave_static"),
                    "counters_source": pmc.get("source", None) and f"static: {pmc.get('source')}",
                },
                "hbm_algorithmic": {
                    "convention": "algorithmic, keys cache-resident (SURVEY 8d: every PBS consumes every key byte once; "
                                  "the keys are shared by the batch and served from L2 / Infinity Cache, so frac may exceed 1)",
                    "bytes_per_unit": a_br,
                    "achieved": br_gbps,
                    "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s",
                    "frac": br_gbps / HBM_PEAK_GBPS,
                    "whole_path": {
                        "bytes_per_unit": A_PATH,
                        "achieved": pbs_per_s / world * A_PATH / 1e9,
                        "frac": pbs_per_s / world * A_PATH / 1e9 / HBM_PEAK_GBPS,
                    },
                },
                # the second kernel group of the path: the keyswitch as an exact int8 product on the matrix cores
                # (keyswitch_mfma.hip): 2 x B x (2048 x 5 digit rows) x (n + 1 columns x 8 byte planes) int8 operations per
                # launch; the duration covers its four launches (memset, decomposition, product, finalize)
                "keyswitch": {
                    "bound": "mfma",
                    "kernel": "ks_decompose + keyswitch_mfma + finalize (v_mfma_i32_32x32x32_i8)",
                    "ops_per_launch": 2 * B * 2048 * 5 * (n + 1) * 8,
                    "achieved": 2 * B * 2048 * 5 * (n + 1) * 8 / (ks_avg_ms * 1e-3) / 1e12 if ks_avg_ms > 0 else None,
                    "peak": INT8_MFMA_PEAK_TOPS,
                    "unit": "TOP/s",
                    "frac": 2 * B * 2048 * 5 * (n + 1) * 8 / (ks_avg_ms * 1e-3) / 1e12 / INT8_MFMA_PEAK_TOPS if ks_avg_ms > 0 else None,
                    "avg_launch_ms": ks_avg_ms,
                    "share_of_step": ks_avg_ms / (ks_avg_ms + br_avg_ms) if ks_avg_ms + br_avg_ms > 0 else None,
                },
                "hbm_physical": {
                    "bytes_per_launch": traffic,
                    "bytes_per_launch_source": pmc.get("source", None) and f"static: {pmc.get('source')} (FETCH_SIZE + WRITE_SIZE, gfx950 corrections)",
                    "GBps": traffic / br_s / 1e9 if (traffic and br_s > 0) else None,
                    "frac": traffic / br_s / 1e9 / HBM_PEAK_GBPS if (traffic and br_s > 0) else None,
                    "compulsory_bytes_per_launch": A_BSK * n // 742 + B * ((n + 1) * 8 + A_BIG) + 4 * A_LUT,
                },
            }
        if world == 1 and not args.no_cpu_baseline and not args.no_device:
            out["cpu_baseline"] = cpu_baseline(args.lwe_dim, host_cores)
        elif world > 1:
            out["cpu_baseline"] = {"value": None, "unit": "PBS/s", "cores": None, "kind": "port",
                                   "sample": "timed on rank 0 at N = 1 only (the contract): see the cpu_baseline object of the "
                                             "N = 1 line of the same run"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
