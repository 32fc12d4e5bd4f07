"""bench.py's multi-rank entry: `python bench.py --gpus N` must start its N ranks itself.

CPU tests drive the launch / rendezvous / gather / reduction plumbing with `--no-device` (gloo, CPU tensors, no
kernels: the line says so and carries no value); the GPU test runs the real kernels on two ranks sharing the one
GPU of the box over gloo.  The fan-out mirrors rayon's par_iter over independent blocks
(reference tfhe/src/integer/server_key/radix_parallel/comparison.rs:22-28)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(cmd, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    return r, lines


def test_self_launch_two_ranks_no_device():
    r, lines = _run([sys.executable, BENCH, "--gpus", "2", "--no-device", "--batch", "8", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, lines                       # exactly ONE JSON line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2
    assert out["value"] is None and "rehearsal" in out  # not a measurement
    assert out["config"]["launched_by"] == "torch.distributed.run"


def test_driver_style_launch_two_ranks_no_device():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--no-device", "--batch", "4", "--steps", "2", "--warmup", "1"]
    r, lines = _run(cmd)
    assert r.returncode == 0, r.stderr[-2000:]
    js = [l for l in lines if l.startswith("{")]
    assert len(js) == 1 and json.loads(js[0])["n_gpus"] == 2


def test_self_launch_propagates_failure():
    # a rank that cannot start (WORLD_SIZE / --gpus mismatch is impossible here, so break the backend name instead)
    r, lines = _run([sys.executable, BENCH, "--gpus", "2", "--backend", "no-such-backend", "--batch", "4", "--steps", "1", "--warmup", "0"],
                    env_extra={"HIP_VISIBLE_DEVICES": ""}, timeout=300)
    assert r.returncode != 0
    assert not [l for l in lines if l.startswith("{")]


def test_schedule_flop_count():
    sys.path.insert(0, ROOT)
    import bench

    # 4 transforms of 37 382 flop + 2 x 2 twists of 6 144 + pointwise 28 672
    assert bench.schedule_flops_per_cmux() == 4 * 37382 + 4 * 6144 + 28672


@pytest.mark.gpu
def test_self_launch_two_ranks_on_one_gpu():
    r, lines = _run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--share-gpu", "--lwe-dim", "8", "--batch", "8",
                     "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["value"] > 0
    assert len(out["per_rank_pbs_per_s"]) == 2
    assert out["roofline"]["bound"] == "valu_f64" and "hbm_physical" in out["roofline"] and "compute" in out["roofline"]


# ---- BASELINE.json config 4 as stated: bench_configs.py --gpus N (FheString::contains 256 / 8 sharded over N ranks) -------------
CONFIGS = os.path.join(ROOT, "bench_configs.py")


def test_config4_self_launch_two_ranks_no_device():
    r, lines = _run([sys.executable, CONFIGS, "--gpus", "2", "--no-device", "--reps", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    js = [json.loads(l) for l in lines if l.startswith("{")]
    assert len(js) == 1, lines
    out = js[0]
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["ms"] is None and "rehearsal" in out
    assert out["windows"] == 249 and out["windows_per_rank"] == [[0, 125], [125, 249]]   # 256 - 8 + 1 windows, ceil split
    assert len(out["gather_ms_per_rank"]) == 2


def test_config4_driver_style_launch_three_ranks_no_device():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", str(port), CONFIGS, "--gpus", "3", "--no-device", "--reps", "1"]
    r, lines = _run(cmd)
    assert r.returncode == 0, r.stderr[-2000:]
    js = [json.loads(l) for l in lines if l.startswith("{")]
    assert len(js) == 1 and js[0]["windows_per_rank"] == [[0, 83], [83, 166], [166, 249]]


@pytest.mark.gpu
def test_config4_two_ranks_and_two_replicas_on_one_gpu():
    """real kernels: two ranks sharing GPU 0 over gloo (windows sharded, all_gather, OR tree) and, on rank 0, the same op over a
    two-device C-ABI context with GPU 0 listed twice; distinct GPUs stay unmeasured until the driver's 8-GPU node"""
    r, lines = _run([sys.executable, CONFIGS, "--gpus", "2", "--backend", "gloo", "--share-gpu", "--reps", "2"], timeout=1100)
    assert r.returncode == 0, r.stderr[-3000:]
    js = [json.loads(l) for l in lines if l.startswith("{")]
    assert len(js) == 2, lines
    sharded, ctx = js
    assert sharded["n_ranks_seen"] == 2 and sharded["ms"] > 0 and sharded["distinct_gpus"] is False
    assert sharded["windows_per_rank"] == [[0, 125], [125, 249]] and len(sharded["windows_ms_per_rank"]) == 2
    assert "error" not in ctx and ctx["device_ids"] == [0, 0] and ctx["ms"] > 0 and ctx["pbs"] > 4000
