"""world_size-2 `gloo` test of the multi-GPU path's host logic on CPU: `contains_sharded` shards the
independent windows of FheString::contains across ranks, all-gathers the per-window boolean blocks
(the path's only exchange) and finishes with the OR tree; the compute backend here is the test-only
oracle backend (tests/emu/oracle_backend.cpp), on the GPU box every rank drives its own MI355X."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    from tfhe_rs_string_amd import shard_range

    for total in (0, 1, 7, 249, 1024):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, hay, needle, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from tests.test_host_ops_cpu import Env, backend as _b  # noqa: F401
    import ctypes as C
    from oracle import pyoracle as O
    import tfhe_rs_string_amd as T

    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = C.CDLL(os.path.join(ROOT, "tests", "emu", "libhost_emu.so"))
    L.emu_server_key_create.restype = C.c_void_p
    L.emu_server_key_create.argtypes = [C.c_void_p] * 4 + [C.c_int]
    ks = O.KeySet(O.param_message_2_carry_2(lwe_dim=8), seed=77)  # same seed: same keys and same encryptions
    env = Env(O, L, ks)
    eh, en = env.encrypt_str(hay), env.encrypt_str(needle)
    got = T.contains_sharded(env.key, eh, en, dist=dist)
    q.put((rank, env.decrypt_bool(got), env.key.stats()["pbs"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("hay,needle", [("sharded haystack with a needle inside", "needle"), ("no such thing in here", "xyz")])
def test_contains_sharded_world2(oracle, hay, needle):
    import torch.multiprocessing as mp

    # build the emu backend once in the parent
    from tests import test_host_ops_cpu as H

    H.backend.__wrapped__(oracle) if hasattr(H.backend, "__wrapped__") else None
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, hay, needle, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] == int(needle in hay) for r in res)
    # both ranks did PBS work, and the windows were split between them
    assert all(r[2] > 0 for r in res)


def _gather_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import tfhe_rs_string_amd as T

    dist.init_process_group("gloo", rank=rank, world_size=world)
    bufs = [torch.zeros((5, 7), dtype=torch.int64) for _ in range(2)]
    g = T.ResultGather(dist, bufs[0])
    seen = []
    for step in range(5):  # same call pattern as bench.py's step()
        out = bufs[step & 1]
        while len(g.pending) >= 2:
            g.pending.pop(0).wait()
        out.fill_(1000 * step + rank)
        got = g.submit(out)
        if rank == 0:
            seen.append(got)
    g.drain()
    ok = True
    if rank == 0:
        # the last two gathers are still intact in the double-buffered receive lists
        for step in (3, 4):
            ok &= all(int(seen[step][r][0, 0]) == 1000 * step + r for r in range(world))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_result_gather_world2():
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _column_worker(rank, world, port, column, target, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import ctypes as C

    import torch.distributed as dist

    import tfhe_rs_string_amd as T
    from oracle import pyoracle as O
    from tests.test_host_ops_cpu import Env

    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = C.CDLL(os.path.join(ROOT, "tests", "emu", "libhost_emu.so"))
    L.emu_server_key_create.restype = C.c_void_p
    L.emu_server_key_create.argtypes = [C.c_void_p] * 4 + [C.c_int]
    ks = O.KeySet(O.param_message_2_carry_2(lwe_dim=8), seed=78)
    env = Env(O, L, ks)
    enc = [env.encrypt_str(s) for s in column]
    got = T.column_sharded(env.key, lambda k, item: k.contains_clear(item, target), enc, dist=dist)
    q.put((rank, [env.decrypt_bool(b) for b in got], env.key.stats()))
    dist.barrier()
    dist.destroy_process_group()


def test_column_sharded_world2(oracle):
    """a column of strings against one clear pattern, the items sharded over two ranks, each rank's shard in one batch"""
    import torch.multiprocessing as mp

    column = ["alpha beta", "gamma", "delta beta", "beta", "x", "abetab", "no"]
    target = "beta"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_column_worker, args=(r, 2, port, column, target, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [int(target in s) for s in column]
    assert res[0][1] == want and res[1][1] == want
    # the work was split (4 + 3 items) and each shard ran as ONE batch: a handful of launches, not one set per item
    assert res[0][2]["pbs"] > 0 and res[1][2]["pbs"] > 0 and res[0][2]["pbs"] != res[1][2]["pbs"]
    assert res[0][2]["launches"] <= 6 and res[1][2]["launches"] <= 6
