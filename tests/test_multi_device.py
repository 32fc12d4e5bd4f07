"""Multi-device contexts: tfhe_hip_context_create(out, device_ids, n_devices); a key on such
a context is replicated and every batch entry point splits its ciphertexts contiguously over the devices (shard.hpp),
which is the fan-out rayon gives the reference (tfhe/src/integer/server_key/radix_parallel/comparison.rs:22-28).

CPU: the test backend exposes fake devices that shard with the SAME function as api.cpp: the host op-graph layer
(contains, mul, a column of eq in batch mode) must give identical bits on 1 and 2 devices.
-m gpu: the real sharding code of api.cpp on the one-GPU box with the SAME device listed twice (two replicas, two
streams, copies between them): host-buffer and device-buffer entry points and a string op, bit-exact against the
single-device key and the oracle.  Distinct GPUs: unmeasured (no multi-GPU box in reach of the tests)."""
import ctypes as C

import numpy as np
import pytest

from tests.test_host_ops_cpu import Env, backend  # noqa: F401


def test_shard_ranges_cover_the_batch():
    import tfhe_rs_string_amd as T

    for total in (0, 1, 7, 64, 1000, 1025):
        for world in (1, 2, 3, 8):
            cuts = [T.shard_range(total, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in cuts]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_op_graphs_give_identical_bits_on_one_and_two_fake_devices(oracle, backend, keyset_small):  # noqa: F811
    import tfhe_rs_string_amd as T

    backend.emu_server_key_set_devices.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    backend.emu_server_key_last_shards.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    backend.emu_server_key_last_shards.restype = C.c_size_t
    results = []
    for n_dev in (1, 2):
        env = Env(oracle, backend, keyset_small)
        backend.emu_server_key_set_devices(env.sk, n_dev, 4)
        key = env.key
        rng = np.random.default_rng(11)                       # same inputs for both runs: the oracle's RNG is per keyset
        env.ks.rng = type(env.ks.rng)()
        oracle.lib().orc_rng_seed(C.byref(env.ks.rng), 1234)
        hay, needle = env.encrypt_str("the quick brown fox jumps"), env.encrypt_str("fox")
        a, b = env.encrypt_radix(0xBEEF, 8), env.encrypt_radix(0x1234, 8)
        out = [key.contains(hay, needle), key.mul_parallelized(a, b)]
        shards = np.zeros(4, dtype=np.uint64)
        n = backend.emu_server_key_last_shards(env.sk, shards.ctypes.data_as(C.c_void_p), 4)
        with key.batch():
            col = [key.eq(env.encrypt_str(w), env.encrypt_str("fox")) for w in ("fox", "fax", "fox", "f", "foxy")]
        out += col
        results.append(out)
        assert env.decrypt_bool(out[0]) == 1 and env.decrypt_radix(out[1]) == (0xBEEF * 0x1234) & 0xFFFF
        assert [env.decrypt_bool(c) for c in col] == [1, 0, 1, 0, 0]
        assert n <= n_dev
    for x, y in zip(*results):
        assert np.array_equal(x, y)


@pytest.mark.gpu
def test_same_gpu_listed_twice_shards_bit_exactly(keyset_small):
    import torch

    import tfhe_rs_string_amd as T

    ks = keyset_small
    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=ks.p.lwe_dim)
    one = T.ServerKey(T.Context(0), p, ks.bsk, ks.ksk)
    ctx2 = T.Context([0, 0])
    assert ctx2.device_count() == 2
    two = T.ServerKey(ctx2, p, ks.bsk, ks.ksk)
    funcs = [lambda x: x % 4, lambda x: x // 4, lambda x: int(x // 4 == x % 4), lambda x: (x * x) % 16]
    luts = np.stack([T.generate_lookup_table(p, f)[0] for f in funcs])
    for B in (1, 63, 128, 129, 700):
        cts = ks.encrypt(np.arange(B) % 16)
        idx = (np.arange(B) % 4).astype(np.uint32)
        want = one.ks_pbs_batch(cts, luts, idx)
        assert np.array_equal(two.ks_pbs_batch(cts, luts, idx), want), B            # host buffers: one thread per device
        assert np.array_equal(two.keyswitch_batch(cts), one.keyswitch_batch(cts)), B
        dev = torch.device("cuda:0")                                                 # device buffers: copies + events
        d_in = torch.from_numpy(cts.view(np.int64)).to(dev)
        d_luts = torch.from_numpy(luts.view(np.int64)).to(dev)
        d_idx = torch.from_numpy(idx.astype(np.int32)).to(dev)
        d_out = torch.zeros_like(d_in)
        two.ks_pbs_batch_device(d_in, d_luts, d_idx, d_out)
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want), B
    assert np.array_equal(want, ks.ks_pbs_batch(cts, luts, idx, threads=8))          # ... and all of it equals the oracle
    assert two.device_status() == 0
    # the op-graph layer on the two-device key: every level is sharded underneath
    k1, k2 = T.StringServerKey(one.handle, p), T.StringServerKey(two.handle, p)
    enc = lambda s: ks.encrypt([(ord(ch) >> (2 * i)) & 3 for ch in s for i in range(4)]).reshape(len(s), 4, -1)  # noqa: E731
    hay, needle = enc("the quick brown fox jumps over the lazy dog " * 3), enc("lazy")
    r1, r2 = k1.contains(hay, needle), k2.contains(hay, needle)
    assert np.array_equal(r1, r2) and int(ks.decrypt(r1.reshape(1, -1))[0]) == 1
    # a wave-pair time-out on a replica is reported through the primary key
    two.set_variant(2)
    two.debug_skip_pair_signal(2)
    cts = ks.encrypt(np.arange(256) % 16)
    with pytest.raises(T.TfheHipError, match="timed out"):
        two.ks_pbs_batch(cts, luts[0], 0)
    two.debug_skip_pair_signal(-1)
    assert two.device_status(clear=True) & 1
    assert np.array_equal(two.ks_pbs_batch(cts, luts[0], 0), one.ks_pbs_batch(cts, luts[0], 0))
