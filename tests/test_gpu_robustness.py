"""-m gpu: the failure paths and the leveled kernel of the HIP engine.

  * a wave pair of the throughput kernel that never hears from its partner must end in rc 1, not in wrong ciphertexts
    with rc 0: provoked with the diagnostics hook tfhe_hip_debug_skip_pair_signal;
  * a lut_idx past the tables in a device-buffer call (which the host cannot inspect) must not read out of bounds;
  * tfhe_hip_device_lincomb_rows (lwe_ciphertext_add_assign / lwe_ciphertext_cleartext_mul_assign,
    core_crypto/algorithms/lwe_linear_algebra.rs:68, 556) against the oracle's orc_lwe_add_assign / orc_lwe_scalar_mul_assign.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import tfhe_rs_string_amd as T

    return T


@pytest.fixture(scope="module")
def env(T, keyset_small):
    ctx = T.Context(0)
    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=keyset_small.p.lwe_dim)
    sk = T.ServerKey(ctx, p, keyset_small.bsk, keyset_small.ksk)
    lut, _ = T.generate_lookup_table(p, lambda x: (x * x) % 16)
    return ctx, p, sk, lut


def test_pair_handoff_timeout_is_an_error_not_a_wrong_result(T, env, keyset_small):
    ctx, p, sk, lut = env
    ks = keyset_small
    cts = ks.encrypt(np.arange(8) % 16)
    want = ks.ks_pbs_batch(cts, lut, np.zeros(8, dtype=np.uint32), threads=4)
    sk.set_variant(2)
    assert np.array_equal(sk.ks_pbs_batch(cts, lut, 0), want)
    assert sk.device_status() == 0
    sk.debug_skip_pair_signal(3)            # one wave of ciphertext 0 goes silent from CMUX step 3 on
    with pytest.raises(T.TfheHipError, match="timed out"):
        sk.ks_pbs_batch(cts, lut, 0)
    assert sk.device_status() & 1
    sk.debug_skip_pair_signal(-1)
    with pytest.raises(T.TfheHipError, match="device error word"):   # sticky until read and cleared
        sk.ks_pbs_batch(cts, lut, 0)
    assert sk.device_status(clear=True) & 1
    assert sk.device_status() == 0
    assert np.array_equal(sk.ks_pbs_batch(cts, lut, 0), want)        # the key works again, bit-exact
    sk.set_variant(-1)


def _cus():
    import torch

    return torch.cuda.get_device_properties(0).multi_processor_count


# (variant, batch size as a function of the CU count): every kernel on a small batch, and the automatic dispatch (-1) in
# each of its switch-over bands (block kernel alone; quad kernel; quad + block; wave-pair head + tail on the others)
LUT_IDX_CASES = [(5, lambda c: 5), (4, lambda c: 5), (3, lambda c: 5), (2, lambda c: 5), (1, lambda c: 5), (0, lambda c: 5),
                 (-1, lambda c: 5), (-1, lambda c: c + 1), (-1, lambda c: 2 * c + 1), (-1, lambda c: 4 * c + 37)]


@pytest.mark.parametrize("variant,size", LUT_IDX_CASES, ids=[f"v{v}_{i}" for i, (v, _) in enumerate(LUT_IDX_CASES)])
def test_device_entry_point_flags_lut_idx_out_of_range(T, env, keyset_small, variant, size):
    import torch

    ctx, p, sk, lut = env
    ks = keyset_small
    sk.set_variant(variant)
    B = size(_cus())
    rng = np.random.default_rng(B)
    cts = ks.encrypt(rng.integers(0, 16, B))
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(cts.view(np.int64)).to(dev)
    d_luts = torch.from_numpy(lut.view(np.int64).reshape(1, -1)).to(dev)
    d_out = torch.empty_like(d_in)
    good = torch.zeros(B, dtype=torch.int32, device=dev)
    sk.ks_pbs_batch_device(d_in, d_luts, good, d_out)
    torch.cuda.synchronize()
    assert sk.device_status() == 0
    want = ks.ks_pbs_batch(cts, lut, np.zeros(B, dtype=np.uint32), threads=8)
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want)
    for where in sorted({2, B // 2, B - 1}):                                # one table, index 7: in every part of the batch
        bad_idx = np.zeros(B, dtype=np.int32)
        bad_idx[where] = 7
        bad = torch.from_numpy(bad_idx).to(dev)
        sk.ks_pbs_batch_device(d_in, d_luts, bad, d_out)
        torch.cuda.synchronize()
        assert sk.device_status() & 2, where
        with pytest.raises(T.TfheHipError, match="lut_idx out of range"):
            sk.ks_pbs_batch_device(d_in, d_luts, good, d_out)              # sticky
        assert sk.device_status(clear=True) & 2
    sk.ks_pbs_batch_device(d_in, d_luts, good, d_out)
    torch.cuda.synchronize()
    assert sk.device_status() == 0 and np.array_equal(d_out.cpu().numpy().view(np.uint64), want)
    sk.set_variant(-1)


def test_multi_bit_kernel_flags_lut_idx_out_of_range(T, oracle):
    """the same guard in blind_rotate_block_mb_kernel (multi-bit keys take only that kernel)"""
    import torch

    g, n = 2, 12
    ks = oracle.MultiBitKeySet(oracle.param_multi_bit_message_2_carry_2_group_2(lwe_dim=n), g, seed=23)
    p = T.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS(lwe_dim=n)
    sk = T.ServerKey(T.Context(0), p, ks.bsk, ks.ksk, grouping_factor=g)
    lut, _ = T.generate_lookup_table(p, lambda x: (x + 1) % 16)
    B = 9
    cts = ks.encrypt(np.arange(B) % 16)
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(cts.view(np.int64)).to(dev)
    d_luts = torch.from_numpy(lut.view(np.int64).reshape(1, -1)).to(dev)
    d_out = torch.empty_like(d_in)
    good = torch.zeros(B, dtype=torch.int32, device=dev)
    bad = torch.tensor([0, 0, 0, 3, 0, 0, 0, 0, 0], dtype=torch.int32, device=dev)
    sk.ks_pbs_batch_device(d_in, d_luts, bad, d_out)
    torch.cuda.synchronize()
    assert sk.device_status(clear=True) & 2
    sk.ks_pbs_batch_device(d_in, d_luts, good, d_out)
    torch.cuda.synchronize()
    assert sk.device_status() == 0
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), ks.ks_pbs_batch(cts, lut, np.zeros(B, dtype=np.uint32), threads=4))


@pytest.mark.parametrize("size", [lambda c: 4 * c + 9])
def test_pair_timeout_in_the_head_of_an_automatic_split(T, env, keyset_small, size):
    """automatic dispatch, wave-pair head + tail on the latency kernels: a silent wave in the head is still rc 1"""
    ctx, p, sk, lut = env
    ks = keyset_small
    B = size(_cus())
    cts = ks.encrypt(np.arange(B) % 16)
    sk.set_variant(-1)
    sk.debug_skip_pair_signal(2)
    with pytest.raises(T.TfheHipError, match="timed out"):
        sk.ks_pbs_batch(cts, lut, 0)
    sk.debug_skip_pair_signal(-1)
    assert sk.device_status(clear=True) & 1
    want = ks.ks_pbs_batch(cts, lut, np.zeros(B, dtype=np.uint32), threads=16)
    assert np.array_equal(sk.ks_pbs_batch(cts, lut, 0), want)


def test_host_entry_point_rejects_lut_idx_out_of_range(T, env, keyset_small):
    ctx, p, sk, lut = env
    cts = keyset_small.encrypt(np.arange(3) % 16)
    with pytest.raises(T.TfheHipError, match="out of range"):
        sk.ks_pbs_batch(cts, lut, np.array([0, 1, 0], dtype=np.uint32))
    assert sk.device_status() == 0


@pytest.mark.parametrize("rows,max_terms", [(1, 1), (7, 3), (300, 16), (64, 0)])
def test_device_lincomb_rows_matches_oracle_linear_algebra(T, env, oracle, rows, max_terms):
    """d_dst[r] = sum_t coef[t] * src[t] + (0, .., 0, body[r])  ==  the reference's leveled operations applied one by one"""
    import torch

    from tfhe_rs_string_amd import _native

    ctx, p, sk, lut = env
    words = p.big_size
    rng = np.random.default_rng(rows * 131 + max_terms)
    n_src = 40
    src_rows = rng.integers(0, 2**64, (n_src, words), dtype=np.uint64)
    dev = torch.device("cuda:0")
    d_src = torch.from_numpy(src_rows.view(np.int64)).to(dev)
    d_dst = torch.full((rows, words), -1, dtype=torch.int64, device=dev)
    counts = rng.integers(0, max_terms + 1, rows) if max_terms else np.zeros(rows, dtype=np.int64)
    row_ptr = np.zeros(rows + 1, dtype=np.uint32)
    row_ptr[1:] = np.cumsum(counts)
    terms = int(row_ptr[-1])
    which = rng.integers(0, n_src, terms)
    # coefficients as the op-graphs use them (small scalars, negations = 2^64 - c) plus arbitrary 64-bit ones
    coef = np.where(rng.random(terms) < 0.5, rng.integers(1, 16, terms, dtype=np.uint64), rng.integers(0, 2**64, terms, dtype=np.uint64))
    coef = np.ascontiguousarray(coef, dtype=np.uint64)
    src = np.ascontiguousarray(d_src.data_ptr() + which.astype(np.uint64) * np.uint64(words * 8), dtype=np.uint64)
    body = rng.integers(0, 2**64, rows, dtype=np.uint64)
    vp = C.c_void_p
    _native.check(_native.lib().tfhe_hip_device_lincomb_rows(sk._h, d_dst.data_ptr(), rows, words, row_ptr.ctypes.data_as(vp),
                                                            coef.ctypes.data_as(vp), src.ctypes.data_as(vp), body.ctypes.data_as(vp)))
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy().view(np.uint64)
    L = oracle.lib()
    for r in range(rows):
        acc = np.zeros(words, dtype=np.uint64)
        for t in range(row_ptr[r], row_ptr[r + 1]):
            tmp = src_rows[which[t]].copy()
            L.orc_lwe_scalar_mul_assign(tmp, int(coef[t]), words)       # lwe_ciphertext_cleartext_mul_assign
            L.orc_lwe_add_assign(acc, tmp, words)                       # lwe_ciphertext_add_assign
        acc[-1] += body[r]                                              # lwe_ciphertext_plaintext_add_assign
        assert np.array_equal(got[r], acc), f"row {r}"


def test_per_stream_workspaces_are_bounded_and_can_be_released(T, env, keyset_small):
    """a host that creates a stream per request must not leave a workspace behind per stream: at most
    TFHE_HIP_MAX_STREAM_WS (8) caller streams keep one, the least recently used goes first, and
    tfhe_hip_server_key_release_stream hands one back explicitly; results stay bit-exact throughout"""
    import torch

    ctx, p, sk, lut = env
    ks = keyset_small
    cts = ks.encrypt(np.arange(6) % 16)
    want = ks.ks_pbs_batch(cts, lut, np.zeros(6, dtype=np.uint32), threads=4)
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(cts.view(np.int64)).to(dev)
    d_luts = torch.from_numpy(lut.view(np.int64).reshape(1, -1)).to(dev)
    d_idx = torch.zeros(6, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    base = sk.release_stream(0xDEAD0)  # an unknown stream: nothing released, just the count
    streams = [torch.cuda.Stream(device=dev) for _ in range(12)]
    outs = []
    for st in streams:
        d_out = torch.empty_like(d_in)
        sk.ks_pbs_batch_device(d_in, d_luts, d_idx, d_out, stream=st.cuda_stream)
        outs.append(d_out)
    torch.cuda.synchronize()
    for d_out in outs:
        assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want)
    held = sk.release_stream(0xDEAD0)
    assert held <= base + 8, (base, held)
    assert sk.release_stream(streams[-1].cuda_stream) == held - 1   # the most recent one was still there
    assert sk.release_stream(streams[0].cuda_stream) == held - 1    # the oldest had been evicted already
    d_out = torch.empty_like(d_in)
    sk.ks_pbs_batch_device(d_in, d_luts, d_idx, d_out, stream=streams[-1].cuda_stream)  # a released stream simply gets a new one
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want)
