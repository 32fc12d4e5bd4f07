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


@pytest.mark.parametrize("variant", [3, 2, 1, 0])
def test_device_entry_point_flags_lut_idx_out_of_range(T, env, keyset_small, variant):
    import torch

    ctx, p, sk, lut = env
    ks = keyset_small
    sk.set_variant(variant)
    cts = ks.encrypt(np.arange(5) % 16)
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(cts.view(np.int64)).to(dev)
    d_luts = torch.from_numpy(lut.view(np.int64).reshape(1, -1)).to(dev)
    d_out = torch.empty_like(d_in)
    good = torch.zeros(5, dtype=torch.int32, device=dev)
    sk.ks_pbs_batch_device(d_in, d_luts, good, d_out)
    torch.cuda.synchronize()
    assert sk.device_status() == 0
    want = ks.ks_pbs_batch(cts, lut, np.zeros(5, dtype=np.uint32), threads=4)
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want)
    bad = torch.tensor([0, 0, 7, 0, 0], dtype=torch.int32, device=dev)   # one table, index 7
    sk.ks_pbs_batch_device(d_in, d_luts, bad, d_out)
    torch.cuda.synchronize()
    assert sk.device_status() & 2
    with pytest.raises(T.TfheHipError, match="lut_idx out of range"):
        sk.ks_pbs_batch_device(d_in, d_luts, good, d_out)                  # sticky
    assert sk.device_status(clear=True) & 2
    sk.ks_pbs_batch_device(d_in, d_luts, good, d_out)
    torch.cuda.synchronize()
    assert sk.device_status() == 0 and np.array_equal(d_out.cpu().numpy().view(np.uint64), want)
    sk.set_variant(-1)


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
