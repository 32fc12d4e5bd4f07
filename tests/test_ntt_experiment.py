"""the exact-integer external product experiment (oracle/tfhe_oracle_ntt.c).

The reference multiplies polynomials through an f64 FFT (fft64/crypto/ggsw.rs:477-598); here the same
add_external_product is computed EXACTLY (negacyclic NTT modulo 2^64 - 2^32 + 1 on 22-bit limbs of the key):
  * it equals the schoolbook convolution of the reference's integer code (polynomial_algorithms.rs:130-199) bit for bit: T0;
  * the f64 path's distance to it is the noise of the transform, measured here (std and max, in units of the torus);
  * a bootstrap built on exact products decrypts like the f64 one.
The throughput half of the experiment is tools/ntt_experiment.py (CPU) / DESIGN.md section 8."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ntt(oracle):
    h = oracle.lib().orc_ntt_new(2048)
    assert h
    yield h
    oracle.lib().orc_ntt_free(h)


def centered(x):
    return x.astype(np.int64).astype(np.float64) / 2.0**64


def test_exact_external_product_equals_schoolbook_and_bounds_the_fft_noise(oracle, ntt, keyset_small):
    L = oracle.lib()
    ks = keyset_small
    p = ks.p
    N = p.poly_size
    rng = np.random.default_rng(5)
    ggsw_len = 4 * N
    for i in (0, 7):                                          # two of the key's GGSW ciphertexts
        ggsw = np.ascontiguousarray(ks.bsk[i * ggsw_len:(i + 1) * ggsw_len])
        prepared = np.zeros(L.orc_ntt_ggsw_len(C.byref(p)), dtype=np.uint64)
        L.orc_ntt_ggsw_prepare(C.byref(p), ntt, ggsw, prepared)
        glwe_in = rng.integers(0, 2**64, 2 * N, dtype=np.uint64)
        start = rng.integers(0, 2**64, 2 * N, dtype=np.uint64)
        exact, school, f64 = start.copy(), start.copy(), start.copy()
        L.orc_ntt_add_external_product(C.byref(p), ntt, exact, prepared, glwe_in)
        L.orc_schoolbook_add_external_product(C.byref(p), school, ggsw, glwe_in)
        assert np.array_equal(exact, school)                  # T0: bit-exact, no tolerance
        again = start.copy()
        L.orc_ntt_add_external_product(C.byref(p), ntt, again, prepared, glwe_in)
        assert np.array_equal(again, exact)                   # deterministic
        n2 = N // 2
        re, im = ks.bsk_re[i * 4 * n2:(i + 1) * 4 * n2], ks.bsk_im[i * 4 * n2:(i + 1) * 4 * n2]
        L.orc_add_external_product(C.byref(p), ks.fft.h, f64, np.ascontiguousarray(re), np.ascontiguousarray(im), glwe_in)
        err = centered(f64 - exact)
        # one external product at 2_2 (2^22 digits x 64-bit key coefficients, 4096-term sums, 53-bit mantissa): the f64
        # transform is off by sigma ~ 2^-26 of the torus, max ~ 2^-24: 3 orders of magnitude below the bootstrap's own
        # noise per step (2.5e-5 / sqrt(742) ~ 2^-20), which is why the reference can afford it
        print(f"\n[f64 transform vs exact] sigma = {err.std():.3e} = 2^{np.log2(err.std()):.1f}, max = {np.abs(err).max():.3e} of the torus")
        assert 0 < np.abs(err).max() < 2.0**-21, np.abs(err).max()
        assert 2.0**-29 < err.std() < 2.0**-24, err.std()


def test_bootstrap_on_exact_products_decrypts(oracle, ntt, keyset_small):
    L = oracle.lib()
    ks = keyset_small
    p = ks.p
    bsk_ntt = np.zeros(p.lwe_dim * L.orc_ntt_ggsw_len(C.byref(p)), dtype=np.uint64)
    L.orc_ntt_bsk_prepare(C.byref(p), ntt, ks.bsk, bsk_ntt)
    msgs = np.arange(16)
    small = ks.keyswitch_batch(ks.encrypt(msgs))
    lut, _ = ks.lut([(3 * x + 1) % 16 for x in range(16)])
    idx = np.zeros(16, dtype=np.uint32)
    out = np.zeros((16, p.big_size), dtype=np.uint64)
    L.orc_ntt_pbs_batch(C.byref(p), ntt, bsk_ntt, small, lut, idx, out, 16, 4)
    assert np.array_equal(ks.decrypt(out).astype(np.int64), (3 * msgs + 1) % 16)
    # NOT comparable ciphertext by ciphertext with the f64 path: one digit that rounds the other way (the f64 noise decides
    # ties differently) changes the accumulator by a whole key coefficient -- both are encryptions of the same message
    # with the same noise distribution, which is all the reference's own tests ask for
    f64 = ks.pbs_batch(small, lut, idx, threads=4)
    delta = np.uint64((1 << 63) // 16)
    want = (((3 * msgs + 1) % 16).astype(np.uint64)) * delta
    e_exact, e_f64 = centered(ks.decrypt_raw(out) - want), centered(ks.decrypt_raw(f64) - want)
    assert np.abs(e_exact).max() < 2.0**-8 and np.abs(e_f64).max() < 2.0**-8


@pytest.mark.gpu
def test_gpu_exact_external_product_is_the_schoolbook_product(oracle, ntt, keyset_small):
    """the HIP kernel of the experiment (csrc/exact_ntt.hip) against the oracle's transform AND the schoolbook product: T0"""
    import tfhe_rs_string_amd as T

    L = oracle.lib()
    ks = keyset_small
    p = ks.p
    N = p.poly_size
    rng = np.random.default_rng(9)
    ggsw = np.ascontiguousarray(ks.bsk[3 * 4 * N:4 * 4 * N])
    B = 5
    glwe_in = rng.integers(0, 2**64, (B, 2 * N), dtype=np.uint64)
    glwe_in[2] = 0
    start = rng.integers(0, 2**64, (B, 2 * N), dtype=np.uint64)
    got, _ = T.exact_external_product_batch(ggsw, p.pbs_base_log, glwe_in, start)
    prepared = np.zeros(L.orc_ntt_ggsw_len(C.byref(p)), dtype=np.uint64)
    L.orc_ntt_ggsw_prepare(C.byref(p), ntt, ggsw, prepared)
    for b in range(B):
        want, school = start[b].copy(), start[b].copy()
        L.orc_ntt_add_external_product(C.byref(p), ntt, want, prepared, np.ascontiguousarray(glwe_in[b]))
        assert np.array_equal(got[b], want), b
        if b < 2:
            L.orc_schoolbook_add_external_product(C.byref(p), school, ggsw, np.ascontiguousarray(glwe_in[b]))
            assert np.array_equal(got[b], school), b
