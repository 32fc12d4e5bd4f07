"""Pins the CPU oracle's integer (T0) functions against the known answers the reference holds for
this path: its doc-tests and unit-test vectors.  Each case cites the reference file:line.
(All reference paths relative to /root/reference/tfhe/src.)  CPU only.
"""
import ctypes as C

import numpy as np


def u64(a):
    return np.array(a, dtype=np.uint64)


def as_u8(a):
    return (np.asarray(a, dtype=np.uint64) & np.uint64(0xFF)).tolist()


def test_closest_representable_doc(oracle):
    # core_crypto/commons/math/decomposition/decomposer.rs:94-95 (u32, base_log 4, 3 levels):
    # closest_representable(1_340_987_234) == 1_341_128_704.  A u32 torus value is the top half of
    # the u64 torus value, so the u64 routine must give the same answer in its top 32 bits.
    L = oracle.lib()
    got = L.orc_closest_representable(1_340_987_234 << 32, 4, 3)
    assert got == 1_341_128_704 << 32


def test_decompose_doc_properties(oracle):
    # decomposer.rs:131-143: 3 terms, each in [-8, 8); decomposer.rs:164-169: recompose(decompose(v))
    # == closest_representable(v) with summand = term << (BITS - base_log*level)
    # (decomposition/term.rs to_recomposition_summand).
    L = oracle.lib()
    rng = np.random.default_rng(7)
    vals = [1_340_987_234 << 32, 1, 0, 2**64 - 1, 2**63] + [int(x) for x in rng.integers(0, 2**64, 200, dtype=np.uint64)]
    for base_log, level in [(4, 3), (3, 5), (23, 1), (15, 2)]:
        for v in vals:
            terms = np.zeros(level, dtype=np.int64)
            L.orc_decompose(v, base_log, level, terms)
            half = 1 << (base_log - 1)
            assert all(-half <= int(t) <= half for t in terms)
            rec = 0
            for i, t in enumerate(terms):  # terms[0] is level `level`
                lvl = level - i
                rec = (rec + (int(t) << (64 - base_log * lvl))) % 2**64
            assert rec == L.orc_closest_representable(v, base_log, level)


def test_monomial_doc(oracle):
    L = oracle.lib()
    # polynomial_algorithms.rs:310-313: [1,2,3] / X^2 == [3,255,254] (u8)
    out = np.zeros(3, dtype=np.uint64)
    L.orc_monomial_div(out, u64([1, 2, 3]), 2, 3)
    assert as_u8(out) == [3, 255, 254]
    # polynomial_algorithms.rs:370-373: [1,2,3] * X^2 == [254,253,1]
    L.orc_monomial_mul(out, u64([1, 2, 3]), 2, 3)
    assert as_u8(out) == [254, 253, 1]
    # mul_and_subtract (:425-490) has no doc-test: check it is mul minus input, all degrees incl. 2N
    rng = np.random.default_rng(3)
    N = 16
    a = rng.integers(0, 2**64, N, dtype=np.uint64)
    for deg in range(0, 2 * N + 1):
        m = np.zeros(N, dtype=np.uint64)
        ms = np.zeros(N, dtype=np.uint64)
        L.orc_monomial_mul(m, a, deg, N)
        L.orc_monomial_mul_and_subtract(ms, a, deg, N)
        assert np.array_equal(ms, m - a)
        d = np.zeros(N, dtype=np.uint64)
        L.orc_monomial_div(d, m, deg, N)
        assert np.array_equal(d, a)


def test_slice_and_poly_doc(oracle):
    L = oracle.lib()
    # slice_algorithms.rs:356-362: first -= second * 4  ->  [5,6,7,0,253,250] (u8)
    first = u64([1, 2, 3, 4, 5, 6])
    L.orc_slice_sub_scalar_mul(first, u64([255, 255, 255, 1, 2, 3]), 4, 6)
    assert as_u8(first) == [5, 6, 7, 0, 253, 250]
    # polynomial_algorithms.rs:84-91: 250 + [100,20,3]*[0,1,1] + [4,5,6]*[1,0,0] == [231,96,120] (u8)
    out = u64([250, 250, 250])
    L.orc_poly_add_mul(out, u64([100, 20, 3]), u64([0, 1, 1]), 3)
    L.orc_poly_add_mul(out, u64([4, 5, 6]), u64([1, 0, 0]), 3)
    assert as_u8(out) == [231, 96, 120]


def test_f64_to_i64_bit_twiddle_vectors(oracle):
    # fft64/math/fft/tests.rs:243-299: the listed inputs must convert like `x as i64`
    L = oracle.lib()
    xs = [0.0, -0.0, 37.1242161, -37.1242161, 0.1, -0.1, 1.0, -1.0, 0.9, -0.9, 2.0, -2.0, 1e-310, -1e-310,
          2.0**62, -(2.0**62), 1.1 * 2.0**62, 1.1 * -(2.0**62), -(2.0**63)]
    for x in xs:
        assert L.orc_f64_to_u64_wrap(x) == int(x) % 2**64


def test_modswitch_range(oracle):
    # fft_impl/common.rs:20-43: result in [0, 2N], 2N reachable
    L = oracle.lib()
    assert L.orc_modswitch(2**64 - 1, 11) == 4096
    assert L.orc_modswitch(0, 11) == 0
    assert L.orc_modswitch(1 << 52, 11) == 1
    assert L.orc_modswitch((1 << 51), 11) == 1 and L.orc_modswitch((1 << 51) - 1, 11) == 0


def test_lut_layout(oracle):
    # shortint/engine/mod.rs:94-127: 16 boxes of 128, first half-box negated, rotate-left 64
    p = oracle.param_message_2_carry_2()
    table = u64([(3 * i + 1) % 16 for i in range(16)])
    out = np.zeros(p.lut_len, dtype=np.uint64)
    deg = oracle.lib().orc_lut_generate(C.byref(p), table, out)
    assert deg == int(table.max())
    delta = 1 << 59
    body = out[2048:]
    assert not out[:2048].any()
    for i in range(16):
        lo, hi = i * 128 - 64, i * 128 + 64
        seg = body[max(lo, 0):hi]
        assert (seg == np.uint64(int(table[i]) * delta % 2**64)).all()
    assert (body[-64:] == np.uint64((-int(table[0]) * delta) % 2**64)).all()
    # trivial_pbs (shortint/server_key/mod.rs:763-781; test shortint.rs:216 `shortint_trivial_pbs`)
    for m in range(32):
        got = oracle.lib().orc_trivial_pbs(C.byref(p), out, (m * delta) % 2**64)
        want = int(table[m % 16]) * delta
        if m >= 16:
            want = -want
        assert got == want % 2**64


def test_sample_extract(oracle):
    # glwe_sample_extraction.rs:125-146: lwe[0]=a[0], lwe[j]=-a[N-j]; body = b[0].
    # Property: <lwe_mask, s> == (a * S)[0] for the flattened key.
    L = oracle.lib()
    rng = np.random.default_rng(5)
    N = 32
    glwe = rng.integers(0, 2**64, 2 * N, dtype=np.uint64)
    s = rng.integers(0, 2, N, dtype=np.uint64)
    lwe = np.zeros(N + 1, dtype=np.uint64)
    L.orc_sample_extract(glwe, 1, N, lwe)
    prod = np.zeros(N, dtype=np.uint64)
    L.orc_poly_add_mul(prod, glwe[:N].copy(), s, N)
    assert int((lwe[:N] * s).sum(dtype=np.uint64)) == int(prod[0])
    assert lwe[N] == glwe[N]


def test_fft_roundtrip_and_product(oracle):
    # fft64/math/fft/tests.rs:9-80 (roundtrip) and :82-200 (product vs schoolbook): the reference
    # accepts |error| < 2^-(BITS - 34)... on the torus; here: roundtrip within 2^13 of 2^64,
    # product with a 23-bit digit polynomial within 2^45.
    L = oracle.lib()
    f = oracle.Fft(2048)
    rng = np.random.default_rng(11)
    b = rng.integers(0, 2**64, 2048, dtype=np.uint64)
    br, bi = f.forward_torus(b)
    back = np.zeros(2048, dtype=np.uint64)
    f.add_backward_torus(back, br, bi)
    assert np.abs((back - b).astype(np.int64)).max() < 2**13
    a = rng.integers(-(2**22), 2**22, 2048).astype(np.int64)
    ar, ai = f.forward_integer(a)
    out = np.zeros(2048, dtype=np.uint64)
    f.add_backward_torus(out, ar * br - ai * bi, ar * bi + ai * br)
    ref = np.zeros(2048, dtype=np.uint64)
    L.orc_poly_add_mul(ref, a.astype(np.uint64), b, 2048)
    assert np.abs((out - ref).astype(np.int64)).max() < 2**45


def test_twiddle_tables(oracle):
    # Twisties::new (fft/mod.rs:58-69): (im, re) = sin_cos(i * pi/(2n)); exact symmetries of W
    f = oracle.Fft(2048)
    tr, ti = f.twist()
    i = np.arange(1024)
    assert np.abs(tr - np.cos(i * (np.pi / 2048))).max() < 2e-16
    assert np.abs(ti - np.sin(i * (np.pi / 2048))).max() < 2e-16
    assert tr[0] == 1.0 and ti[0] == 0.0
    wr, wi = f.twiddles()
    assert wr[0] == 1.0 and wi[0] == 0.0 and wr[256] == 0.0 and wi[256] == -1.0
    assert wr[512] == -1.0 and wi[512] == 0.0 and wr[128] == -wi[128]
    k = np.arange(1024)
    assert np.abs(wr - np.cos(2 * np.pi * k / 1024)).max() < 1e-15  # numpy rounds the angle
    assert np.abs(wi + np.sin(2 * np.pi * k / 1024)).max() < 1e-15


def test_fast_port_matches_the_oracle(oracle, keyset_small):
    """oracle/tfhe_oracle_fast.c -- the tuned port bench.py's cpu_baseline leg times -- returns the plain restatement's bits
    (same FFT schedule, same operation order; only the loop structure differs), incl. all-zero masks and every rotation"""
    import numpy as np

    ks = keyset_small
    rng = np.random.default_rng(17)
    cts = ks.encrypt(np.arange(24) % 16)
    cts[3, :-1] = 0                     # a trivial-looking ciphertext: all-zero mask
    cts[5] = rng.integers(0, 2**64, cts.shape[1], dtype=np.uint64)
    luts = np.stack([ks.lut([(x * x) % 16 for x in range(16)])[0], ks.lut(list(range(16)))[0]])
    idx = (np.arange(24) % 2).astype(np.uint32)
    want = ks.ks_pbs_batch(cts, luts, idx, threads=4)
    assert np.array_equal(ks.ks_pbs_batch(cts, luts, idx, threads=4, fast=True), want)
    assert np.array_equal(ks.ks_pbs_batch(cts, luts, idx, threads=1, fast=True), want)
