"""Seeded (compressed) keys on the GPU: the device CSPRNG kernel against the oracle (which is pinned on the reference's
AES known-answer vectors, tests/test_seeded_keys_cpu.py), and keys created from their seeded form / from
`bincode::serialize(&CompressedServerKey)` bytes against keys created from the oracle's decompression: bit-exact
(T0: integer / byte work)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_csprng_matches_oracle(oracle):
    import tfhe_rs_string_amd as T

    ctx = T.Context(0)
    for seed, first, count in [(0x0F0E0D0C0B0A09080706050403020100, 0, 64), (0x0123456789ABCDEF0FEDCBA987654321, 0, 100003),
                               (1, 7, 1), (2**128 - 1, 12345, 4097), (0xDEADBEEF, 2**33 + 1, 1000)]:
        got = T.csprng_words(ctx, seed, first, count)
        want = oracle.csprng_words(seed, first, count)
        assert np.array_equal(got, want), (hex(seed), first, count)


@pytest.mark.parametrize("order", ["ks_pbs", "pbs_ks"])
def test_seeded_key_equals_decompressed_key(oracle, order):
    """reduced n, full N: same kernels; the seeded key must give the same Fourier key and the same outputs as the
    standard key made of the oracle's decompression"""
    import tfhe_rs_string_amd as T

    op = oracle.param_message_2_carry_2(lwe_dim=24) if order == "ks_pbs" else oracle.param_message_2_carry_2_pbs_ks(lwe_dim=24)
    ks = oracle.KeySet(op, seed=91, fourier=False)
    seeded = oracle.SeededKeys(ks)
    bsk, ksk = seeded.decompress()
    assert np.array_equal(bsk, ks.bsk) and np.array_equal(ksk, ks.ksk)
    p = (T.PARAM_MESSAGE_2_CARRY_2_KS_PBS if order == "ks_pbs" else T.PARAM_MESSAGE_2_CARRY_2_PBS_KS)(lwe_dim=24)
    ctx = T.Context(0)
    ref = T.ServerKey(ctx, p, bsk, ksk)
    data = T.serialize_compressed_server_key(p, seeded.bsk_bodies, seeded.bsk_seed, seeded.ksk_bodies, seeded.ksk_seed)
    for key in (T.ServerKey.from_seeded(ctx, p, seeded.bsk_bodies, seeded.bsk_seed, seeded.ksk_bodies, seeded.ksk_seed),
                T.ServerKey.from_compressed_bytes(ctx, data)):
        for a, b in zip(key.fourier_bsk(), ref.fourier_bsk()):
            assert np.array_equal(a, b)
        assert key.params.lwe_dim == 24 and key.params.pbs_order == p.pbs_order
        msgs = [i % 16 for i in range(48)]
        lut = T.generate_lookup_table(p, [(5 * x + 3) % 16 for x in range(16)])[0]
        cts = ks.encrypt(msgs) if order == "ks_pbs" else ks.encrypt_small(msgs)
        run = (lambda k: k.ks_pbs_batch(cts, lut, 0)) if order == "ks_pbs" else (lambda k: k.pbs_ks_batch(cts, lut, 0))
        out, want = run(key), run(ref)
        assert np.array_equal(out, want)
        dec = ks.decrypt(out) if order == "ks_pbs" else ks.decrypt_small(out)
        assert list(dec) == [(5 * m + 3) % 16 for m in msgs]


def test_full_size_compressed_key_and_ciphertext_stream(oracle):
    """PARAM_MESSAGE_2_CARRY_2_KS_PBS at full size: 25 MB of bodies instead of 110 MB of keys cross PCIe; ciphertexts go
    through their bincode form"""
    import time

    import tfhe_rs_string_amd as T

    ks = oracle.KeySet(oracle.param_message_2_carry_2(), seed=0x5EED0003, fourier=False)
    seeded = oracle.SeededKeys(ks, fourier=False)
    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
    data = T.serialize_compressed_server_key(p, seeded.bsk_bodies, seeded.bsk_seed, seeded.ksk_bodies, seeded.ksk_seed)
    assert len(data) < 25_000_000
    ctx = T.Context(0)
    t0 = time.perf_counter()
    key = T.ServerKey.from_compressed_bytes(ctx, data)
    dt = time.perf_counter() - t0
    print(f"\n[seeded] CompressedServerKey ({len(data) / 1e6:.1f} MB) -> device key: {dt * 1e3:.0f} ms")
    msgs = list(range(16)) * 2
    stream = b"".join(T.serialize_ciphertext(p, ct, degree=3) for ct in ks.encrypt(msgs))
    cts, pos = [], 0
    while pos < len(stream):
        lwe, deg, noise, mm, cm, order, used = T.deserialize_ciphertext(stream[pos:])
        assert (deg, noise, mm, cm, order) == (3, 1, 4, 4, 0)
        cts.append(lwe)
        pos += used
    lut = T.generate_lookup_table(p, [(x * x) % 16 for x in range(16)])[0]
    out = key.ks_pbs_batch(np.stack(cts), lut, 0)
    assert list(ks.decrypt(out)) == [(m * m) % 16 for m in msgs]


def test_compressed_ciphertexts_decompress_on_device(oracle):
    """shortint CompressedCiphertext: every ciphertext has its own seed; device == oracle, and the decompressed
    ciphertexts bootstrap to the right messages"""
    import ctypes as C

    import tfhe_rs_string_amd as T

    op = oracle.param_message_2_carry_2(lwe_dim=24)
    ks = oracle.KeySet(op, seed=5, fourier=False)
    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=24)
    ctx = T.Context(0)
    msgs = [i % 16 for i in range(70)]
    cts = ks.encrypt(msgs)
    dim = p.big_size - 1
    seeds = [0x9E3779B97F4A7C15F39CC0605CEDC834 * (i + 1) % 2**128 for i in range(len(msgs))]
    bodies = np.zeros(len(msgs), dtype=np.uint64)
    L = oracle.lib()
    for i, s in enumerate(seeds):  # compress: the mask becomes the seed's stream, the body follows
        one = np.zeros(1, dtype=np.uint64)
        L.orc_seeded_reseed_lwe_list(oracle.seed_bytes(s), ks.glwe_sk, dim, cts[i], 1, one)
        bodies[i] = one[0]
    wire = b"".join(T.serialize_compressed_ciphertext(p, int(b), s, degree=3) for b, s in zip(bodies, seeds))
    rb, rs, pos = [], [], 0
    while pos < len(wire):
        body, size, seed, deg, noise, mm, cm, order, used = T.deserialize_compressed_ciphertext(wire[pos:])
        assert size == p.big_size
        rb.append(body)
        rs.append(seed)
        pos += used
    got = T.decompress_ciphertexts(ctx, np.array(rb, dtype=np.uint64), rs, dim)
    assert np.array_equal(got, cts)  # == decompress_seeded_lwe_ciphertext of the oracle's stream
    want0 = np.zeros(dim + 1, dtype=np.uint64)
    L.orc_seeded_expand(oracle.seed_bytes(seeds[3]), bodies[3:4], 1, dim, 1, want0)
    assert np.array_equal(got[3], want0)
    sk = T.ServerKey(ctx, p, ks.bsk, ks.ksk)
    lut = T.generate_lookup_table(p, list(range(16)))[0]
    assert list(ks.decrypt(sk.ks_pbs_batch(got, lut, 0))) == msgs


def test_seeded_multi_bit_key(oracle):
    """a multi-bit key (grouping factor 2) from its seeded form / its bincode image == the key built from the oracle's
    decompression"""
    import tfhe_rs_string_amd as T

    op = oracle.param_multi_bit_message_2_carry_2_group_2(lwe_dim=24)
    ks = oracle.MultiBitKeySet(op, 2, seed=17)
    seeded = oracle.SeededKeys(ks)
    bsk, ksk = seeded.decompress()
    assert np.array_equal(bsk, ks.bsk) and np.array_equal(ksk, ks.ksk)
    p = T.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS(lwe_dim=24)
    ctx = T.Context(0)
    ref = T.ServerKey(ctx, p, bsk, ksk, grouping_factor=2)
    data = T.serialize_compressed_server_key(p, seeded.bsk_bodies, seeded.bsk_seed, seeded.ksk_bodies, seeded.ksk_seed, grouping_factor=2)
    info = T.compressed_server_key_info(data)
    assert info[1] == 2 and info[0].lwe_dim == 24
    msgs = [i % 16 for i in range(40)]
    lut = T.generate_lookup_table(p, [(7 * x + 1) % 16 for x in range(16)])[0]
    cts = ks.encrypt(msgs)
    want = ref.ks_pbs_batch(cts, lut, 0)
    assert list(ks.decrypt(want)) == [(7 * m + 1) % 16 for m in msgs]
    for key in (T.ServerKey.from_seeded(ctx, p, seeded.bsk_bodies, seeded.bsk_seed, seeded.ksk_bodies, seeded.ksk_seed, grouping_factor=2),
                T.ServerKey.from_compressed_bytes(ctx, data)):
        assert key.grouping_factor == 2
        assert np.array_equal(key.ks_pbs_batch(cts, lut, 0), want)
