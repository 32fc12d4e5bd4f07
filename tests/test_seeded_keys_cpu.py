"""Seeded (compressed) keys on the CPU: the oracle's AES-128-CTR CSPRNG against the reference's known-answer vectors
(tests/golden/aes128_fips197.json, from concrete-csprng's own block-cipher tests), the structure of the generator's
stream (concrete-csprng/src/generators/aes_ctr/generic.rs:38: a new generator starts at TableIndex::SECOND) and the
decompression of seeded bootstrap / keyswitch keys (core_crypto/algorithms/seeded_*_decompression.rs)."""
import ctypes as C
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aes128_fips197.json")


def _hex(h):
    return np.frombuffer(bytes.fromhex(h), dtype=np.uint8).copy()


def test_aes128_known_answers(oracle):
    L = oracle.lib()
    g = json.load(open(GOLDEN))
    rk = np.zeros(176, dtype=np.uint8)
    L.orc_aes128_key_schedule(_hex(g["key"]), rk)
    assert [rk[16 * i:16 * i + 16].tobytes().hex() for i in range(11)] == g["key_schedule"]
    out = np.zeros(16, dtype=np.uint8)
    L.orc_aes128_encrypt(rk, _hex(g["plaintext"]), out)
    assert out.tobytes().hex() == g["ciphertext"]
    sbox = np.ctypeslib.as_array(L.orc_aes_sbox(), shape=(256,))
    assert sbox[0] == 0x63 and sbox[0x53] == 0xED and sorted(sbox.tolist()) == list(range(256))  # FIPS-197 figure 7


def test_generator_stream_structure(oracle):
    """byte t of a fresh generator = byte t + 1 of AES_seed(0) || AES_seed(1) || ... with little-endian counters; u64 outputs
    are 8 consecutive bytes, little-endian; any stretch can be produced on its own (what forked children do)"""
    L = oracle.lib()
    seed = 0x0F0E0D0C0B0A09080706050403020100  # native bytes 00 01 .. 0f: the FIPS key
    sb = oracle.seed_bytes(seed)
    assert sb.tobytes().hex() == "000102030405060708090a0b0c0d0e0f"
    rk = np.zeros(176, dtype=np.uint8)
    L.orc_aes128_key_schedule(sb, rk)
    blocks = []
    for i in range(5):
        ctr, out = np.zeros(16, dtype=np.uint8), np.zeros(16, dtype=np.uint8)
        ctr[:8] = np.frombuffer(int(i).to_bytes(8, "little"), dtype=np.uint8)
        L.orc_aes128_encrypt(rk, ctr, out)
        blocks.append(out.tobytes())
    flat = b"".join(blocks)
    got = np.zeros(60, dtype=np.uint8)
    L.orc_csprng_bytes(sb, 0, got, 60)
    assert got.tobytes() == flat[1:61]
    part = np.zeros(23, dtype=np.uint8)
    L.orc_csprng_bytes(sb, 17, part, 23)
    assert part.tobytes() == flat[18:41]
    words = oracle.csprng_words(seed, 1, 4)
    assert [int(w) for w in words] == [int.from_bytes(flat[1 + 8 * (i + 1):1 + 8 * (i + 2)], "little") for i in range(4)]


def test_seeded_keys_decompress_to_working_keys(oracle):
    """a re-seeded toy key set: decompression reproduces the keys bit for bit, and they still bootstrap correctly"""
    p = oracle.param_message_2_carry_2(lwe_dim=16)
    ks = oracle.KeySet(p, seed=77, fourier=False)
    before = (ks.bsk.copy(), ks.ksk.copy())
    sk = oracle.SeededKeys(ks)
    assert not np.array_equal(before[0], ks.bsk) and not np.array_equal(before[1], ks.ksk)
    bsk, ksk = sk.decompress()
    assert np.array_equal(bsk, ks.bsk) and np.array_equal(ksk, ks.ksk)
    n, N = p.lwe_dim, p.poly_size
    assert sk.bsk_bodies.size == n * 1 * 2 * N and sk.ksk_bodies.size == N * p.ks_level
    # masks are the stream, in memory order
    assert np.array_equal(bsk.reshape(-1, 2, N)[:3, 0].ravel(), oracle.csprng_words(sk.bsk_seed, 0, 3 * N))
    assert np.array_equal(ksk.reshape(-1, n + 1)[5, :n], oracle.csprng_words(sk.ksk_seed, 5 * n, n))
    msgs = list(range(16))
    lut = ks.lut([(3 * x + 1) % 16 for x in range(16)])[0]
    out = ks.ks_pbs_batch(ks.encrypt(msgs), lut[None], np.zeros(16, dtype=np.uint32))
    assert list(ks.decrypt(out)) == [(3 * x + 1) % 16 for x in msgs]


# ---- wire format (tfhe_rs_string_amd/csrc/wire.cpp): pure host code of the product library, no GPU needed ---------
def _bincode_compressed_key(p, g, bsk_bodies, bsk_seed, ksk_bodies, ksk_seed):
    """independent statement of `bincode::serialize(&CompressedServerKey)` (bincode 1.3 defaults), field by field in the
    declaration order of the serde derives (shortint/server_key/compressed.rs:10-17, 43-55,
    core_crypto/entities/seeded_lwe_keyswitch_key.rs:10-21, seeded_ggsw_ciphertext_list.rs:11-23,
    seeded_lwe_multi_bit_bootstrap_key.rs:16-25, commons/ciphertext_modulus.rs:41-64)"""
    import struct

    u64 = lambda v: struct.pack("<Q", v)
    u32 = lambda v: struct.pack("<I", v)
    vec = lambda a: u64(a.size) + a.astype("<u8").tobytes()
    native = (0).to_bytes(16, "little") + u64(64)
    out = vec(ksk_bodies) + u64(p.ks_base_log) + u64(p.ks_level) + u64(p.lwe_dim + 1) + int(ksk_seed).to_bytes(16, "little") + native
    out += u32(1 if g else 0)
    out += vec(bsk_bodies) + u64(p.glwe_dim + 1) + u64(p.poly_size) + u64(p.pbs_base_log) + u64(p.pbs_level)
    out += int(bsk_seed).to_bytes(16, "little") + native
    if g:
        out += u64(g) + b"\x01"
    out += u64(p.message_modulus) + u64(p.carry_modulus) + u64(p.message_modulus * p.carry_modulus - 1) + native + u32(p.pbs_order)
    return out


def test_compressed_server_key_bincode_round_trip(oracle):
    import tfhe_rs_string_amd as T

    rng = np.random.default_rng(3)
    for p, g in [(T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=12), 0), (T.PARAM_MESSAGE_2_CARRY_2_PBS_KS(lwe_dim=10), 0),
                 (T.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS(lwe_dim=9), 3)]:
        ggsw = (p.lwe_dim // g) << g if g else p.lwe_dim
        bsk_bodies = rng.integers(0, 2**64, ggsw * p.pbs_level * (p.glwe_dim + 1) * p.poly_size, dtype=np.uint64)
        ksk_bodies = rng.integers(0, 2**64, p.glwe_dim * p.poly_size * p.ks_level, dtype=np.uint64)
        s1, s2 = 0x0123456789ABCDEF0FEDCBA987654321, 0xFFEEDDCCBBAA99887766554433221100
        data = T.serialize_compressed_server_key(p, bsk_bodies, s1, ksk_bodies, s2, grouping_factor=g)
        assert data == _bincode_compressed_key(p, g, bsk_bodies, s1, ksk_bodies, s2)
        q, g2, max_degree, used = T.compressed_server_key_info(data + b"trailing")
        assert used == len(data) and g2 == g and max_degree == 15
        for f, _ in p._fields_:
            assert getattr(q, f) == getattr(p, f), f
        # corrupt inputs are refused, with the position
        for bad in (data[:100], data[:-3], b"\xff" * 64):
            try:
                T.compressed_server_key_info(bad)
                raise AssertionError("accepted a corrupt key")
            except T.TfheHipError as e:
                assert "bincode" in str(e)
        broken = bytearray(data)
        broken[8 + ksk_bodies.size * 8 + 24 + 16] = 1  # a custom ciphertext modulus
        try:
            T.compressed_server_key_info(bytes(broken))
            raise AssertionError("accepted a custom modulus")
        except T.TfheHipError as e:
            assert "native" in str(e)


def test_compressed_key_header_fields_that_wrap_or_truncate_are_errors():
    """the compressed-key reader bounds every untrusted u64 header field before multiplying or narrowing it (ADVICE r2:
    glwe_size = 2^63 with two levels made the per-GGSW product 0 -> a division by zero)"""
    import copy

    import pytest
    import tfhe_rs_string_amd as T

    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=12)
    bsk_bodies = np.zeros(p.lwe_dim * p.pbs_level * (p.glwe_dim + 1) * p.poly_size, dtype=np.uint64)
    ksk_bodies = np.zeros(p.glwe_dim * p.poly_size * p.ks_level, dtype=np.uint64)

    class Fields:  # duck-typed parameter set with unbounded integers
        pass

    def blob(**over):
        q = Fields()
        for f, _ in p._fields_:
            setattr(q, f, getattr(p, f))
        for k, v in over.items():
            setattr(q, k, v)
        return _bincode_compressed_key(q, 0, bsk_bodies, 1, ksk_bodies, 2)

    T.compressed_server_key_info(blob())  # the untouched image is accepted
    for over in (dict(glwe_dim=(1 << 63) - 1, pbs_level=2), dict(pbs_level=1 << 63), dict(poly_size=1 << 62), dict(poly_size=2047),
                 dict(ks_level=p.ks_level + (1 << 32)), dict(ks_level=1 << 63), dict(lwe_dim=1 << 40), dict(pbs_base_log=65),
                 dict(message_modulus=1 << 40)):
        with pytest.raises(T.TfheHipError, match="bincode"):
            T.compressed_server_key_info(blob(**over))


def test_shortint_ciphertext_bincode_round_trip(oracle):
    import struct

    import tfhe_rs_string_amd as T

    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
    lwe = np.random.default_rng(4).integers(0, 2**64, p.big_size, dtype=np.uint64)
    data = T.serialize_ciphertext(p, lwe, degree=3, noise_level=1)
    # shortint/ciphertext/mod.rs:261-270: ct { data, ciphertext_modulus }, degree, noise_level, message_modulus, carry_modulus, pbs_order
    want = struct.pack("<Q", lwe.size) + lwe.tobytes() + (0).to_bytes(16, "little") + struct.pack("<QQQQQI", 64, 3, 1, 4, 4, 0)
    assert data == want and len(data) == 8 + 8 * 2049 + 16 + 8 + 4 * 8 + 4
    got = T.deserialize_ciphertext(data * 2)  # a stream of ciphertexts: `consumed` finds the next one
    assert np.array_equal(got[0], lwe) and got[1:] == (3, 1, 4, 4, 0, len(data))
    try:
        T.deserialize_ciphertext(data[:-1])
        raise AssertionError("accepted a truncated ciphertext")
    except T.TfheHipError as e:
        assert "truncated" in str(e)


def test_compressed_ciphertext_bincode_round_trip(oracle):
    import struct

    import tfhe_rs_string_amd as T

    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
    seed = 0x1F1E1D1C1B1A19181716151413121110
    data = T.serialize_compressed_ciphertext(p, body=0x1122334455667788, seed=seed, degree=3)
    # shortint/ciphertext/mod.rs:470-478: ct { data, lwe_size, compression_seed, ciphertext_modulus }, degree, message_modulus,
    # carry_modulus, pbs_order, noise_level
    want = (struct.pack("<QQ", 0x1122334455667788, 2049) + seed.to_bytes(16, "little") + (0).to_bytes(16, "little") +
            struct.pack("<QQQQIQ", 64, 3, 4, 4, 0, 1))
    assert data == want
    got = T.deserialize_compressed_ciphertext(data + data)
    assert got == (0x1122334455667788, 2049, seed.to_bytes(16, "little"), 3, 1, 4, 4, 0, len(data))
    try:
        T.deserialize_compressed_ciphertext(data[:40])
        raise AssertionError("accepted a truncated ciphertext")
    except T.TfheHipError as e:
        assert "truncated" in str(e)
