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
