"""Uncompressed (standard-domain) server keys on the wire -- SURVEY 8f-3's first item: `bincode::serialize` images of
LweKeyswitchKey<Vec<u64>> (tfhe/src/core_crypto/entities/lwe_keyswitch_key.rs:76-108) and LweBootstrapKey<Vec<u64>>
(lwe_bootstrap_key.rs:98-106 -> ggsw_ciphertext_list.rs:9-20).  The layout is stated a second time here, independently of
wire.cpp, from the serde derives (field order = declaration order; bincode 1.3 defaults: little-endian fixed-width integers,
usize / sequence lengths as u64, u128 as 16 bytes, newtype structs transparent).  PARITY UNPINNED: the reference holds no
serialized fixture."""
import struct

import numpy as np
import pytest


def native_modulus():
    # CiphertextModulus<u64> serializes as SerializableCiphertextModulus { modulus: u128 (0 = native 2^64), scalar_bits: usize }
    # (core_crypto/commons/ciphertext_modulus.rs:41-64)
    return (0).to_bytes(16, "little") + struct.pack("<Q", 64)


def py_ksk_bytes(ksk, base_log, level, out_dim):
    return struct.pack("<Q", ksk.size) + ksk.tobytes() + struct.pack("<QQQ", base_log, level, out_dim + 1) + native_modulus()


def py_bsk_bytes(bsk, glwe_dim, poly, base_log, level):
    return struct.pack("<Q", bsk.size) + bsk.tobytes() + struct.pack("<QQQQ", glwe_dim + 1, poly, base_log, level) + native_modulus()


@pytest.fixture(scope="module")
def T():
    import tfhe_rs_string_amd as T

    return T


def test_writers_match_the_independent_layout_and_readers_round_trip(T, keyset_small):
    ks = keyset_small
    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=ks.p.lwe_dim)
    kb = T.serialize_keyswitch_key(p, ks.ksk)
    bb = T.serialize_bootstrap_key(p, ks.bsk)
    assert kb.tobytes() == py_ksk_bytes(ks.ksk, 3, 5, p.lwe_dim)
    assert bb.tobytes() == py_bsk_bytes(ks.bsk, 1, 2048, 23, 1)
    ksk, bl, lv, din, dout, used = T.deserialize_keyswitch_key(kb)
    assert np.array_equal(ksk, ks.ksk) and (bl, lv, din, dout, used) == (3, 5, 2048, p.lwe_dim, kb.size)
    bsk, k, N, bl, lv, din, used = T.deserialize_bootstrap_key(bb)
    assert np.array_equal(bsk, ks.bsk) and (k, N, bl, lv, din, used) == (1, 2048, 23, 1, p.lwe_dim, bb.size)
    # a key followed by other data: `consumed` says where it ends
    assert T.deserialize_keyswitch_key(np.concatenate([kb, np.arange(9, dtype=np.uint8)]))[5] == kb.size


def test_corrupt_and_truncated_inputs_are_errors(T, keyset_small):
    ks = keyset_small
    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=ks.p.lwe_dim)
    kb = T.serialize_keyswitch_key(p, ks.ksk)
    with pytest.raises(T.TfheHipError, match="truncated"):
        T.deserialize_keyswitch_key(kb[:-5])
    with pytest.raises(T.TfheHipError, match="sequence length exceeds"):
        T.deserialize_keyswitch_key(kb[:100])
    bad = kb.copy()
    bad[-24] = 1  # a custom ciphertext modulus
    with pytest.raises(T.TfheHipError, match="custom ciphertext moduli"):
        T.deserialize_keyswitch_key(bad)
    bad = kb.copy()
    bad[8 + ks.ksk.size * 8 + 16:8 + ks.ksk.size * 8 + 24] = np.frombuffer(struct.pack("<Q", 7), dtype=np.uint8)  # output_lwe_size 7
    with pytest.raises(T.TfheHipError, match="inconsistent keyswitch key header"):
        T.deserialize_keyswitch_key(bad)
    bb = T.serialize_bootstrap_key(p, ks.bsk)
    with pytest.raises(T.TfheHipError, match="truncated"):
        T.deserialize_bootstrap_key(bb[:-1])


def test_header_fields_whose_products_wrap_are_errors_not_crashes(T):
    """untrusted u64 header fields: level_count * output_lwe_size (and the four-factor GGSW product) wrapping to 0
    mod 2^64 used to be an integer division by zero (SIGFPE) in the readers; every field is now bounded before it is multiplied"""
    data = np.arange(4, dtype=np.uint64)
    for base_log, level, out_size in [(3, 1 << 63, 2), (3, 1 << 32, 1 << 32), (3, 5, 1 << 63), (1 << 63, 5, 7), (0, 5, 7), (33, 2, 2)]:
        blob = struct.pack("<Q", data.size) + data.tobytes() + struct.pack("<QQQ", base_log, level, out_size) + native_modulus()
        with pytest.raises(T.TfheHipError, match="out of range|inconsistent"):
            T.deserialize_keyswitch_key(np.frombuffer(blob, dtype=np.uint8))
    for glwe_size, poly, base_log, level in [(1 << 32, 1 << 32, 23, 1), (2, 1 << 62, 23, 4), (1 << 63, 2, 23, 2), (2, 2048, 23, 1 << 63),
                                             (2, 2047, 23, 1), (2, 0, 23, 1)]:
        blob = struct.pack("<Q", data.size) + data.tobytes() + struct.pack("<QQQQ", glwe_size, poly, base_log, level) + native_modulus()
        with pytest.raises(T.TfheHipError, match="out of range|inconsistent|power of two"):
            T.deserialize_bootstrap_key(np.frombuffer(blob, dtype=np.uint8))
    # a field that would read correctly once truncated to 32 bits must not pass
    ksk = np.zeros(2 * 5 * 3, dtype=np.uint64)
    blob = struct.pack("<Q", ksk.size) + ksk.tobytes() + struct.pack("<QQQ", 3, 5 + (1 << 32), 3) + native_modulus()
    with pytest.raises(T.TfheHipError, match="out of range"):
        T.deserialize_keyswitch_key(np.frombuffer(blob, dtype=np.uint8))


@pytest.mark.gpu
def test_server_key_from_standard_key_bytes_bootstraps_like_the_key_from_arrays(T, keyset_small):
    ks = keyset_small
    ctx = T.Context(0)
    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=ks.p.lwe_dim)
    sk = T.ServerKey.from_standard_key_bytes(ctx, py_ksk_bytes(ks.ksk, 3, 5, p.lwe_dim), py_bsk_bytes(ks.bsk, 1, 2048, 23, 1))
    q = sk.params
    assert (q.lwe_dim, q.glwe_dim, q.poly_size, q.pbs_base_log, q.pbs_level, q.ks_base_log, q.ks_level, q.pbs_order) == \
        (p.lwe_dim, 1, 2048, 23, 1, 3, 5, 0)
    msgs = np.arange(24) % 16
    cts = ks.encrypt(msgs)
    lut, _ = T.generate_lookup_table(q, lambda x: (3 * x + 1) % 16)
    got = sk.ks_pbs_batch(cts, lut, 0)
    assert np.array_equal(got, ks.ks_pbs_batch(cts, lut, np.zeros(24, dtype=np.uint32), threads=4))
    assert np.array_equal(ks.decrypt(got).astype(np.int64), (3 * msgs + 1) % 16)
    # mismatched pair: the bootstrap key of another dimension
    other = np.zeros(4 * 1 * 2 * 2 * 2048, dtype=np.uint64)
    with pytest.raises(T.TfheHipError, match="input dimension"):
        T.ServerKey.from_standard_key_bytes(ctx, py_ksk_bytes(ks.ksk, 3, 5, p.lwe_dim), py_bsk_bytes(other, 1, 2048, 23, 1))
