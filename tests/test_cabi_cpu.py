"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/tfhe_hip.h
declares, and its host-only entry points (LUT generation, argument checks) behave like the
reference's (`fill_accumulator`, tfhe/src/shortint/engine/mod.rs:72-128).  No GPU compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    from tfhe_rs_string_amd import build

    build.build()
    from tfhe_rs_string_amd import _native

    return _native


def test_exports_every_declared_symbol(native):
    hdr = open(os.path.join(ROOT, "include", "tfhe_hip.h")).read() + open(os.path.join(ROOT, "include", "tfhe_hip_experimental.h")).read()
    declared = set(re.findall(r"\b(tfhe_hip_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = C.CDLL(native.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/tfhe_hip.h / tfhe_hip_experimental.h but not exported"
    assert declared == set(native.exported_symbols())


def test_no_oracle_dependency(native):
    # the product library must not link or reference the oracle
    import subprocess

    out = subprocess.check_output(["ldd", native.LIB_PATH]).decode()
    assert "oracle" not in out
    syms = subprocess.check_output(["nm", "-D", native.LIB_PATH]).decode()
    assert "orc_" not in syms


def test_lut_generate_matches_oracle(native, oracle):
    import tfhe_rs_string_amd as T

    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
    op = oracle.param_message_2_carry_2()
    rng = np.random.default_rng(0)
    for _ in range(20):
        table = rng.integers(0, 16, 16).astype(np.uint64)
        got, deg = T.generate_lookup_table(p, table)
        want = np.zeros(op.lut_len, dtype=np.uint64)
        wdeg = oracle.lib().orc_lut_generate(C.byref(op), table, want)
        assert np.array_equal(got, want) and deg == wdeg


def test_classic_parameter_sets_are_accepted_and_their_luts_match_the_oracle(native, oracle):
    """the 15 classic KS_PBS sets of 1 ... 5 bits (glwe_dim 1 ... 5, N = 256 ... 4096; shortint/parameters/mod.rs:599-820) pass the
    table of the harness: sizes and the accumulator of a random function are the oracle's (engine/mod.rs:94-127) for each
    (that the kernels take them, and refuse a 6-bit set with N = 8192, is a GPU test: test_gpu_parity.py)"""
    import tfhe_rs_string_amd as T

    assert len(T.CLASSIC_KS_PBS) == 15
    rng = np.random.default_rng(1)
    for name, v in T.CLASSIC_KS_PBS.items():
        p = T.classic_params(name)
        op = oracle.param_custom(v[0], v[2], v[3], v[4], v[5], v[6], glwe_dim=v[1], message_modulus=v[7], carry_modulus=v[8])
        assert p.big_size == op.big_size and p.lut_len == op.lut_len, name
        space = v[7] * v[8]
        table = rng.integers(0, space, space).astype(np.uint64)
        got, deg = T.generate_lookup_table(p, table)
        want = np.zeros(op.lut_len, dtype=np.uint64)
        wdeg = oracle.lib().orc_lut_generate(C.byref(op), table, want)
        assert np.array_equal(got, want) and deg == wdeg, name


def test_error_returns_not_panics(native):
    import tfhe_rs_string_amd as T

    L = native.lib()
    # null out pointer -> 1 with a message (c_api/utils.rs:14-28 conventions)
    assert L.tfhe_hip_context_create(None, None, 0) == 1
    assert b"null" in L.tfhe_hip_last_error()
    assert L.tfhe_hip_lut_generate(None, None, None, None) == 1
    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
    with pytest.raises(ValueError):
        T.generate_lookup_table(p, [1, 2, 3])
    # compute entry points reject a null handle instead of crashing
    buf = np.zeros(4, dtype=np.uint64)
    assert L.tfhe_hip_keyswitch_batch(None, buf.ctypes.data, buf.ctypes.data, 1) == 1
    assert L.tfhe_hip_ks_pbs_batch_device(None, None, None, 0, None, None, 0, None) == 1
    # the newer entry points follow the same convention
    out, n = C.c_void_p(), C.c_size_t()
    seed = bytes(16)
    assert L.tfhe_hip_server_key_create_seeded(None, C.byref(out), C.byref(p), 0, buf.ctypes.data, 4, seed, buf.ctypes.data, 4, seed) == 1
    assert out.value is None
    assert L.tfhe_hip_server_key_create_seeded(None, C.byref(out), C.byref(p), 0, buf.ctypes.data, 4, None, buf.ctypes.data, 4, seed) == 1
    assert b"seed" in L.tfhe_hip_last_error()
    assert L.tfhe_hip_server_key_create_seeded(None, C.byref(out), C.byref(p), 7, buf.ctypes.data, 4, seed, buf.ctypes.data, 4, seed) == 1
    assert b"grouping_factor" in L.tfhe_hip_last_error()
    assert L.tfhe_hip_server_key_create_from_compressed(None, C.byref(out), b"\x01\x02", 2, None, None) == 1
    assert b"bincode" in L.tfhe_hip_last_error()
    assert L.tfhe_hip_csprng_fill(None, seed, 0, buf.ctypes.data, 4) == 1
    assert L.tfhe_hip_lwe_decompress_seeded_batch(None, buf.ctypes.data, seed, 4, buf.ctypes.data, 1) == 1
    assert L.tfhe_hip_device_alloc(None, C.byref(out), 16) == 1 and out.value is None
    assert L.tfhe_hip_device_upload(None, None, None, 0) == 1
    assert L.tfhe_hip_device_download(None, None, None, 0) == 1
    assert L.tfhe_hip_device_synchronize(None) == 1
    assert L.tfhe_hip_device_lincomb_rows(None, None, 1, 8, None, None, None, None) == 1
    assert L.tfhe_hip_device_free(None) == 0
    assert L.tfhe_hip_server_key_batch_rounds(None, C.byref(n), C.byref(n)) == 1
