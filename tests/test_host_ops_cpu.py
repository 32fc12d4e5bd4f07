"""CPU coverage of the host op-graph layer (tfhe_rs_string_amd/csrc/host_ops.hpp through the C-ABI
of include/tfhe_hip_strings.h): the product's host code is linked against a TEST-ONLY backend that
answers `tfhe_hip_ks_pbs_batch` with the CPU oracle (tests/emu/oracle_backend.cpp), so level
scheduling, degree bookkeeping, the trivial shortcut and the window sharding are checked without a
GPU.  Tests mirror the reference's: encrypt -> op -> decrypt -> compare with the clear result
(tfhe/examples/fhe_strings/server_key/contains.rs:165-282, comparisons.rs tests,
tfhe/src/integer/server_key/radix_parallel/tests_cases_unsigned.rs default_mul_test)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")
CSRC = os.path.join(ROOT, "tfhe_rs_string_amd", "csrc")


@pytest.fixture(scope="module")
def backend(oracle):
    lib = os.path.join(EMU, "libhost_emu.so")
    srcs = [os.path.join(CSRC, "host_api.cpp"), os.path.join(EMU, "oracle_backend.cpp")]
    deps = srcs + [os.path.join(CSRC, "host_ops.hpp"), os.path.join(ROOT, "oracle", "liboracle.so")]
    if not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", lib] + srcs +
                              ["-L" + os.path.join(ROOT, "oracle"), "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    L = C.CDLL(lib)
    L.emu_server_key_create.restype = C.c_void_p
    L.emu_server_key_create.argtypes = [C.c_void_p] * 4 + [C.c_int]
    L.emu_server_key_destroy.argtypes = [C.c_void_p]
    return L


class Env:
    def __init__(self, oracle, backend, ks):
        import tfhe_rs_string_amd as T

        self.T, self.ks, self.L = T, ks, backend
        self.params = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=ks.p.lwe_dim)
        self.sk = backend.emu_server_key_create(C.byref(ks.p), ks.ksk.ctypes.data, ks.bsk_re.ctypes.data,
                                                ks.bsk_im.ctypes.data, 8)
        self.key = T.StringServerKey(self.sk, self.params, lib=backend)

    # client side (fhe_strings/client_key.rs:27-60): a char is a 4-block radix, least significant first
    def encrypt_str(self, s):
        blocks = [(ord(ch) >> (2 * i)) & 3 for ch in s for i in range(4)]
        return self.ks.encrypt(blocks).reshape(len(s), 4, -1) if s else np.zeros((0, 4, self.params.big_size), dtype=np.uint64)

    def encrypt_radix(self, value, nb):
        return self.ks.encrypt([(value >> (2 * i)) & 3 for i in range(nb)])

    def decrypt_radix(self, blocks):
        return sum(int(d % 4) << (2 * i) for i, d in enumerate(self.ks.decrypt(blocks)))

    def decrypt_bool(self, block):
        return int(self.ks.decrypt(block.reshape(1, -1))[0])


@pytest.fixture(scope="module")
def env(oracle, backend, keyset_small):
    return Env(oracle, backend, keyset_small)


def test_symbols_declared_are_exported(backend):
    import re

    hdr = open(os.path.join(ROOT, "include", "tfhe_hip_strings.h")).read()
    declared = set(re.findall(r"\b(tfhe_hip_[a-z0-9_]+)\s*\(", hdr)) - {"tfhe_hip_last_error"}
    from tfhe_rs_string_amd import _native, strings

    assert declared == set(strings.exported_symbols())
    prod = C.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(prod, name), f"{name} missing from libtfhe_hip.so"


def test_shortint_apply_lookup_table_and_trivial_shortcut(env):
    T, ks, p = env.T, env.ks, env.params
    key = T.ShortintServerKey(env.sk, p, lib=env.L)
    lut, deg = T.generate_lookup_table(p, lambda x: (x * x) % 16)
    msgs = np.arange(8)
    cts = ks.encrypt(msgs)
    # append trivial ciphertexts (mask zero, noise 0): they must not reach the PBS
    delta = 1 << 59
    triv = np.zeros((3, p.big_size), dtype=np.uint64)
    triv[:, -1] = [np.uint64(2 * delta), np.uint64(3 * delta), np.uint64(0)]
    allc = np.concatenate([cts, triv])
    noise = np.array([1] * 8 + [0] * 3)
    out, degs, nz = key.apply_lookup_table(allc, 3, noise, lut, deg, 0)
    assert key.stats()["pbs"] == 8 and key.stats()["launches"] == 1
    assert [int(x) for x in ks.decrypt(out[:8])] == [(m * m) % 16 for m in msgs]
    assert [int(v) for v in out[8:, -1]] == [4 * delta, 9 * delta, 0] and not out[8:, :-1].any()
    assert list(degs) == [deg] * 11 and list(nz) == [1] * 8 + [0] * 3


@pytest.mark.parametrize("a,b", [(0x5A, 0x5A), (0x5A, 0x5B), (0, 0), (0xFF, 0x7F)])
def test_integer_eq(env, a, b):
    got = env.key.eq_parallelized(env.encrypt_radix(a, 4), env.encrypt_radix(b, 4))
    assert env.decrypt_bool(got) == int(a == b)


def test_string_eq(env):
    key = env.key
    cases = [("hello world", "hello world"), ("hello world", "hello worle"), ("abc", "abcd"), ("", ""), ("a", "b"),
             ("The quick brown fox", "The quick brown fox")]
    for s1, s2 in cases:
        key.stats(reset=True)
        got = key.eq(env.encrypt_str(s1), env.encrypt_str(s2))
        assert env.decrypt_bool(got) == int(s1 == s2), (s1, s2)
        if len(s1) == len(s2) and s1:
            st = key.stats()
            n = 4 * len(s1)
            levels, pbs = 1, n
            while n > 1:
                n = -(-n // 15)
                pbs += n
                levels += 1
            assert st["launches"] == levels and st["pbs"] == pbs


def test_string_contains(env):
    key = env.key
    hay = "the needle is in here"
    for needle in ["needle", "the", "here", "xyz", "e", "the needle is in here", "the needle is in here!", ""]:
        got = key.contains(env.encrypt_str(hay), env.encrypt_str(needle))
        assert env.decrypt_bool(got) == int(needle in hay), needle


def test_more_string_ops_encrypted_and_clear_patterns(env):
    """the next string operations on the same engine: ne, starts_with, ends_with, and the clear-pattern
    (&str) variants of eq / contains / starts_with / ends_with (reference tests: starts_with.rs,
    ends_with.rs, contains.rs `test_contains_clear_string_padding`, comparisons.rs)"""
    key = env.key
    s = "hello fhe world"
    es = env.encrypt_str(s)
    assert env.decrypt_bool(key.ne(es, env.encrypt_str(s))) == 0
    assert env.decrypt_bool(key.ne(es, env.encrypt_str("hello fhe w0rld"))) == 1
    for pat in ["hello", "world", "hello fhe world", "x", ""]:
        ep = env.encrypt_str(pat)
        assert env.decrypt_bool(key.starts_with(es, ep)) == int(s.startswith(pat)), pat
        assert env.decrypt_bool(key.ends_with(es, ep)) == int(s.endswith(pat)), pat
        assert env.decrypt_bool(key.starts_with(es, pat)) == int(s.startswith(pat)), pat
        assert env.decrypt_bool(key.ends_with(es, pat)) == int(s.endswith(pat)), pat
        assert env.decrypt_bool(key.contains_clear(es, pat)) == int(pat in s), pat
        assert env.decrypt_bool(key.eq_clear(es, pat)) == int(pat == s), pat
    assert env.decrypt_bool(key.contains_clear(es, "fhe")) == 1
    assert env.decrypt_bool(key.contains_clear(es, "\x00\x01")) == 0  # bytes < 16: x == 0 nibble LUT
    assert env.decrypt_bool(key.eq_clear(env.encrypt_str("\x05\x10"), "\x05\x10")) == 1
    # clear patterns cost 2 PBS per char comparison instead of 4
    key.stats(reset=True)
    key.eq_clear(es, s)
    assert key.stats()["pbs"] == 2 * len(s) + 2 + 1  # 30 -> 2 -> 1
    key.stats(reset=True)
    key.contains_clear(es, "hello")
    assert key.stats()["pbs"] == 11 * 10 + 11 + 1  # 11 windows x 5 chars x 2; AND per window; OR 11 -> 1


def test_contains_windows_match_positions(env):
    key = env.key
    hay, needle = "abcabcab", "bc"
    w = key.contains_windows(env.encrypt_str(hay), env.encrypt_str(needle), 0, key.window_count(env.encrypt_str(hay), env.encrypt_str(needle)))
    assert [int(x) for x in env.ks.decrypt(w)] == [int(hay[i:i + 2] == needle) for i in range(len(hay) - 1)]
    assert env.decrypt_bool(key.boolean_any(w)) == 1 and env.decrypt_bool(key.boolean_all(w)) == 0


def test_integer_mul(env):
    rng = np.random.default_rng(21)
    for nb in (4, 8):
        for _ in range(2):
            a, b = (int(x) for x in rng.integers(0, 4**nb, 2))
            got = env.key.mul_parallelized(env.encrypt_radix(a, nb), env.encrypt_radix(b, nb))
            assert env.decrypt_radix(got) == (a * b) % 4**nb
    # carries are clean: every block decrypts below the message modulus
    assert all(int(d) < 4 for d in env.ks.decrypt(got))


def test_errors(env):
    T = env.T
    out = np.zeros(env.params.big_size, dtype=np.uint64)
    assert env.L.tfhe_hip_integer_eq(env.key._h, None, None, 4, out.ctypes.data_as(C.c_void_p)) == 1
    assert b"null" in env.L.tfhe_hip_last_error()
    with pytest.raises(T.TfheHipError):
        env.key.contains_windows(env.encrypt_str("abc"), env.encrypt_str("ab"), 0, 5)
