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
    deps = srcs + [os.path.join(CSRC, "host_ops.hpp"), os.path.join(CSRC, "host_split.hpp"), os.path.join(ROOT, "oracle", "liboracle.so")]
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
            n = 2 * len(s1)  # one PBS per packed pair of blocks
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
    # one flag per distinct (position, nibble half, nibble value) the 11 windows ask for -- "hello" has ONE high nibble
    # (6) and four low ones --, then AND per window and OR 11 -> 1
    pat, windows = "hello", len(s) - 5 + 1
    distinct = {(w + j, half, (ord(pat[j]) >> (4 * half)) & 15) for w in range(windows) for j in range(5) for half in (0, 1)}
    assert len(distinct) == 60 < 11 * 10
    assert key.stats()["pbs"] == len(distinct) + 11 + 1


def decrypt_str(env, blocks):
    d = env.ks.decrypt(blocks.reshape(-1, blocks.shape[-1])).reshape(-1, 4)
    assert (d < 4).all()  # clean carries
    return "".join(chr(int(sum(int(v) << (2 * i) for i, v in enumerate(ch)))) for ch in d)


def test_integer_comparisons_and_selection(env):
    """gt/ge/lt/le (encrypted and scalar), min/max, if_then_else
    (radix_parallel/tests_cases_unsigned.rs default_{gt,ge,lt,le,min,max}_test, default_if_then_else_test)"""
    key = env.key
    rng = np.random.default_rng(11)
    nb = 4
    cases = [(0, 0), (255, 255), (255, 0), (0, 255), (0x40, 0x3F), (0x10, 0x0F), (5, 5)]
    cases += [tuple(int(v) for v in rng.integers(0, 256, 2)) for _ in range(4)]
    ops = {"gt": lambda a, b: a > b, "ge": lambda a, b: a >= b, "lt": lambda a, b: a < b, "le": lambda a, b: a <= b}
    for a, b in cases:
        ea, eb = env.encrypt_radix(a, nb), env.encrypt_radix(b, nb)
        for name, f in ops.items():
            assert env.decrypt_bool(key.compare_parallelized(ea, eb, name)) == int(f(a, b)), (a, b, name)
            assert env.decrypt_bool(key.compare_parallelized(ea, b, name)) == int(f(a, b)), (a, b, name, "scalar")
    a, b = 0x5A, 0xA5
    ea, eb = env.encrypt_radix(a, nb), env.encrypt_radix(b, nb)
    assert env.decrypt_radix(key.max_parallelized(ea, eb)) == max(a, b)
    assert env.decrypt_radix(key.min_parallelized(ea, eb)) == min(a, b)
    assert env.decrypt_bool(key.gt_parallelized(ea, 1000)) == 0  # scalar wider than the ciphertext
    for cond in (0, 1):
        ec = env.ks.encrypt([cond])[0]
        got = key.if_then_else_parallelized(ec, ea, eb)
        assert env.decrypt_radix(got) == (a if cond else b)
        assert (env.ks.decrypt(got) < 4).all()
    # odd block count (the last packed pair holds one block)
    ea, eb = env.encrypt_radix(0x155, 5), env.encrypt_radix(0x255, 5)
    assert env.decrypt_bool(key.lt_parallelized(ea, eb)) == 1
    assert env.decrypt_bool(key.ge_parallelized(ea, eb)) == 0


def test_integer_add_sub_neg_scalar(env):
    """add / sub / neg / scalar_add / scalar_mul, wrapping (tests_cases_unsigned.rs default_add_test,
    default_sub_test, default_neg_test, default_scalar_add_test, default_scalar_mul_test)"""
    key = env.key
    nb = 4
    mod = 4**nb
    rng = np.random.default_rng(12)
    cases = [(0, 0), (255, 1), (1, 255), (0, 1), (200, 100)] + [tuple(int(v) for v in rng.integers(0, 256, 2)) for _ in range(3)]
    for a, b in cases:
        ea, eb = env.encrypt_radix(a, nb), env.encrypt_radix(b, nb)
        for got, want in ((key.add_parallelized(ea, eb), (a + b) % mod), (key.sub_parallelized(ea, eb), (a - b) % mod),
                          (key.scalar_add_parallelized(ea, b), (a + b) % mod), (key.scalar_mul_parallelized(ea, b), (a * b) % mod)):
            assert env.decrypt_radix(got) == want, (a, b)
            assert (env.ks.decrypt(got) < 4).all()
        assert env.decrypt_radix(key.neg_parallelized(ea)) == (-a) % mod


def test_string_order_and_case(env):
    """le / ge (encrypted and clear right-hand side), to_uppercase / to_lowercase, eq_ignore_case
    (comparisons.rs test_le_ge, test_le_ge_clear; change_case.rs test_to_upper_fhe, test_to_lower_fhe)"""
    key = env.key
    words = ["abc", "abd", "ab", "abcd", "b", "", "abc", "aBc", "Zz", "zZ"]
    for a in words:
        ea = env.encrypt_str(a)
        for b in words:
            eb = env.encrypt_str(b)
            assert env.decrypt_bool(key.le(ea, eb)) == int(a <= b), (a, b)
            assert env.decrypt_bool(key.ge(ea, eb)) == int(a >= b), (a, b)
            assert env.decrypt_bool(key.le(ea, b)) == int(a <= b), (a, b, "clear")
            assert env.decrypt_bool(key.ge(ea, b)) == int(a >= b), (a, b, "clear")
    s = "aB.`az{@AZ[ ~09pP"
    es = env.encrypt_str(s)
    assert decrypt_str(env, key.to_uppercase(es)) == s.upper()
    assert decrypt_str(env, key.to_lowercase(es)) == s.lower()
    assert decrypt_str(env, key.to_uppercase(env.encrypt_str("aB."))) == "AB."  # the reference's doc-test
    assert decrypt_str(env, key.to_lowercase(env.encrypt_str("BCD"))) == "bcd"
    key.stats(reset=True)
    key.to_lowercase(es)
    # nibble classes, "is a letter of the other case", and (strict outputs, the default) the cleaning bootstrap of the
    # case-adjusted block: the reference's output contract (noise level NOMINAL)
    assert key.stats() == {"pbs": 4 * len(s), "launches": 3}
    key.set_strict_outputs(False)
    key.stats(reset=True)
    low = key.to_lowercase(es)
    assert key.stats() == {"pbs": 3 * len(s), "launches": 2}  # the adjustment is leveled and handed back as it is
    assert decrypt_str(env, low) == s.lower()
    key.set_strict_outputs(True)
    for a, b in [("Hello", "hELLO"), ("Hello", "hELLo!"), ("abc", "abd"), ("[", "{"), ("", "")]:
        assert env.decrypt_bool(key.eq_ignore_case(env.encrypt_str(a), env.encrypt_str(b))) == int(a.lower() == b.lower()), (a, b)
        assert env.decrypt_bool(key.eq_ignore_case(env.encrypt_str(a), b)) == int(a.lower() == b.lower()), (a, b, "clear")


def test_string_find_rfind(env):
    """find / rfind -> FheOptionInt (find.rs doc-tests: "aba".find("a") == Some(0), rfind == Some(2))"""
    key = env.key

    def check(s, pat, enc):
        es = env.encrypt_str(s)
        ep = env.encrypt_str(pat) if enc else pat
        for rev, fn, clear in ((0, key.find, s.find(pat)), (1, key.rfind, s.rfind(pat))):
            found, index = fn(es, ep)
            assert env.decrypt_bool(found) == int(clear >= 0), (s, pat, rev)
            if clear >= 0:
                assert env.decrypt_radix(index) == clear, (s, pat, rev)
            assert (env.ks.decrypt(index) < 4).all()

    for enc in (True, False):
        check("aba", "a", enc)
        check("aba", "b", enc)
        check("aba", "c", enc)
        check("hello fhe world, hello again", "hello", enc)
        check("hello fhe world, hello again", "again", enc)
        check("hello fhe world, hello again", "xyz", enc)
        check("abc", "abcd", enc)
        check("abc", "", enc)
    s = "ab" * 30 + "c" + "ab" * 10  # 81 windows: three levels of the blocked scan
    check(s, "c", True)
    check(s, "ba", False)
    found, index = key.find(env.encrypt_str(s), "zz")
    assert env.decrypt_bool(found) == 0 and env.decrypt_radix(index) == len(s)  # the reference's loop leaves len


def test_strings_with_final_padding(env):
    """Padding::Final inputs, as `encrypt_str_padding(s, n)` produces them (client_key.rs): eq, starts_with,
    contains, find, le / ge, case (reference tests: test_*_padding in starts_with.rs / contains.rs / find.rs
    with string_padding / pattern_padding in {0, 2})"""
    key = env.key
    pad = lambda s, n: env.encrypt_str(s + "\x00" * n)
    for a, b in [("abc", "abc"), ("abc", "abd"), ("abc", "ab"), ("", ""), ("a", "")]:
        for na in (0, 2):
            for nb in (0, 3):
                got = key.eq_padded(pad(a, na), pad(b, nb), padding=(int(na > 0), int(nb > 0)))
                assert env.decrypt_bool(got) == int(a == b), (a, na, b, nb)
                assert env.decrypt_bool(key.le(pad(a, na), pad(b, nb))) == int(a <= b), (a, na, b, nb)
                assert env.decrypt_bool(key.ge(pad(a, na), pad(b, nb))) == int(a >= b), (a, na, b, nb)
    s = "hello fhe"
    for pat in ["hello", "fhe", "hello fhe", "hello fhe!", "lo f", "x", ""]:
        for ns in (0, 2):
            for npat in (0, 2):
                es, ep = pad(s, ns), pad(pat, npat)
                padding = (int(ns > 0), int(npat > 0))
                assert env.decrypt_bool(key.starts_with_padded(es, ep, padding)) == int(s.startswith(pat)), (pat, ns, npat)
                assert env.decrypt_bool(key.contains_padded(es, ep, padding)) == int(pat in s), (pat, ns, npat)
                found, index = key.find_padded(es, ep, padding)
                assert env.decrypt_bool(found) == int(pat in s), (pat, ns, npat)
                if pat in s:
                    assert env.decrypt_radix(index) == s.find(pat), (pat, ns, npat)
    found, index = key.find_padded(pad("abcabc", 2), pad("bc", 3), (1, 1), reverse=True)
    assert env.decrypt_bool(found) == 1 and env.decrypt_radix(index) == 4
    assert decrypt_str(env, key.to_uppercase(pad("aB.", 2))) == "AB.\x00\x00"
    assert env.decrypt_bool(key.eq_ignore_case(pad("Hello", 2), pad("hELLO", 2))) == 1


def test_trim_and_strip(env):
    """trim_start / trim_end / trim / trim_*_char, strip_prefix / strip_suffix: decrypt, drop the padding zeros and
    compare with str (trim.rs commented tests: trim_start_char("ab", 'a') == "b" ...; strip.rs tests)"""
    key = env.key

    def plain(content, length, want):
        got = decrypt_str(env, content).replace("\x00", "")
        assert got == want, (got, want)
        assert env.decrypt_radix(length) == len(want)

    for s in [" \t hello  world \n ", "   ", "", "x", "  x", "x  ", "\r\n a b \x0c"]:
        es = env.encrypt_str(s)
        plain(*key.trim_start(es), s.lstrip(" \t\n\x0b\x0c\r"))
        plain(*key.trim_end(es), s.rstrip(" "))
        plain(*key.trim(es), s.strip(" "))
    es = env.encrypt_str("aabaXaa")
    plain(*key.trim_char(es, "a"), "baX")
    plain(*key.trim_char(es, "a", right=False), "baXaa")
    plain(*key.trim_char(es, "a", left=False), "aabaX")
    plain(*key.trim_char(env.encrypt_str("ab"), "a", right=False), "b")  # the reference's test_trim_start_char
    long_s = " " * 20 + "fhe" + " " * 25  # more than one 15-chunk of the scan on both sides
    plain(*key.trim(env.encrypt_str(long_s)), "fhe")
    s = "prefix-body-suffix"
    es = env.encrypt_str(s)
    for pat, suffix in [("prefix", 0), ("suffix", 1), ("body", 0), ("body", 1), ("", 0), (s, 0), (s + "x", 1)]:
        for ep in (env.encrypt_str(pat), pat):
            found, content, length = (key.strip_suffix if suffix else key.strip_prefix)(es, ep)
            hit = s.endswith(pat) if suffix else s.startswith(pat)
            assert env.decrypt_bool(found) == int(hit), (pat, suffix)
            want = (s[: len(s) - len(pat)] if suffix else s[len(pat):]) if hit else s
            plain(content, length, want)


def test_push_padding_to_end_and_any_padding(env):
    """push_padding_to_end (change_padding.rs; the reference's commented test: padding zeros anywhere are moved
    behind the characters, order kept) and operations on `encrypt_str_random_padding`-style inputs"""
    key = env.key
    rng = np.random.default_rng(17)

    def random_padding(s, n):  # client_key.rs encrypt_str_random_padding: n zeros at random positions
        chars = list(s)
        for _ in range(n):
            chars.insert(int(rng.integers(0, len(chars) + 1)), "\x00")
        return "".join(chars)

    # up to 24 chars: the scatter by rank; beyond: the shift network (both also forced on a mid-size input)
    for s, n in [("abc", 2), ("hello fhe world", 9), ("", 3), ("x", 0), ("abcdefgh", 8), ("ab", 17),
                 ("the shift network moves these", 11), ("z", 30)]:
        padded = random_padding(s, n)
        got = decrypt_str(env, key.push_padding_to_end(env.encrypt_str(padded)))
        assert got == s + "\x00" * n, (padded, got)
    padded = random_padding("both ways", 7)
    by_rank = key.push_padding_to_end(env.encrypt_str(padded))
    os.environ["TFHE_HIP_COMPACT_NETWORK"] = "1"
    try:
        by_network = key.push_padding_to_end(env.encrypt_str(padded))
    finally:
        del os.environ["TFHE_HIP_COMPACT_NETWORK"]
    assert decrypt_str(env, by_rank) == decrypt_str(env, by_network) == "both ways" + "\x00" * 7
    s = "the quick brown fox"
    es = env.encrypt_str(random_padding(s, 6))
    for pat in ("quick", "fox", "the", "dog"):
        ep = env.encrypt_str(random_padding(pat, 2))
        assert env.decrypt_bool(key.contains_padded(es, ep, (4, 4))) == int(pat in s), pat
        assert env.decrypt_bool(key.starts_with_padded(es, ep, (4, 4))) == int(s.startswith(pat)), pat
        found, index = key.find_padded(es, ep, (4, 4))
        assert env.decrypt_bool(found) == int(pat in s)
        if pat in s:
            assert env.decrypt_radix(index) == s.find(pat)
    assert env.decrypt_bool(key.eq_padded(es, env.encrypt_str("\x00" + s), (4, 2))) == 1
    assert env.decrypt_bool(key.eq_padded(es, env.encrypt_str(s[:-1] + "\x00y"), (4, 4))) == 0


def test_len_and_is_empty_of_padded_strings(env):
    key = env.key
    for padded in ["ab\x00c\x00", "\x00\x00\x00", "abc", "", "\x00" * 9 + "x" * 19 + "\x00" * 3, "y" * 15, "y" * 16,
                   "q" * 44 + "\x00", "\x00" + "w" * 59, "v" * 61 + "\x00" * 2, "u" * 63, "\x00" * 70]:
        es = env.encrypt_str(padded)
        assert env.decrypt_radix(key.len(es)) == len(padded.replace("\x00", "")), repr(padded)
        assert env.decrypt_bool(key.is_empty(es)) == int(padded.replace("\x00", "") == ""), repr(padded)
    assert env.decrypt_radix(key.len(env.encrypt_str("abc"), padding=0)) == 3
    assert env.decrypt_bool(key.is_empty(env.encrypt_str("abc"), padding=0)) == 0


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


def test_host_key_rejects_the_other_pbs_order(env):
    import tfhe_rs_string_amd as T

    p = T.PARAM_MESSAGE_2_CARRY_2_PBS_KS(lwe_dim=env.params.lwe_dim)
    with pytest.raises(T.TfheHipError, match="KeyswitchBootstrap"):
        T.StringServerKey(env.sk, p, lib=env.L)


def test_errors(env):
    T = env.T
    out = np.zeros(env.params.big_size, dtype=np.uint64)
    assert env.L.tfhe_hip_integer_eq(env.key._h, None, None, 4, out.ctypes.data_as(C.c_void_p)) == 1
    assert b"null" in env.L.tfhe_hip_last_error()
    with pytest.raises(T.TfheHipError):
        env.key.contains_windows(env.encrypt_str("abc"), env.encrypt_str("ab"), 0, 5)


def test_errors_of_the_wider_api(env):
    """argument validation of the operations beyond eq / contains: the call fails with 1 and names the check
    (the reference panics on the corresponding asserts)"""
    T, key = env.T, env.key
    es = env.encrypt_str("abc")
    with pytest.raises(T.TfheHipError, match="padding"):
        key.eq_padded(es, es, padding=(7, 0))
    with pytest.raises(T.TfheHipError, match="trim mode"):
        key._trim(es, 5)
    with pytest.raises(T.TfheHipError, match="comparison op"):
        key._check(env.L.tfhe_hip_integer_compare(key._h, es.ctypes.data_as(C.c_void_p), es.ctypes.data_as(C.c_void_p), 4, 9,
                                                   np.zeros(env.params.big_size, dtype=np.uint64).ctypes.data_as(C.c_void_p)))
    long_s = np.zeros((256, 4, env.params.big_size), dtype=np.uint64)  # 256 chars: the index would not fit 8 bits
    with pytest.raises(T.TfheHipError, match="255"):
        key.find(long_s, "a")
    with pytest.raises(T.TfheHipError, match="255"):
        key.trim(long_s)
    with pytest.raises(ValueError, match="block counts"):
        key.compare_parallelized(env.encrypt_radix(1, 4), env.encrypt_radix(1, 3), "gt")
    with pytest.raises(ValueError, match="block counts"):
        key.add_parallelized(env.encrypt_radix(1, 4), env.encrypt_radix(1, 3))
    out = np.zeros((4, env.params.big_size), dtype=np.uint64)
    assert env.L.tfhe_hip_string_len(key._h, None, 3, 1, out.ctypes.data_as(C.c_void_p)) == 1
    assert b"null" in env.L.tfhe_hip_last_error()
    # a failed call leaves no pending work behind: the next call is unaffected
    assert env.decrypt_bool(key.eq(es, env.encrypt_str("abc"))) == 1


def test_batch_mode_merges_independent_operations(env):
    """batch mode: operations staged between batch_begin / batch_end run as ONE graph -- independent operations
    share their launches -- and the outputs are written at the end"""
    key = env.key
    words = ["alpha", "beta", "gamma", "delta", "alpha", "omega"]
    enc = [env.encrypt_str(w) for w in words]
    target = env.encrypt_str("alpha")
    key.stats(reset=True)
    single = [env.decrypt_bool(key.eq(e, target)) for e in enc]
    st_single = key.stats()
    key.stats(reset=True)
    with key.batch():
        outs = [key.eq(e, target) for e in enc]
        contains = [key.contains_clear(e, "a") for e in enc]
        upper = [key.to_uppercase(e) for e in enc]
        assert key.stats()["launches"] == 0  # nothing has run yet
    st_batch = key.stats()
    assert [env.decrypt_bool(o) for o in outs] == single == [int(w == "alpha") for w in words]
    assert [env.decrypt_bool(o) for o in contains] == [int("a" in w) for w in words]
    assert [decrypt_str(env, u) for u in upper] == [w.upper() for w in words]
    assert st_batch["launches"] <= 4 < st_single["launches"]  # the deepest graph decides, not the number of operations
    # an invalid call inside a batch fails alone; the batch still completes
    with key.batch():
        ok = key.eq(enc[0], target)
        with pytest.raises(env.T.TfheHipError):
            key._trim(enc[0], 9)
    assert env.decrypt_bool(ok) == 1
    with pytest.raises(env.T.TfheHipError, match="no batch is open"):
        key._check(env.L.tfhe_hip_host_key_batch_end(key._h))


def test_host_key_is_safe_under_concurrent_host_threads(env):
    """several host threads on ONE host key (operations are serialised by its lock) and on separate host keys over the
    same server key (the reference shares its ServerKey between rayon workers)"""
    import threading

    strings = ["alpha beta", "gamma", "delta epsilon zeta", "eta"]
    enc = [env.encrypt_str(s) for s in strings]
    other = env.T.StringServerKey(env.sk, env.params, lib=env.L)
    results, errors = {}, []

    def work(tid, key):
        try:
            for rep in range(3):
                for i, (s, e) in enumerate(zip(strings, enc)):
                    results[(tid, rep, i, "c")] = (env.decrypt_bool(key.contains_clear(e, "eta")), "eta" in s)
                    results[(tid, rep, i, "u")] = (decrypt_str(env, key.to_uppercase(e)), s.upper())
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    threads = [threading.Thread(target=work, args=(t, env.key if t < 3 else other)) for t in range(5)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(results) == 5 * 3 * 4 * 2 and all(got == want for got, want in results.values())


def test_host_key_checks_its_params_against_the_server_key(env):
    import tfhe_rs_string_amd as T

    wrong = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=env.params.lwe_dim + 1)
    with pytest.raises(T.TfheHipError, match="params differ"):
        T.StringServerKey(env.sk, wrong, lib=env.L)
