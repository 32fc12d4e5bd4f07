"""The split family, replace, substring / nth / repeat / add on the GPU at full PARAM_MESSAGE_2_CARRY_2_KS_PBS
(host_split.hpp through the C-ABI): decrypt and compare with Rust's `str` semantics, as the reference's tests do
(tfhe/examples/fhe_strings/test_generating_macros.rs:23-33; inputs of server_key/{split,rsplit,rsplitn,split_terminator,
split_inclusive,replace}.rs `mod tests`)."""
import os
import time

import numpy as np
import pytest

from tests.test_gpu_ops import env  # noqa: F401  (fixture)
from tests.test_host_split_cpu import (SPLIT_CASES, random_padding, std_rsplit, std_rsplit_terminator, std_rsplitn, std_split,
                                       std_split_ascii_whitespace, std_split_inclusive, std_split_terminator, std_splitn)

pytestmark = pytest.mark.gpu


def decrypt_str(env, blocks):
    d = env.ks.decrypt(blocks.reshape(-1, blocks.shape[-1])).reshape(-1, 4)
    assert (d < 4).all()
    return "".join(chr(int(sum(int(v) << (2 * i) for i, v in enumerate(ch)))) for ch in d)


def split_result(env, fhe):
    parts, number = fhe
    n = env.decrypt_radix(number)
    assert n <= len(parts), (n, len(parts))
    return [decrypt_str(env, p).replace("\x00", "") for p in parts[:n]]


STD = {"split": std_split, "rsplit": std_rsplit, "split_terminator": std_split_terminator,
       "rsplit_terminator": std_rsplit_terminator, "split_inclusive": std_split_inclusive}


def test_split_family_reference_cases_on_gpu(env):
    key = env.key
    for i, (s, pat) in enumerate(SPLIT_CASES):
        es = env.encrypt_str(s)
        form = pat if i % 2 else env.encrypt_str(pat)
        for kind, std in STD.items():
            assert split_result(env, getattr(key, kind)(es, form)) == std(s, pat), (kind, s, pat)


def test_splitn_whitespace_once_on_gpu(env):
    key = env.key
    for s, pat in [("aaaaa", "aa"), ("a.b.c.d", "."), ("acbc", "c")]:
        es = env.encrypt_str(s)
        for n in (0, 1, 3):
            assert split_result(env, key.splitn(es, n, pat)) == std_splitn(s, n, pat)
            assert split_result(env, key.rsplitn(es, env.encrypt_radix(n, 4), env.encrypt_str(pat))) == std_rsplitn(s, n, pat)
    s = "  Mary had\ta little  \n lamb "
    assert split_result(env, key.split_ascii_whitespace(env.encrypt_str(s))) == std_split_ascii_whitespace(s)
    found, before, after = key.split_once(env.encrypt_str("key=value=x"), "=")
    assert env.decrypt_bool(found) == 1
    assert (decrypt_str(env, before).replace("\x00", ""), decrypt_str(env, after).replace("\x00", "")) == ("key", "value=x")
    found, before, after = key.rsplit_once(env.encrypt_str("key=value=x"), env.encrypt_str("="))
    assert (decrypt_str(env, before).replace("\x00", ""), decrypt_str(env, after).replace("\x00", "")) == ("key=value", "x")
    assert env.decrypt_bool(key.split_once(env.encrypt_str("abc"), "=")[0]) == 0


def test_split_padded_and_long_on_gpu(env):
    key = env.key
    rng = np.random.default_rng(9)
    s, pat = "x--y--", "--"
    for padded, code in [(s + "\x00\x00", 1), (random_padding(rng, s, 3), 4)]:
        es = env.encrypt_str(padded)
        for kind, std in STD.items():
            assert split_result(env, getattr(key, kind)(es, pat, padding=code)) == std(s, pat), (kind, padded)
    s = "the quick brown fox, jumps over, the lazy dog, again"
    es = env.encrypt_str(s)
    key.stats(reset=True)
    t0 = time.perf_counter()
    got = key.split(es, ", ")
    dt = time.perf_counter() - t0
    st = key.stats()
    assert split_result(env, got) == s.split(", ")
    print(f"\n[split] {len(s)} chars by a 2-char clear pattern: {dt * 1e3:.1f} ms, {st['pbs']} PBS in {st['launches']} launches")
    s = ",".join("abcdefghijklmnopqrstu")  # 21 parts: both nibbles of the count
    assert split_result(env, key.split(env.encrypt_str(s), ",")) == s.split(",")


def test_padded_encrypted_patterns_on_gpu(env):
    """`encrypt_str_random_padding(pattern, 2)` patterns: encrypted length (test_generating_macros.rs:397-404)"""
    key = env.key
    rng = np.random.default_rng(31)
    for s, pat in [("acbccbcbcbc", "cbc"), ("aaaaa", "aa"), ("hello, world, x,", ", "), ("axbx", "x")]:
        ep = env.encrypt_str(random_padding(rng, pat, 2))
        es = env.encrypt_str(random_padding(rng, s, 2))
        for kind, std in STD.items():
            assert split_result(env, getattr(key, kind)(es, ep, padding=(4, 4))) == std(s, pat), (kind, s, pat)
        assert split_result(env, key.splitn(es, 2, ep, padding=(4, 4))) == std_splitn(s, 2, pat)
        assert decrypt_str(env, key.replace(es, ep, "<>", padding=(4, 4))[0]).replace("\x00", "") == s.replace(pat, "<>")


def test_replace_on_gpu(env):
    key = env.key
    for s, old, new in [("abab", "ab", "cd"), ("aaaa", "aa", "b"), ("a.b.c", ".", "::"), ("ababab", "ab", ""),
                        ("hello world, hello fhe", "hello", "HELLO!")]:
        es = env.encrypt_str(s)
        for f_old in (old, env.encrypt_str(old)):
            for f_new in ((new, env.encrypt_str(new)) if new else (new,)):
                content, pad = key.replace(es, f_old, f_new)
                assert decrypt_str(env, content).replace("\x00", "") == s.replace(old, new), (s, old, new)
        assert decrypt_str(env, key.replacen(es, old, new, 1)[0]).replace("\x00", "") == s.replace(old, new, 1)
        assert decrypt_str(env, key.replacen(es, old, new, env.encrypt_radix(2, 4))[0]).replace("\x00", "") == s.replace(old, new, 2)


def test_add_repeat_substring_nth_on_gpu(env):
    key = env.key
    a, b = env.encrypt_str("ab"), env.encrypt_str("bc")
    content, pad = key.add(a, b)
    assert decrypt_str(env, content) == "abbc" and pad == 0
    for n in (0, 2, 3):
        content, pad = key.repeat(a, env.encrypt_radix(n, 4), n_max=3)
        assert decrypt_str(env, content) == "ab" * n + "\x00" * (2 * (3 - n)) and pad == 1
    s = "hello world"
    es = env.encrypt_str(s)
    ok, content, pad = key.substring(es, 2, 7)
    assert env.decrypt_bool(ok) == 1 and decrypt_str(env, content) == s[2:7]
    for start, end in [(0, 11), (3, 5), (6, 2), (2, 14)]:
        ok, content, pad = key.substring(es, env.encrypt_radix(start, 4), env.encrypt_radix(end, 4))
        assert env.decrypt_bool(ok) == int(start <= end <= len(s)), (start, end)
        assert decrypt_str(env, content).replace("\x00", "") == s[start:end] and pad == 3, (start, end)
    for n in (0, 4, 10, 11):
        want = ord(s[n]) if n < len(s) else 0
        assert env.decrypt_radix(key.nth(es, n)) == want
        assert env.decrypt_radix(key.nth(es, env.encrypt_radix(n, 4))) == want


@pytest.mark.parametrize("seed", range(int(os.environ.get("TFHE_HIP_FUZZ_SEEDS", "6"))))
def test_split_fuzz_at_full_parameters(env, seed):
    key = env.key
    rng = np.random.default_rng(100 + seed)
    L = int(rng.integers(1, 40))
    s = "".join(rng.choice(list("abc "), size=L))
    pat = "".join(rng.choice(list("abc "), size=int(rng.integers(1, 4))))
    es = env.encrypt_str(s)
    form = pat if seed % 2 else env.encrypt_str(pat)
    kinds = list(STD.items())
    for kind, std in (kinds[seed % 5], kinds[(seed + 2) % 5]):
        assert split_result(env, getattr(key, kind)(es, form)) == std(s, pat), (kind, s, pat)
    n = int(rng.integers(0, 6))
    assert split_result(env, key.splitn(es, n, form)) == std_splitn(s, n, pat), (s, pat, n)
    assert split_result(env, key.split_ascii_whitespace(es)) == std_split_ascii_whitespace(s), s
    assert decrypt_str(env, key.replace(es, form, "xyz")[0]).replace("\x00", "") == s.replace(pat, "xyz"), (s, pat)


def test_encrypted_char_trims_and_padding_helpers_on_gpu(env):
    key = env.key
    content, length = key.trim_encrypted_char(env.encrypt_str("--a-b--"), env.encrypt_str("-")[0])
    assert decrypt_str(env, content).replace("\x00", "") == "a-b" and env.decrypt_radix(length) == 3
    assert decrypt_str(env, key.push_padding_to_start(env.encrypt_str("a\x00b\x00\x00c"))) == "\x00\x00\x00abc"
    ch, content = key.pop_non_zero_char(env.encrypt_str("\x00ab\x00c\x00"), last=True)
    assert env.decrypt_radix(ch) == ord("c") and decrypt_str(env, content) == "\x00ab\x00\x00\x00"
    ch, content = key.pop_non_zero_char(env.encrypt_str("\x00ab\x00c\x00"))
    assert env.decrypt_radix(ch) == ord("a") and decrypt_str(env, content) == "\x00\x00b\x00c\x00"


def test_long_one_hot_sums_at_full_parameters(env):
    """one-hot weighted sums of hundreds of terms (find index, nth select) stay inside the noise budget at the real
    parameter set: a 250-char string, several repetitions"""
    key = env.key
    rng = np.random.default_rng(5)
    for rep in range(3):
        s = "".join(rng.choice(list("abcdefghij"), size=250))
        pos = int(rng.integers(200, 246))
        s = s[:pos] + "XYZ" + s[pos + 3:]
        es = env.encrypt_str(s)
        found, index = key.find(es, env.encrypt_str("XYZ"))
        assert env.decrypt_bool(found) == 1 and env.decrypt_radix(index) == s.find("XYZ")
        n = int(rng.integers(0, 250))
        assert env.decrypt_radix(key.nth(es, env.encrypt_radix(n, 4))) == ord(s[n])


def test_resident_and_host_staged_executors_give_the_same_bits(env, monkeypatch):
    """the executor that keeps ciphertexts in HBM between levels (default) and the one that stages every level through the
    host, the round-filling scheduler and plain as-soon-as-possible levels: every bootstrap sees the same input row, the
    kernels are bit-exact, so the results agree bit for bit"""
    import tfhe_rs_string_amd as T

    s = env.encrypt_str("  alpha, beta,, gamma ")
    pat = env.encrypt_str(", ")

    def run(key):
        parts, number = key.split(s, pat)
        content, length = key.trim(s)
        found, index = key.find(s, pat)
        return [parts, number, content, length, found, index, key.replace(s, ", ", "--")[0]]

    want = run(env.key)
    monkeypatch.setenv("TFHE_HIP_HOST_STAGING", "1")
    monkeypatch.setenv("TFHE_HIP_HOST_ASAP", "1")
    other = T.StringServerKey(env.sk.handle, env.p)
    monkeypatch.delenv("TFHE_HIP_HOST_STAGING")
    monkeypatch.delenv("TFHE_HIP_HOST_ASAP")
    got = run(other)
    for a, b in zip(want, got):
        assert np.array_equal(a, b)


def test_empty_pattern_in_every_kind_at_full_parameters(env):
    """The empty-pattern op-graphs (clear "", an encrypted pattern of length 0, a pattern hidden under padding) of
    splitn / rsplitn / split_terminator / split_inclusive / split_once / replace / replacen at n = 742 on the GPU: the very
    cases of tests/test_host_split_cpu.py (reference semantics find.rs:321-554, rsplit.rs:169-230, rsplitn.rs:112-170),
    so that their packings meet full-parameter noise"""
    from tests import test_host_split_cpu as cpu_cases

    cpu_cases.test_empty_pattern_in_every_kind(env)


def test_rfind_of_an_empty_pattern_at_full_parameters(env):
    from tests import test_host_split_cpu as cpu_cases

    cpu_cases.test_rfind_of_an_empty_pattern_hidden_under_padding(env)
