"""Randomised cross-check of the host op-graph layer against Python's str semantics on the CPU (oracle
backend, reduced lwe_dim): every operation, random strings / patterns, encrypted and clear patterns.
The reference tests its string layer the same way (decrypt and compare with the std result,
e.g. fhe_strings/server_key/contains.rs:165-282, find.rs tests)."""
import numpy as np
import pytest

from tests.test_host_ops_cpu import backend, decrypt_str, env  # noqa: F401  (fixtures)

ALPHABET = "ab AB.z"


def rand_str(rng, lo, hi):
    return "".join(ALPHABET[int(i)] for i in rng.integers(0, len(ALPHABET), int(rng.integers(lo, hi + 1))))


@pytest.mark.parametrize("seed", range(16))
def test_random_strings(env, seed):
    rng = np.random.default_rng(1000 + seed)
    key = env.key
    s = rand_str(rng, 0, 12)
    # a pattern that occurs (a slice of s) or a random one
    if len(s) and rng.integers(0, 2):
        a = int(rng.integers(0, len(s)))
        pat = s[a:a + int(rng.integers(0, 4))]
    else:
        pat = rand_str(rng, 0, 4)
    other = s if rng.integers(0, 3) == 0 else rand_str(rng, 0, 12)
    es, ep, eo = env.encrypt_str(s), env.encrypt_str(pat), env.encrypt_str(other)
    b = env.decrypt_bool
    ctx = (s, pat, other)
    assert b(key.eq(es, eo)) == int(s == other), ctx
    assert b(key.ne(es, eo)) == int(s != other), ctx
    assert b(key.le(es, eo)) == int(s <= other), ctx
    assert b(key.ge(es, eo)) == int(s >= other), ctx
    assert b(key.le(es, other)) == int(s <= other), ctx
    assert b(key.ge(es, other)) == int(s >= other), ctx
    assert b(key.eq_clear(es, other)) == int(s == other), ctx
    assert b(key.eq_ignore_case(es, eo)) == int(s.lower() == other.lower()), ctx
    for enc_pat in (ep, pat):
        assert b(key.starts_with(es, enc_pat)) == int(s.startswith(pat)), ctx
        assert b(key.ends_with(es, enc_pat)) == int(s.endswith(pat)), ctx
        for fn, want in ((key.find, s.find(pat)), (key.rfind, s.rfind(pat))):
            found, index = fn(es, enc_pat)
            assert b(found) == int(want >= 0), ctx
            if want >= 0:
                assert env.decrypt_radix(index) == want, ctx
        for fn, hit, rest in ((key.strip_prefix, s.startswith(pat), s[len(pat):]), (key.strip_suffix, s.endswith(pat), s[:len(s) - len(pat)])):
            found, content, length = fn(es, enc_pat)
            assert b(found) == int(hit), ctx
            want = rest if hit else s
            assert decrypt_str(env, content).replace("\x00", "") == want, ctx
            assert env.decrypt_radix(length) == len(want), ctx
    assert b(key.contains(es, ep)) == int(pat in s), ctx
    assert b(key.contains_clear(es, pat)) == int(pat in s), ctx
    assert decrypt_str(env, key.to_uppercase(es)) == s.upper(), ctx
    assert decrypt_str(env, key.to_lowercase(es)) == s.lower(), ctx
    for (content, length), want in ((key.trim_start(es), s.lstrip()), (key.trim_end(es), s.rstrip(" ")), (key.trim(es), s.strip(" ")),
                                    (key.trim_char(es, "a"), s.strip("a"))):
        assert decrypt_str(env, content).replace("\x00", "") == want, ctx
        assert env.decrypt_radix(length) == len(want), ctx
    # the same string with zeros sprinkled in (Padding::Anywhere) behaves like the string
    chars = list(s)
    for _ in range(int(rng.integers(0, 4))):
        chars.insert(int(rng.integers(0, len(chars) + 1)), "\x00")
    padded = "".join(chars)
    ez = env.encrypt_str(padded)
    if padded:
        assert decrypt_str(env, key.push_padding_to_end(ez)) == s + "\x00" * (len(padded) - len(s)), (ctx, padded)
    assert b(key.contains_padded(ez, ep, (4, 0))) == int(pat in s), (ctx, padded)
    assert b(key.eq_padded(ez, eo, (4, 0))) == int(s == other), (ctx, padded)


@pytest.mark.parametrize("seed", range(6))
def test_random_integers(env, seed):
    rng = np.random.default_rng(2000 + seed)
    key = env.key
    nb = int(rng.integers(1, 6))
    mod = 4**nb
    a, c = (int(v) for v in rng.integers(0, mod, 2))
    if rng.integers(0, 3) == 0:
        c = a
    ea, ec = env.encrypt_radix(a, nb), env.encrypt_radix(c, nb)
    b = env.decrypt_bool
    for name, want in (("gt", a > c), ("ge", a >= c), ("lt", a < c), ("le", a <= c)):
        assert b(key.compare_parallelized(ea, ec, name)) == int(want), (a, c, nb, name)
        assert b(key.compare_parallelized(ea, c, name)) == int(want), (a, c, nb, name, "scalar")
    assert b(key.eq_parallelized(ea, ec)) == int(a == c)
    r = env.decrypt_radix
    assert r(key.add_parallelized(ea, ec)) == (a + c) % mod
    assert r(key.sub_parallelized(ea, ec)) == (a - c) % mod
    assert r(key.mul_parallelized(ea, ec)) == (a * c) % mod
    assert r(key.neg_parallelized(ea)) == (-a) % mod
    assert r(key.scalar_add_parallelized(ea, c)) == (a + c) % mod
    assert r(key.scalar_mul_parallelized(ea, c)) == (a * c) % mod
    assert r(key.max_parallelized(ea, ec)) == max(a, c)
    assert r(key.min_parallelized(ea, ec)) == min(a, c)
