"""BASELINE.json configs 3-5 on the GPU at full PARAM_MESSAGE_2_CARRY_2_KS_PBS: the host op-graph
layer drives level-batched KS-PBS launches through the C-ABI; results are checked like the
reference checks them (decrypt and compare with the clear computation:
tfhe/examples/fhe_strings/server_key/contains.rs:165-282, comparisons.rs tests,
tfhe/src/integer/server_key/radix_parallel/tests_cases_unsigned.rs default_mul_test)."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(keyset_full):
    import tfhe_rs_string_amd as T

    ks = keyset_full
    ctx = T.Context(0)
    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
    sk = T.ServerKey(ctx, p, ks.bsk, ks.ksk)
    key = T.StringServerKey(sk.handle, p)

    class E:
        pass

    e = E()
    e.T, e.ks, e.sk, e.key, e.p, e.ctx = T, ks, sk, key, p, ctx
    e.encrypt_str = lambda s: (ks.encrypt([(ord(ch) >> (2 * i)) & 3 for ch in s for i in range(4)]).reshape(len(s), 4, -1) if s
                               else np.zeros((0, 4, p.big_size), dtype=np.uint64))
    e.encrypt_radix = lambda v, nb: ks.encrypt([(v >> (2 * i)) & 3 for i in range(nb)])
    e.decrypt_radix = lambda blocks: sum(int(d % 4) << (2 * i) for i, d in enumerate(ks.decrypt(blocks)))
    e.decrypt_bool = lambda b: int(ks.decrypt(b.reshape(1, -1))[0])
    return e


def rand_ascii(rng, n):
    return "".join(chr(int(c)) for c in rng.integers(32, 127, n))


def test_config3_string_eq_64_chars(env):
    rng = np.random.default_rng(3)
    s1 = rand_ascii(rng, 64)
    s2 = s1[:37] + chr(ord(s1[37]) ^ 1) + s1[38:]
    e1, e2, e1b = env.encrypt_str(s1), env.encrypt_str(s2), env.encrypt_str(s1)
    env.key.stats(reset=True)
    t0 = time.perf_counter()
    r_eq = env.key.eq(e1, e1b)
    dt = time.perf_counter() - t0
    st = env.key.stats()
    assert env.decrypt_bool(r_eq) == 1
    assert env.decrypt_bool(env.key.eq(e1, e2)) == 0
    assert st == {"pbs": 128 + 9 + 1, "launches": 3}  # packed-pair equalities, 15-ary all-true tree
    print(f"\n[config 3] FheString::eq 64/64 chars: {dt * 1e3:.1f} ms, {st['pbs']} PBS in {st['launches']} launches")


def test_config4_string_contains_256_8(env):
    rng = np.random.default_rng(4)
    hay = rand_ascii(rng, 256)
    absent = "~~~~~~~~"
    hay = hay.replace("~", "-")
    ehay = env.encrypt_str(hay)
    for pos in (0, 123, 248, None):
        needle = absent if pos is None else hay[pos:pos + 8]
        env.key.stats(reset=True)
        t0 = time.perf_counter()
        got = env.key.contains(ehay, env.encrypt_str(needle))
        dt = time.perf_counter() - t0
        assert env.decrypt_bool(got) == int(needle in hay)
    st = env.key.stats()
    assert st["launches"] == 1 + 2 + 3  # pair comparisons, per-window AND tree (16->2->1), OR tree (249->17->2->1)
    assert st["pbs"] == 249 * 16 + 249 * 2 + 249 + 17 + 2 + 1
    print(f"\n[config 4, 1 GPU] FheString::contains 256/8: {dt * 1e3:.1f} ms, {st['pbs']} PBS in {st['launches']} launches")


def test_config5_fheuint32_mul(env):
    rng = np.random.default_rng(5)
    for _ in range(2):
        a, b = (int(x) for x in rng.integers(0, 2**32, 2))
        ea, eb = env.encrypt_radix(a, 16), env.encrypt_radix(b, 16)
        env.key.stats(reset=True)
        t0 = time.perf_counter()
        got = env.key.mul_parallelized(ea, eb)
        dt = time.perf_counter() - t0
        st = env.key.stats()
        assert env.decrypt_radix(got) == (a * b) % 2**32
    print(f"\n[config 5] FheUint32 mul (16 blocks): {dt * 1e3:.1f} ms, {st['pbs']} PBS in {st['launches']} launches "
          f"(reference: 333 ms on m6i.metal, tfhe/docs/getting_started/benchmarks.md:17)")


def test_integer_eq_32bit(env):
    a = 0xDEADBEEF
    assert env.decrypt_bool(env.key.eq_parallelized(env.encrypt_radix(a, 16), env.encrypt_radix(a, 16))) == 1
    assert env.decrypt_bool(env.key.eq_parallelized(env.encrypt_radix(a, 16), env.encrypt_radix(a ^ (1 << 20), 16))) == 0


def test_more_string_ops_on_gpu(env):
    """ne / starts_with / ends_with with encrypted and clear (&str) patterns, eq_clear, contains_clear at the
    full parameter set (decrypt and compare, as starts_with.rs / ends_with.rs / contains.rs tests do)"""
    s = "the quick brown fox jumps over"
    es = env.encrypt_str(s)
    key = env.key
    assert env.decrypt_bool(key.ne(es, env.encrypt_str(s))) == 0
    assert env.decrypt_bool(key.ne(es, env.encrypt_str(s.replace("q", "Q")))) == 1
    for pat in ["the quick", "over", "fox", s]:
        ep = env.encrypt_str(pat)
        assert env.decrypt_bool(key.starts_with(es, ep)) == int(s.startswith(pat)), pat
        assert env.decrypt_bool(key.ends_with(es, ep)) == int(s.endswith(pat)), pat
        assert env.decrypt_bool(key.starts_with(es, pat)) == int(s.startswith(pat)), pat
        assert env.decrypt_bool(key.ends_with(es, pat)) == int(s.endswith(pat)), pat
        assert env.decrypt_bool(key.contains_clear(es, pat)) == 1, pat
        assert env.decrypt_bool(key.eq_clear(es, pat)) == int(s == pat), pat
    assert env.decrypt_bool(key.contains_clear(es, "foxy")) == 0
    key.stats(reset=True)
    t0 = time.perf_counter()
    key.contains_clear(es, "jumps")
    dt = time.perf_counter() - t0
    st = key.stats()
    # one flag per distinct (position, nibble half, value) the 26 windows ask for, then AND per window; OR 26 -> 2 -> 1
    distinct = {(w + j, half, (ord("jumps"[j]) >> (4 * half)) & 15) for w in range(26) for j in range(5) for half in (0, 1)}
    assert st["pbs"] == len(distinct) + 26 + 2 + 1 and len(distinct) < 26 * 10
    print(f"\n[contains_clear 30/5] {dt * 1e3:.1f} ms, {st['pbs']} PBS in {st['launches']} launches")


def decrypt_str(env, blocks):
    d = env.ks.decrypt(blocks.reshape(-1, blocks.shape[-1])).reshape(-1, 4)
    assert (d < 4).all()
    return "".join(chr(int(sum(int(v) << (2 * i) for i, v in enumerate(ch)))) for ch in d)


def test_integer_ops_on_gpu(env):
    """comparisons, min / max, cmux, add / sub / neg / scalar ops on FheUint16-sized radix ciphertexts at
    the full parameter set (tests_cases_unsigned.rs default_*_test: decrypt and compare)"""
    key = env.key
    nb, mod = 8, 4**8
    rng = np.random.default_rng(21)
    for a, b in [(0, 0), (mod - 1, 1), (12345, 12345), (12345, 12346)] + [tuple(int(v) for v in rng.integers(0, mod, 2)) for _ in range(2)]:
        ea, eb = env.encrypt_radix(a, nb), env.encrypt_radix(b, nb)
        for name, f in {"gt": a > b, "ge": a >= b, "lt": a < b, "le": a <= b}.items():
            assert env.decrypt_bool(key.compare_parallelized(ea, eb, name)) == int(f), (a, b, name)
            assert env.decrypt_bool(key.compare_parallelized(ea, b, name)) == int(f), (a, b, name, "scalar")
        assert env.decrypt_radix(key.max_parallelized(ea, eb)) == max(a, b)
        assert env.decrypt_radix(key.min_parallelized(ea, eb)) == min(a, b)
        assert env.decrypt_radix(key.add_parallelized(ea, eb)) == (a + b) % mod
        assert env.decrypt_radix(key.sub_parallelized(ea, eb)) == (a - b) % mod
        assert env.decrypt_radix(key.neg_parallelized(ea)) == (-a) % mod
        assert env.decrypt_radix(key.scalar_add_parallelized(ea, b)) == (a + b) % mod
        assert env.decrypt_radix(key.scalar_mul_parallelized(ea, b % 1000)) == (a * (b % 1000)) % mod
        cond = int(a > b)
        got = key.if_then_else_parallelized(env.ks.encrypt([cond])[0], ea, eb)
        assert env.decrypt_radix(got) == (a if cond else b)


def test_string_order_and_case_on_gpu(env):
    key = env.key
    a, b = "the quick brown fox", "the quick brown fog"
    ea, eb = env.encrypt_str(a), env.encrypt_str(b)
    for x, ex in ((a, ea), (b, eb)):
        for y, ey in ((a, ea), (b, eb), (a[:9], env.encrypt_str(a[:9]))):
            assert env.decrypt_bool(key.le(ex, ey)) == int(x <= y), (x, y)
            assert env.decrypt_bool(key.ge(ex, ey)) == int(x >= y), (x, y)
            assert env.decrypt_bool(key.le(ex, y)) == int(x <= y), (x, y, "clear")
            assert env.decrypt_bool(key.ge(ex, y)) == int(x >= y), (x, y, "clear")
    s = "Hello, FHE World! [`az{@AZ]"
    es = env.encrypt_str(s)
    t0 = time.perf_counter()
    up = key.to_uppercase(es)
    dt = time.perf_counter() - t0
    assert decrypt_str(env, up) == s.upper()
    assert decrypt_str(env, key.to_lowercase(es)) == s.lower()
    assert env.decrypt_bool(key.eq_ignore_case(es, env.encrypt_str(s.swapcase()))) == 1
    assert env.decrypt_bool(key.eq_ignore_case(es, s.upper())) == 1
    assert env.decrypt_bool(key.eq_ignore_case(es, s.upper().replace("W", "V"))) == 0
    print(f"\n[to_uppercase {len(s)} chars] {dt * 1e3:.1f} ms")


def test_string_find_rfind_on_gpu(env):
    """find / rfind -> (found, 8-bit index) at the full parameter set (find.rs doc-tests and clear str.find)"""
    key = env.key
    s = "it was the best of times, it was the worst of times"
    es = env.encrypt_str(s)
    for pat in ("times", "it was", "worst", "z"):
        for ep in (env.encrypt_str(pat), pat):
            for fn, clear in ((key.find, s.find(pat)), (key.rfind, s.rfind(pat))):
                key.stats(reset=True)
                t0 = time.perf_counter()
                found, index = fn(es, ep)
                dt = time.perf_counter() - t0
                assert env.decrypt_bool(found) == int(clear >= 0), pat
                if clear >= 0:
                    assert env.decrypt_radix(index) == clear, pat
    st = key.stats()
    print(f"\n[rfind {len(s)}/1 clear] {dt * 1e3:.1f} ms, {st['pbs']} PBS in {st['launches']} launches")


def test_final_padding_on_gpu(env):
    """Padding::Final inputs at the full parameter set: eq / starts_with / contains / find / le / ge"""
    key = env.key
    pad = lambda s, n: env.encrypt_str(s + "\x00" * n)
    s = "padded fhe string"
    es = pad(s, 3)
    assert env.decrypt_bool(key.eq_padded(es, pad(s, 1))) == 1
    assert env.decrypt_bool(key.eq_padded(es, pad(s[:-1], 4))) == 0
    for pat in ("padded", "fhe", "string", "strings", "q"):
        ep = pad(pat, 2)
        assert env.decrypt_bool(key.starts_with_padded(es, ep)) == int(s.startswith(pat)), pat
        assert env.decrypt_bool(key.contains_padded(es, ep)) == int(pat in s), pat
        found, index = key.find_padded(es, ep)
        assert env.decrypt_bool(found) == int(pat in s), pat
        if pat in s:
            assert env.decrypt_radix(index) == s.find(pat), pat
    assert env.decrypt_bool(key.le(es, pad("padded fhe strinh", 1))) == 1
    assert env.decrypt_bool(key.ge(es, pad("padded fhe strin", 5))) == 1


def test_trim_and_strip_on_gpu(env):
    key = env.key

    def plain(content, length, want):
        got = decrypt_str(env, content).replace("\x00", "")
        assert got == want, (got, want)
        assert env.decrypt_radix(length) == len(want)

    s = "  \t the quick brown fox   "
    es = env.encrypt_str(s)
    plain(*key.trim_start(es), s.lstrip())
    plain(*key.trim_end(es), s.rstrip(" "))
    t0 = time.perf_counter()
    content, length = key.trim(es)
    dt = time.perf_counter() - t0
    plain(content, length, s.strip(" "))
    plain(*key.trim_char(env.encrypt_str("xxabcxx"), "x"), "abc")
    body = "the quick brown fox"
    eb = env.encrypt_str(body)
    for pat, fn, hit, want in (("the ", key.strip_prefix, 1, body[4:]), ("fox", key.strip_suffix, 1, body[:-3]),
                               ("fox", key.strip_prefix, 0, body)):
        for ep in (env.encrypt_str(pat), pat):
            found, content, length = fn(eb, ep)
            assert env.decrypt_bool(found) == hit
            plain(content, length, want)
    print(f"\n[trim {len(s)} chars] {dt * 1e3:.1f} ms")


def test_push_padding_to_end_on_gpu(env):
    """padding anywhere (encrypt_str_random_padding) at the full parameter set: compaction + operations on it"""
    key = env.key
    s = "anywhere padded string"
    padded = "\x00a\x00nywhe\x00\x00re pad\x00ded\x00 str\x00ing\x00\x00"
    assert padded.replace("\x00", "") == s
    es = env.encrypt_str(padded)
    key.stats(reset=True)
    t0 = time.perf_counter()
    out = key.push_padding_to_end(es)
    dt = time.perf_counter() - t0
    st = key.stats()
    assert decrypt_str(env, out) == s + "\x00" * (len(padded) - len(s))
    assert env.decrypt_bool(key.contains_padded(es, env.encrypt_str("pad\x00ded"), (4, 4))) == 1
    found, index = key.find_padded(es, env.encrypt_str("\x00string"), (4, 2))
    assert env.decrypt_bool(found) == 1 and env.decrypt_radix(index) == s.find("string")
    print(f"\n[push_padding_to_end {len(padded)} chars] {dt * 1e3:.1f} ms, {st['pbs']} PBS in {st['launches']} launches")


def test_string_ops_on_a_multi_bit_key(oracle):
    """the op-graph layer is key-agnostic: the same string operations over a multi-bit PBS server key
    (PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS at full size)"""
    import tfhe_rs_string_amd as T

    ks = oracle.MultiBitKeySet(oracle.param_multi_bit_message_2_carry_2_group_2(), 2, seed=31)
    p = T.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS()
    sk = T.ServerKey(T.Context(0), p, ks.bsk, ks.ksk, grouping_factor=2)
    key = T.StringServerKey(sk.handle, p)
    enc = lambda s: ks.encrypt([(ord(ch) >> (2 * i)) & 3 for ch in s for i in range(4)]).reshape(len(s), 4, -1)
    dec = lambda b: int(ks.decrypt(b.reshape(1, -1))[0])
    s = "multi-bit keys too"
    es = enc(s)
    assert dec(key.eq(es, enc(s))) == 1
    assert dec(key.eq(es, enc(s.replace("k", "K")))) == 0
    assert dec(key.contains(es, enc("bit"))) == 1
    assert dec(key.contains_clear(es, "byte")) == 0
    assert dec(key.le(es, enc("multi-bit keys twice"))) == 1
    found, index = key.find(es, "keys")
    assert dec(found) == 1 and sum(int(d % 4) << (2 * i) for i, d in enumerate(ks.decrypt(index))) == s.find("keys")


@pytest.mark.parametrize("seed", range(6))
def test_random_strings_at_full_parameters(env, seed):
    """the randomised cross-check of tests/test_host_ops_fuzz.py at PARAM_MESSAGE_2_CARRY_2_KS_PBS with the real
    noise: every leveled combination the op-graphs form (sums of 15 blocks, 4x + y packings, E + 5 Z,
    padding-bit differences) must stay inside the noise budget of the parameter set"""
    rng = np.random.default_rng(5000 + seed)
    alphabet = "ab AB.z"
    rs = lambda lo, hi: "".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), int(rng.integers(lo, hi + 1))))
    key, b = env.key, env.decrypt_bool
    s = rs(20, 40)
    a = int(rng.integers(0, len(s) - 4))
    pat = s[a:a + int(rng.integers(1, 5))] if seed % 2 == 0 else rs(1, 4)
    other = s if seed == 1 else rs(20, 40)
    es, ep, eo = env.encrypt_str(s), env.encrypt_str(pat), env.encrypt_str(other)
    ctx = (s, pat, other)
    assert b(key.eq(es, eo)) == int(s == other), ctx
    assert b(key.le(es, eo)) == int(s <= other), ctx
    assert b(key.ge(es, other)) == int(s >= other), ctx
    assert b(key.eq_ignore_case(es, env.encrypt_str(s.swapcase()))) == 1, ctx
    for enc_pat in (ep, pat):
        assert b(key.starts_with(es, enc_pat)) == int(s.startswith(pat)), ctx
        assert b(key.ends_with(es, enc_pat)) == int(s.endswith(pat)), ctx
        for fn, want in ((key.find, s.find(pat)), (key.rfind, s.rfind(pat))):
            found, index = fn(es, enc_pat)
            assert b(found) == int(want >= 0), ctx
            if want >= 0:
                assert env.decrypt_radix(index) == want, ctx
    assert b(key.contains(es, ep)) == int(pat in s), ctx
    assert decrypt_str(env, key.to_uppercase(es)) == s.upper(), ctx
    for (content, length), want in ((key.trim_start(es), s.lstrip()), (key.trim(es), s.strip(" ")), (key.trim_char(es, "a"), s.strip("a"))):
        assert decrypt_str(env, content).replace("\x00", "") == want, ctx
        assert env.decrypt_radix(length) == len(want), ctx
    base = s[:24]
    chars = list(base)
    for _ in range(5):
        chars.insert(int(rng.integers(0, len(chars) + 1)), "\x00")
    padded = "".join(chars)
    ez = env.encrypt_str(padded)
    assert decrypt_str(env, key.push_padding_to_end(ez)) == base + "\x00" * 5, (ctx, padded)
    assert env.decrypt_radix(key.len(ez)) == len(base)
    assert b(key.contains_padded(ez, ep, (4, 0))) == int(pat in base), (ctx, padded)
    nb, mod = 8, 4**8
    x, y = (int(v) for v in rng.integers(0, mod, 2))
    ex, ey = env.encrypt_radix(x, nb), env.encrypt_radix(y, nb)
    assert env.decrypt_radix(key.mul_parallelized(ex, ey)) == (x * y) % mod
    assert env.decrypt_radix(key.sub_parallelized(ex, ey)) == (x - y) % mod
    assert b(key.compare_parallelized(ex, ey, "gt")) == int(x > y)


def test_batch_mode_on_gpu(env):
    """a column of strings against one pattern in one batch: the union of the op-graphs runs level by level"""
    key = env.key
    words = ["encrypted", "strings", "on", "the", "mi355x", "encrypted!", "", "Encrypted"] * 4
    enc = [env.encrypt_str(w) for w in words]
    target = env.encrypt_str("encrypted")
    key.stats(reset=True)
    t0 = time.perf_counter()
    with key.batch():
        eqs = [key.eq(e, target) for e in enc]
        has = [key.contains_clear(e, "ypt") for e in enc if e.shape[0] >= 3]
        les = [key.le(e, target) for e in enc]
    dt = time.perf_counter() - t0
    st = key.stats()
    assert [env.decrypt_bool(o) for o in eqs] == [int(w == "encrypted") for w in words]
    assert [env.decrypt_bool(o) for o in has] == [int("ypt" in w) for w in words if len(w) >= 3]
    assert [env.decrypt_bool(o) for o in les] == [int(w <= "encrypted") for w in words]
    print(f"\n[batch of {len(eqs) + len(has) + len(les)} string operations] {dt * 1e3:.1f} ms, {st['pbs']} PBS in {st['launches']} launches")


def test_big_first_levels_on_gpu(env, monkeypatch):
    """the two ways a big first level reaches the device: finished rows in pieces over the transfer stream while the
    earlier rounds run (a column of 24 `eq` on 64-char strings: 3 072 rows formed by the host), and the caller's arrays
    uploaded once with the rows formed on the device (`contains` with a 160-char haystack: every block is read by 6
    windows).  Decrypt-checked, and bit for bit what the host-staged executor returns."""
    import tfhe_rs_string_amd as T

    key = env.key
    rng = np.random.default_rng(17)
    base = "".join(rng.choice(list("abcdefgh"), size=64))
    col = [base if i % 3 == 0 else base[:i] + "X" + base[i + 1:] for i in range(24)]
    enc = [env.encrypt_str(w) for w in col]
    target = env.encrypt_str(base)
    hay = "".join(rng.choice(list("ab"), size=160))
    hay = hay[:100] + "abbaab" + hay[106:]
    e_hay, e_pat, e_miss = env.encrypt_str(hay), env.encrypt_str("abbaab"), env.encrypt_str("abcabc")

    def run(k):
        with k.batch():
            eqs = [k.eq(e, target) for e in enc]
        return eqs + [k.contains(e_hay, e_pat), k.contains(e_hay, e_miss)]

    got = run(key)
    assert [env.decrypt_bool(o) for o in got[:24]] == [int(w == base) for w in col]
    assert env.decrypt_bool(got[24]) == 1 and env.decrypt_bool(got[25]) == int("abcabc" in hay)
    monkeypatch.setenv("TFHE_HIP_HOST_STAGING", "1")
    other = T.StringServerKey(env.sk.handle, env.p)
    monkeypatch.delenv("TFHE_HIP_HOST_STAGING")
    for a, b in zip(got, run(other)):
        assert np.array_equal(a, b)


def test_string_ops_on_the_reference_fast_test_parameter_set(oracle):
    """`gen_keys_test()` of the reference string layer (tfhe/examples/fhe_strings/ciphertext.rs:94-105) uses
    PARAM_MESSAGE_2_CARRY_2_TEST (n = 1, N = 256): the same op-graphs over the generic kernel, decrypt and compare"""
    import tfhe_rs_string_amd as T

    ks = oracle.KeySet(oracle.param_message_2_carry_2_test(), seed=5)
    p = T.PARAM_MESSAGE_2_CARRY_2_TEST()
    sk = T.ServerKey(T.Context(0), p, ks.bsk, ks.ksk)
    key = T.StringServerKey(sk.handle, p)
    enc = lambda s: ks.encrypt([(ord(ch) >> (2 * i)) & 3 for ch in s for i in range(4)]).reshape(len(s), 4, -1)
    dec_bool = lambda b: int(ks.decrypt(b.reshape(1, -1))[0])
    dec_radix = lambda blocks: sum(int(d % 4) << (2 * i) for i, d in enumerate(ks.decrypt(blocks)))
    a, b = enc("hello world"), enc("hello w0rld")
    assert dec_bool(key.eq(a, enc("hello world"))) == 1 and dec_bool(key.eq(a, b)) == 0
    assert dec_bool(key.contains(a, enc("o w"))) == 1 and dec_bool(key.contains_clear(a, "xyz")) == 0
    found, index = key.find(a, "world")
    assert dec_bool(found) == 1 and dec_radix(index) == 6
    up = key.to_uppercase(a)
    d = ks.decrypt(up.reshape(-1, up.shape[-1])).reshape(-1, 4)
    assert "".join(chr(int(sum(int(v) << (2 * i) for i, v in enumerate(ch)))) for ch in d) == "HELLO WORLD"
