"""CPU coverage of host_split.hpp (concatenation, repeat, substring, nth, the split family, replace) through the
C-ABI, on the oracle-backed test backend of test_host_ops_cpu.py.  As in the reference's tests
(tfhe/examples/fhe_strings/test_generating_macros.rs:23-33, server_key/{split,rsplit,rsplitn,split_terminator,
split_inclusive,split_ascii_whitespace,replace,substring,repeat,add}.rs `mod tests`), the decrypted result must equal
what Rust's `str` methods give on the clear strings; the cases are the reference's own test inputs plus seeded
random ones.  The `std_*` helpers restate the documented semantics of the Rust methods."""
import os

import numpy as np
import pytest

from tests.test_host_ops_cpu import backend, decrypt_str, env  # noqa: F401  (fixtures)


# ---- what Rust's str methods return ------------------------------------------------------------------
def std_split(s, pat):
    if pat == "":
        return [""] + list(s) + [""]
    return s.split(pat)


def std_rsplit(s, pat):
    if pat == "":
        return [""] + list(reversed(s)) + [""]
    return list(reversed(s.rsplit(pat)))


def std_splitn(s, n, pat):
    if pat == "":  # matches at every char boundary: "abc".splitn(3, "") == ["", "a", "bc"]
        full = std_split(s, "")
        return [] if n == 0 else (full if n >= len(full) else full[:n - 1] + ["".join(full[n - 1:])])
    return [] if n == 0 else s.split(pat, n - 1)


def std_rsplitn(s, n, pat):
    if pat == "":  # "abc".rsplitn(3, "") == ["", "c", "ab"]
        full = std_rsplit(s, "")
        return [] if n == 0 else (full if n >= len(full) else full[:n - 1] + [s[:len(s) - (n - 2)] if n >= 2 else s])
    return [] if n == 0 else list(reversed(s.rsplit(pat, n - 1)))


def std_split_terminator(s, pat):
    if pat == "":  # the trailing "" is dropped: "abc" -> ["", "a", "b", "c"]
        return std_split(s, "")[:-1]
    parts = s.split(pat)
    return parts[:-1] if parts[-1] == "" else parts


def std_rsplit_terminator(s, pat):
    if pat == "":  # the part yielded first from the back (the trailing "") is dropped: "abc" -> ["c", "b", "a", ""]
        return std_rsplit(s, "")[1:]
    parts = s.rsplit(pat)
    if parts[-1] == "":
        parts = parts[:-1]
    return list(reversed(parts))


def std_split_inclusive(s, pat):
    if pat == "":  # every piece ends with a (zero-length) match; the empty remainder is dropped: "abc" -> ["", "a", "b", "c"]
        return std_split(s, "")[:-1]
    parts = s.split(pat)
    out = [p + pat for p in parts[:-1]]
    return out + ([parts[-1]] if parts[-1] != "" else [])


def std_split_ascii_whitespace(s):
    return [w for w in s.replace("\t", " ").replace("\n", " ").replace("\r", " ").replace("\x0c", " ").split(" ") if w]


def test_std_helpers_match_documented_rust_results():
    # examples from the documentation of core::str
    assert std_split("Mary had a little lamb", " ") == ["Mary", "had", "a", "little", "lamb"]
    assert std_split("", "X") == [""]
    assert std_split("lionXXtigerXleopard", "X") == ["lion", "", "tiger", "leopard"]
    assert std_split("||||a||b|c", "|") == ["", "", "", "", "a", "", "b", "c"]
    assert std_split("rust", "") == ["", "r", "u", "s", "t", ""]
    assert std_rsplit("lion::tiger::leopard", "::") == ["leopard", "tiger", "lion"]
    assert std_splitn("Mary had a little lambda", 3, " ") == ["Mary", "had", "a little lambda"]
    assert std_splitn("lionXXtigerXleopard", 3, "X") == ["lion", "", "tigerXleopard"]
    assert std_splitn("abcXdef", 1, "X") == ["abcXdef"]
    assert std_rsplitn("Mary had a little lamb", 3, " ") == ["lamb", "little", "Mary had a"]
    assert std_rsplitn("lion::tiger::leopard", 2, "::") == ["leopard", "lion::tiger"]
    assert std_split_terminator("A.B.", ".") == ["A", "B"]
    assert std_split_terminator("A..B..", ".") == ["A", "", "B", ""]
    assert std_rsplit_terminator("A.B.", ".") == ["B", "A"]
    assert std_rsplit_terminator("A..B..", ".") == ["", "B", "", "A"]
    assert std_split_inclusive("Mary had a little lamb\nlittle lamb\nlittle lamb.", "\n") == \
        ["Mary had a little lamb\n", "little lamb\n", "little lamb."]
    assert std_split_inclusive("a\nb\n", "\n") == ["a\n", "b\n"]
    assert std_split_ascii_whitespace("Mary had\ta little  \n lamb") == ["Mary", "had", "a", "little", "lamb"]
    assert std_split_ascii_whitespace("  ") == []
    # the empty pattern (core::str documentation and the iterator's definition: a match at every char boundary)
    assert std_splitn("abc", 3, "") == ["", "a", "bc"] and std_splitn("abc", 1, "") == ["abc"] and std_splitn("abc", 9, "") == ["", "a", "b", "c", ""]
    assert std_rsplitn("abc", 3, "") == ["", "c", "ab"] and std_rsplitn("abc", 2, "") == ["", "abc"]
    assert std_split_terminator("abc", "") == ["", "a", "b", "c"] and std_split_terminator("", "") == [""]
    assert std_rsplit_terminator("abc", "") == ["c", "b", "a", ""] and std_rsplit_terminator("", "") == [""]
    assert std_split_inclusive("abc", "") == ["", "a", "b", "c"]


def split_result(env, fhe, unpad=True):
    parts, number = fhe
    n = env.decrypt_radix(number)
    assert n <= len(parts), (n, len(parts))
    out = [decrypt_str(env, p) for p in parts[:n]]
    return [p.replace("\x00", "") for p in out] if unpad else out


def pattern_forms(env, pat):
    yield pat
    if pat:
        yield env.encrypt_str(pat)


# the reference's split / rsplit / terminator / inclusive test inputs (split.rs, rsplit.rs:883-919, split_terminator.rs,
# split_inclusive.rs `mod tests`) and a few more
SPLIT_CASES = [("", "e"), ("", "ab"), ("acbc", "c"), ("acbccbcbcbc", "cbc"), ("aaaaa", "a"), ("axbx", "x"), ("ab", "ab"),
               ("abab", "ab"), ("cbca", "c"), ("acbc", "bc"), ("a", "a"), ("cccc", "c"), ("cabd", "a"), ("acb", "c"),
               ("cacb", "c"), ("aaaaa", "aa"), ("abcab", "abcabc"), ("a.b.c", "."), ("hello, world, x,", ", ")]


@pytest.mark.parametrize("kind", ["split", "rsplit", "split_terminator", "rsplit_terminator", "split_inclusive"])
def test_split_family_reference_cases(env, kind):
    key = env.key
    std = {"split": std_split, "rsplit": std_rsplit, "split_terminator": std_split_terminator,
           "rsplit_terminator": std_rsplit_terminator, "split_inclusive": std_split_inclusive}[kind]
    for s, pat in SPLIT_CASES:
        es = env.encrypt_str(s)
        for form in pattern_forms(env, pat):
            got = split_result(env, getattr(key, kind)(es, form))
            assert got == std(s, pat), (kind, s, pat, isinstance(form, str), got)


def test_split_parts_keep_positions_and_empty_pattern(env):
    key = env.key
    s = "a,bc,,d"
    parts, number = key.split(env.encrypt_str(s), ",")
    assert env.decrypt_radix(number) == 4
    raw = [decrypt_str(env, p) for p in parts]
    assert raw[:4] == ["a" + "\x00" * 6, "\x00\x00bc\x00\x00\x00", "\x00" * 7, "\x00" * 6 + "d"]
    assert all(r == "\x00" * 7 for r in raw[4:])  # parts beyond number_parts are empty as well
    for s in ["", "abc", "aczb"]:
        assert split_result(env, key.split(env.encrypt_str(s), "")) == std_split(s, "")
        assert split_result(env, key.rsplit(env.encrypt_str(s), "")) == std_rsplit(s, "")
    assert split_result(env, key.split_terminator(env.encrypt_str("abc"), "")) == ["", "a", "b", "c"]  # test_empty_pattern_in_every_kind


def test_splitn_rsplitn_clear_and_encrypted_n(env):
    key = env.key
    # rsplitn.rs:729-760 / split.rs test inputs
    cases = [("aaaaa", "aa"), ("ccc", "c"), ("acbc", "c"), ("abc", "ab"), ("a.b.c.d", "."), ("", "ab")]
    for s, pat in cases:
        es = env.encrypt_str(s)
        for n in [0, 1, 2, 3, 5]:
            for form in pattern_forms(env, pat):
                assert split_result(env, key.splitn(es, n, form)) == std_splitn(s, n, pat), (s, pat, n)
                assert split_result(env, key.rsplitn(es, n, form)) == std_rsplitn(s, n, pat), (s, pat, n)
            en = env.encrypt_radix(n, 4)
            assert split_result(env, key.splitn(es, en, pat)) == std_splitn(s, n, pat), (s, pat, n, "enc")
            assert split_result(env, key.rsplitn(es, en, pat)) == std_rsplitn(s, n, pat), (s, pat, n, "enc")


def test_split_ascii_whitespace_and_split_once(env):
    key = env.key
    for s in ["", " ", "a", "Mary had\ta little  \n lamb", "  lead and trail  ", "x  y", "\t\tz"]:
        assert split_result(env, key.split_ascii_whitespace(env.encrypt_str(s))) == std_split_ascii_whitespace(s), repr(s)
    for s, pat in [("cfg=a=b", "="), ("cfg", "="), ("=x", "="), ("x=", "="), ("a::b::c", "::"), ("aaa", "aa")]:
        es = env.encrypt_str(s)
        for form in pattern_forms(env, pat):
            for rev in (False, True):
                found, before, after = key.split_once(es, form, reverse=rev)
                want = (s.rsplit(pat, 1) if rev else s.split(pat, 1)) if pat in s else None
                assert env.decrypt_bool(found) == int(want is not None), (s, pat, rev)
                if want is not None:
                    got = [decrypt_str(env, before).replace("\x00", ""), decrypt_str(env, after).replace("\x00", "")]
                    assert got == want, (s, pat, rev, got)


def random_padding(rng, s, n):
    chars = list(s)
    for _ in range(n):
        chars.insert(int(rng.integers(0, len(chars) + 1)), "\x00")
    return "".join(chars)


def test_split_family_on_padded_strings(env):
    key = env.key
    rng = np.random.default_rng(5)
    for s, pat in [("a,b,", ","), ("", ","), ("ab", "ab"), ("x--y--", "--"), ("aaa", "a")]:
        for padded, code in [(s + "\x00\x00\x00", 1), (random_padding(rng, s, 3), 4)]:
            es = env.encrypt_str(padded)
            assert split_result(env, key.split(es, pat, padding=code)) == std_split(s, pat), (padded, pat)
            assert split_result(env, key.rsplit(es, pat, padding=code)) == std_rsplit(s, pat), (padded, pat)
            assert split_result(env, key.split_terminator(es, pat, padding=code)) == std_split_terminator(s, pat), (padded, pat)
            assert split_result(env, key.rsplit_terminator(es, pat, padding=code)) == std_rsplit_terminator(s, pat), (padded, pat)
            assert split_result(env, key.split_inclusive(es, pat, padding=code)) == std_split_inclusive(s, pat), (padded, pat)
            assert split_result(env, key.splitn(es, 2, pat, padding=code)) == std_splitn(s, 2, pat), (padded, pat)
    for s in ["a b", "", " a  bc "]:
        es = env.encrypt_str(s + "\x00\x00")
        assert split_result(env, key.split_ascii_whitespace(es, padding=1)) == std_split_ascii_whitespace(s)
        assert split_result(env, key.split(es, "", padding=1)) == std_split(s, "")
        assert split_result(env, key.rsplit(es, "", padding=1)) == std_rsplit(s, "")


def test_split_fuzz_against_std(env):
    key = env.key
    rng = np.random.default_rng(11)
    kinds = [("split", std_split), ("rsplit", std_rsplit), ("split_terminator", std_split_terminator),
             ("rsplit_terminator", std_rsplit_terminator), ("split_inclusive", std_split_inclusive)]
    for it in range(int(os.environ.get("TFHE_HIP_FUZZ_SEEDS", "10"))):
        L = int(rng.integers(1, 22))
        s = "".join(rng.choice(list("abc"), size=L))
        pat = "".join(rng.choice(list("abc"), size=int(rng.integers(1, 4))))
        es = env.encrypt_str(s)
        form = pat if it % 2 else env.encrypt_str(pat)
        name, std = kinds[it % len(kinds)]
        assert split_result(env, getattr(key, name)(es, form)) == std(s, pat), (name, s, pat)
        n = int(rng.integers(0, 5))
        assert split_result(env, key.splitn(es, n, form)) == std_splitn(s, n, pat), (s, pat, n)
        assert decrypt_str(env, key.replace(es, form, "xy")[0]).replace("\x00", "") == s.replace(pat, "xy"), (s, pat)


def test_long_string_uses_both_count_nibbles(env):
    """more than 16 parts and more than 12 positions: the second chunk of the prefix count and the high nibble flags"""
    key = env.key
    s = "a,b,,c,d,e,f,g,h,i,j,k,l,m,n,o,p,q,r,s"
    assert len(s.split(",")) > 16
    assert split_result(env, key.split(env.encrypt_str(s), ",")) == s.split(",")
    assert split_result(env, key.rsplitn(env.encrypt_str(s), 18, ",")) == std_rsplitn(s, 18, ",")
    for n in (5, 17):  # the remainder part: "count >= n - 1" from the flags of both nibbles
        assert split_result(env, key.splitn(env.encrypt_str(s), n, ",")) == std_splitn(s, n, ","), n
    s2 = "ab--cd--ef--gh--ij--kl--mn--op--qr"
    assert split_result(env, key.splitn(env.encrypt_str(s2), 3, "--")) == std_splitn(s2, 3, "--")
    assert split_result(env, key.rsplitn(env.encrypt_str(s2), 4, env.encrypt_str("--"))) == std_rsplitn(s2, 4, "--")


def test_speculative_scan_of_self_overlapping_patterns(env):
    """more than 12 candidate positions: the selection scan runs chunk-wise on speculated states"""
    key = env.key
    cases = [("a" * 30, "aa"), ("a" * 29, "aaa"), ("ab" * 12 + "a", "abab"), ("x--y----z-----w--", "--"),
             ("aabaabaabaabaabaabaa", "aabaa"), ("bbbbbbbbbbbbbabbbbbbbbbbbbbbb", "bbbb")]
    for s, pat in cases:
        es = env.encrypt_str(s)
        for form in pattern_forms(env, pat):
            assert split_result(env, key.split(es, form)) == std_split(s, pat), (s, pat)
        assert split_result(env, key.rsplit(es, pat)) == std_rsplit(s, pat), (s, pat)
        assert decrypt_str(env, key.replace(es, pat, "Q")[0]).replace("\x00", "") == s.replace(pat, "Q"), (s, pat)
    key.stats(reset=True)
    key.split(env.encrypt_str("a" * 50), "aa")
    assert key.stats()["launches"] <= 30  # 49 candidate positions: not one level per position


def test_replace_and_replacen(env):
    key = env.key
    # replace.rs `mod tests` inputs and more: equal / shorter / longer / empty replacement, overlapping occurrences
    cases = [("abab", "ab", "cd"), ("aaaa", "aa", "b"), ("hello world", "o", "0"), ("abc", "x", "y"), ("a.b.c", ".", "::"),
             ("ababab", "ab", ""), ("xaax", "a", "bcd"), ("ab", "abc", "x"), ("", "a", "b")]
    for s, old, new in cases:
        es = env.encrypt_str(s)
        for f_old in pattern_forms(env, old):
            for f_new in pattern_forms(env, new):
                content, pad = key.replace(es, f_old, f_new)
                assert decrypt_str(env, content).replace("\x00", "") == s.replace(old, new), (s, old, new)
        for n in [0, 1, 2, 7]:
            content, pad = key.replacen(es, old, new, n)
            assert decrypt_str(env, content).replace("\x00", "") == s.replace(old, new, n), (s, old, new, n)
            content, pad = key.replacen(es, old, new, env.encrypt_radix(n, 4))
            assert decrypt_str(env, content).replace("\x00", "") == s.replace(old, new, n), (s, old, new, n, "enc")
    content, pad = key.replace(env.encrypt_str("a\x00b\x00ab"), "ab", "Z", padding=4)
    assert decrypt_str(env, content).replace("\x00", "") == "ZZ" and pad == 4


def test_add_repeat_substring_nth(env):
    key = env.key
    a, b = env.encrypt_str("ab"), env.encrypt_str("bc")
    content, pad = key.add(a, b)  # add.rs doctest
    assert decrypt_str(env, content) == "abbc" and pad == 0
    assert key.add(a, env.encrypt_str("c\x00"), padding=(0, 1))[1] == 1
    assert key.add(env.encrypt_str("\x00a"), b, padding=(2, 0))[1] == 2
    assert key.add(env.encrypt_str("a\x00"), b, padding=(1, 0))[1] == 4
    content, pad = key.repeat(env.encrypt_str("ab\x00"), 3, padding=1)  # repeat.rs:10-16
    assert decrypt_str(env, content).replace("\x00", "") == "ababab" and pad == 4
    assert key.repeat(a, 0)[0].shape[0] == 0
    for n in range(4):  # repeat.rs:36-42
        content, pad = key.repeat(a, env.encrypt_radix(n, 4), n_max=3)
        assert decrypt_str(env, content) == "ab" * n + "\x00" * (2 * (3 - n)) and pad == 1
    s = "hello world"
    es = env.encrypt_str(s)
    ok, content, pad = key.substring(es, 2, 7)
    assert env.decrypt_bool(ok) == 1 and decrypt_str(env, content) == s[2:7]
    ok, content, pad = key.substring(env.encrypt_str(s + "\x00\x00"), 9, 12, padding=1)
    assert env.decrypt_bool(ok) == 0
    ok, content, pad = key.substring(env.encrypt_str(s + "\x00\x00"), 9, 11, padding=1)
    assert env.decrypt_bool(ok) == 1 and decrypt_str(env, content) == "ld"
    with pytest.raises(ValueError):
        key.substring(es, 3, 40)
    for start, end in [(0, 11), (3, 5), (4, 4), (6, 2), (2, 14)]:
        ok, content, pad = key.substring(es, env.encrypt_radix(start, 4), env.encrypt_radix(end, 4))
        assert env.decrypt_bool(ok) == int(start <= end <= len(s)), (start, end)
        assert decrypt_str(env, content).replace("\x00", "") == s[start:end] and pad == 3, (start, end)
    for n in [0, 4, 10, 11, 200]:
        want = ord(s[n]) if n < len(s) else 0
        assert env.decrypt_radix(key.nth(es, n)) == want
        assert env.decrypt_radix(key.nth(es, env.encrypt_radix(n, 4))) == want
    assert env.decrypt_radix(key.nth(env.encrypt_str("\x00a\x00b"), 1, padding=4)) == ord("b")
    for c in (0, 1):
        content, pad = key.cmux_empty_string(env.ks.encrypt([c])[0], a)
        assert decrypt_str(env, content).replace("\x00", "") == ("ab" if c else "") and pad == 1


def test_scan_only_where_the_pattern_overlaps_itself(env):
    """launch counts: a pattern without a border needs no selection scan, "aa" needs one step per position"""
    key = env.key
    es = env.encrypt_str("abcabcabcabc")
    key.stats(reset=True)
    key.split(es, "bc")
    free = key.stats(reset=True)
    key.split(es, "bb")
    scan = key.stats(reset=True)
    assert scan["launches"] >= free["launches"] + 8, (free, scan)


def test_split_family_in_batch_mode_and_errors(env):
    """batch mode: several splits / replaces staged together share their launches and give the same results; argument
    errors come back as non-zero returns with a message (no partial writes, no crash)"""
    import ctypes as C

    key = env.key
    strings = ["a,b,,c", "x--y--z", "no separators", ",,"]
    enc = [env.encrypt_str(s) for s in strings]
    key.stats(reset=True)
    one_by_one = [split_result(env, key.split(e, ",")) for e in enc]
    single = key.stats(reset=True)
    with key.batch():
        staged = [key.split(e, ",") for e in enc]
        replaced = [key.replace(e, ",", ";") for e in enc]
    batched = key.stats(reset=True)
    assert [split_result(env, r) for r in staged] == one_by_one == [s.split(",") for s in strings]
    assert [decrypt_str(env, c).replace("\x00", "") for c, _ in replaced] == [s.replace(",", ";") for s in strings]
    assert batched["launches"] < single["launches"]
    # errors
    L, h = key._lib, key._h
    es = env.encrypt_str("a,b")
    ptr = es.ctypes.data_as(C.c_void_p)
    parts = np.zeros((1, 3, 4, env.params.big_size), dtype=np.uint64)
    number = np.zeros((4, env.params.big_size), dtype=np.uint64)
    got = C.c_size_t()
    args = (parts.ctypes.data_as(C.c_void_p), 1, number.ctypes.data_as(C.c_void_p), C.byref(got))
    assert L.tfhe_hip_string_split(h, ptr, 3, 0, None, b",", 1, 0, 0, 0, None, *args) != 0  # room for 1 part, 4 needed
    assert b"too small" in L.tfhe_hip_last_error()
    assert not parts.any()
    big = np.zeros((4, 3, 4, env.params.big_size), dtype=np.uint64)
    args = (big.ctypes.data_as(C.c_void_p), 4, number.ctypes.data_as(C.c_void_p), C.byref(got))
    assert L.tfhe_hip_string_split(h, ptr, 3, 0, ptr, b",", 1, 0, 0, 0, None, *args) != 0  # encrypted AND clear pattern
    assert L.tfhe_hip_string_split(h, ptr, 3, 0, None, b",", 1, 0, 42, 0, None, *args) != 0  # unknown kind
    assert L.tfhe_hip_string_split(h, ptr, 3, 9, None, b",", 1, 0, 0, 0, None, *args) != 0  # unknown padding
    assert L.tfhe_hip_string_split(h, ptr, 3, 0, None, b"\xff", 1, 0, 0, 0, None, *args) != 0  # not ASCII
    assert L.tfhe_hip_string_split(h, ptr, 3, 0, None, b",", 1, 0, 0, 0, None, *args) == 0 and got.value == 4
    with pytest.raises(ValueError):
        key.splitn(es, env.encrypt_radix(2, 2), ",")  # encrypted n must have 4 blocks
    with pytest.raises(env.T.TfheHipError):
        key.split(env.encrypt_str("a" * 255), ",")  # counts are 8-bit


def test_split_family_with_padded_encrypted_patterns(env):
    """the reference's `<string>_padding_2_<pattern>_padding_2` cases (test_generating_macros.rs:397-404): the pattern is
    `encrypt_str_random_padding(pattern, 2)`, its length is encrypted"""
    key = env.key
    rng = np.random.default_rng(23)
    kinds = {"split": std_split, "rsplit": std_rsplit, "split_terminator": std_split_terminator,
             "rsplit_terminator": std_rsplit_terminator, "split_inclusive": std_split_inclusive}
    cases = [c for c in SPLIT_CASES if c[1]] + [("aXbXXc", "X"), ("ab-ab--ab", "--")]
    for it, (s, pat) in enumerate(cases):
        ep = env.encrypt_str(random_padding(rng, pat, 2))
        for padded, code in [(s, 0), (random_padding(rng, s, 2), 4)]:
            es = env.encrypt_str(padded)
            for kind, std in list(kinds.items())[it % 2::2] if it > 4 else kinds.items():
                got = split_result(env, getattr(key, kind)(es, ep, padding=(code, 4)))
                assert got == std(s, pat), (kind, padded, pat, got)
        es = env.encrypt_str(s)
        for n in (0, 1, 2, 3):
            assert split_result(env, key.splitn(es, n, ep, padding=(0, 4))) == std_splitn(s, n, pat), (s, pat, n)
        assert split_result(env, key.rsplitn(es, env.encrypt_radix(2, 4), ep, padding=(0, 4))) == std_rsplitn(s, 2, pat), (s, pat)
        for new in ("", "Z", "long"):
            content, pad = key.replace(es, ep, new, padding=(0, 4))
            assert decrypt_str(env, content).replace("\x00", "") == s.replace(pat, new), (s, pat, new)
        content, pad = key.replacen(es, ep, env.encrypt_str("q\x00r"), 1, padding=(0, 4))
        assert decrypt_str(env, content).replace("\x00", "") == s.replace(pat, "qr", 1), (s, pat)
        found, before, after = key.split_once(es, ep, padding=(0, 4))
        assert env.decrypt_bool(found) == int(pat in s)
        if pat in s:
            assert [decrypt_str(env, before).replace("\x00", ""), decrypt_str(env, after).replace("\x00", "")] == s.split(pat, 1)
    # a pattern that is fully padded up front (Final padding given as such) and one without any zero
    es = env.encrypt_str("a--b")
    assert split_result(env, key.split(es, env.encrypt_str("--\x00\x00\x00"), padding=(0, 1))) == ["a", "b"]
    assert split_result(env, key.split(es, env.encrypt_str("--"), padding=(0, 1))) == ["a", "b"]


def test_encrypted_char_trims_and_padding_helpers(env):
    """trim_*_encrypted (trim.rs:135-200), push_padding_to_start and pop_first / pop_last_non_zero_char (change_padding.rs)"""
    key = env.key
    s = "xxhixx"
    ex = env.encrypt_str("x")[0]
    for left, right, want in [(True, True, "hi"), (True, False, "hixx"), (False, True, "xxhi")]:
        content, length = key.trim_encrypted_char(env.encrypt_str(s), ex, left, right)
        assert decrypt_str(env, content).replace("\x00", "") == want and env.decrypt_radix(length) == len(want)
    assert decrypt_str(env, key.push_padding_to_start(env.encrypt_str("a\x00b\x00\x00c"))) == "\x00\x00\x00abc"
    for padded in ["\x00ab\x00c\x00", "abc", "\x00\x00", ""]:
        real = padded.replace("\x00", "")
        for last in (False, True):
            ch, content = key.pop_non_zero_char(env.encrypt_str(padded), last=last)
            assert env.decrypt_radix(ch) == (ord(real[-1 if last else 0]) if real else 0), (padded, last)
            assert decrypt_str(env, content).replace("\x00", "") == (real[:-1] if last else real[1:]), (padded, last)
            assert len(decrypt_str(env, content)) == len(padded)


def test_encrypted_pattern_that_is_only_padding_splits_like_the_empty_pattern(env):
    """split / rsplit with an encrypted pattern whose emptiness is hidden behind its padding (the reference's
    `*_allow_empty_pattern` functions, split.rs:23-31): "", every char, \"\" -- and the same call still splits by a real
    pattern"""
    key = env.key
    for s, padded, code in [("abc", "abc", 0), ("ab", "ab\x00\x00", 1), ("", "\x00", 1), ("xyz", "x\x00y\x00z", 4)]:
        es = env.encrypt_str(padded)
        empty = env.encrypt_str("\x00\x00")
        assert split_result(env, key.split(es, empty, padding=(code, 1))) == std_split(s, ""), (padded,)
        assert split_result(env, key.rsplit(es, empty, padding=(code, 1))) == std_rsplit(s, ""), (padded,)
    es = env.encrypt_str("a,b")
    assert split_result(env, key.split(es, env.encrypt_str(",\x00"), padding=(0, 1))) == ["a", "b"]


def test_executors_and_schedulers_agree_bit_for_bit(env, monkeypatch):
    """resident vs host-staged levels, round-filling vs as-soon-as-possible scheduling: same bits (every bootstrap sees the
    same input row and the backend is deterministic)"""
    s = env.encrypt_str(" alpha, beta,, gamma ")
    pat = env.encrypt_str(", ")
    holes = env.encrypt_str("a\x00b\x00c")

    def run(key):
        parts, number = key.split(s, pat)
        content, length = key.trim(s)
        found, index = key.find(s, pat)
        return [parts, number, content, length, found, index, key.replace(s, ", ", "--")[0], key.push_padding_to_end(holes)]

    want = run(env.key)
    for staging, asap in [("1", None), (None, "1"), ("1", "1")]:
        for name, v in (("TFHE_HIP_HOST_STAGING", staging), ("TFHE_HIP_HOST_ASAP", asap)):
            if v is None:
                monkeypatch.delenv(name, raising=False)
            else:
                monkeypatch.setenv(name, v)
        other = env.T.StringServerKey(env.sk, env.params, lib=env.L)
        for a, b in zip(want, run(other)):
            assert np.array_equal(a, b), (staging, asap)
    monkeypatch.delenv("TFHE_HIP_HOST_STAGING", raising=False)
    monkeypatch.delenv("TFHE_HIP_HOST_ASAP", raising=False)


def test_empty_pattern_in_every_kind(env):
    """The empty pattern outside split / rsplit (the reference's `*_empty_pattern` and `*_allow_empty_pattern` functions:
    split.rs:201-345, 491-720, rsplit.rs:169-230, 415-700, rsplitn.rs:112-290, 367-520, split_terminator.rs:27-120, 266-330,
    split_inclusive.rs:14-28, 82-140, find.rs:321-345, 435-470): clear "", an encrypted pattern of length 0 and an encrypted
    pattern that is padding only must all behave like Rust's str methods with "" -- and a padded pattern that is NOT empty
    still splits normally through the same call."""
    key = env.key
    for s, code in (("abc", 0), ("", 0), ("ab\x00\x00", 1), ("\x00\x00", 1), ("hello", 0)):
        clear = s.replace("\x00", "")
        es = env.encrypt_str(s)
        hidden = env.encrypt_str("\x00\x00")
        forms = [("", code), (hidden, (code, 1))]
        for pat, pad in forms:
            tag = (s, isinstance(pat, str))
            assert split_result(env, key.split_terminator(es, pat, padding=pad)) == std_split_terminator(clear, ""), tag
            assert split_result(env, key.rsplit_terminator(es, pat, padding=pad)) == std_rsplit_terminator(clear, ""), tag
            assert split_result(env, key.split_inclusive(es, pat, padding=pad)) == std_split_inclusive(clear, ""), tag
            for n in (0, 1, 2, 3, 4, 9):
                assert split_result(env, key.splitn(es, n, pat, padding=pad)) == std_splitn(clear, n, ""), tag + (n,)
                assert split_result(env, key.rsplitn(es, n, pat, padding=pad)) == std_rsplitn(clear, n, ""), tag + (n,)
            found, before, after = key.split_once(es, pat, padding=pad)
            assert env.decrypt_bool(found) == 1 and decrypt_str(env, before).replace("\x00", "") == "" and \
                decrypt_str(env, after).replace("\x00", "") == clear, tag
            found, before, after = key.rsplit_once(es, pat, padding=pad)
            assert env.decrypt_bool(found) == 1 and decrypt_str(env, before).replace("\x00", "") == clear and \
                decrypt_str(env, after).replace("\x00", "") == "", tag
            for new in ("-", "xy", ""):
                content, _ = key.replace(es, pat, new, padding=pad)
                assert decrypt_str(env, content).replace("\x00", "") == clear.replace("", new), tag + (new,)
            content, _ = key.replacen(es, pat, "-", 2, padding=pad)
            assert decrypt_str(env, content).replace("\x00", "") == clear.replace("", "-", 2), tag
    # encrypted n with the empty pattern
    es = env.encrypt_str("abcd")
    for n in (0, 1, 3, 6, 7):
        assert split_result(env, key.splitn(es, env.encrypt_radix(n, 4), "")) == std_splitn("abcd", n, ""), n
        assert split_result(env, key.rsplitn(es, env.encrypt_radix(n, 4), env.encrypt_str("\x00"), padding=(0, 1))) == std_rsplitn("abcd", n, ""), n
    content, _ = key.replacen(es, "", "+", env.encrypt_radix(3, 4))
    assert decrypt_str(env, content).replace("\x00", "") == "abcd".replace("", "+", 3)
    # the same padded pattern argument, now NOT empty: ordinary behaviour
    assert split_result(env, key.split_terminator(env.encrypt_str("a,b,"), env.encrypt_str(",\x00"), padding=(0, 1))) == ["a", "b"]
    assert split_result(env, key.splitn(env.encrypt_str("a,b,c"), 2, env.encrypt_str(",\x00"), padding=(0, 1))) == ["a", "b,c"]
    content, _ = key.replace(env.encrypt_str("a,b"), env.encrypt_str(",\x00"), "--", padding=(0, 1))
    assert decrypt_str(env, content).replace("\x00", "") == "a--b"


def test_rfind_of_an_empty_pattern_hidden_under_padding(env):
    """"abc".rfind("") == Some(3): an encrypted pattern that is padding only must give the (possibly encrypted) length, like the
    clear empty pattern (find.rs:435-470, 545-565); find gives Some(0)"""
    key = env.key
    found, index = key.rfind(env.encrypt_str("abc"), "")
    assert env.decrypt_bool(found) == 1 and env.decrypt_radix(index) == 3
    for s, code in (("abc", 0), ("ab\x00\x00", 1), ("", 0)):
        clear = s.replace("\x00", "")
        es = env.encrypt_str(s)
        hidden = env.encrypt_str("\x00\x00")
        found, index = key.find_padded(es, hidden, padding=(code, 1))
        assert env.decrypt_bool(found) == 1 and env.decrypt_radix(index) == 0, s
        found, index = key.find_padded(es, hidden, padding=(code, 1), reverse=True)
        assert env.decrypt_bool(found) == 1 and env.decrypt_radix(index) == len(clear), s
        # ... and a padded pattern that is not empty is still searched from the end
        if clear:
            found, index = key.find_padded(es, env.encrypt_str(clear[-1] + "\x00"), padding=(code, 1), reverse=True)
            assert env.decrypt_bool(found) == 1 and env.decrypt_radix(index) == len(clear) - 1, s
