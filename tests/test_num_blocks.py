"""num_blocks of the reference's gen_keys_number_blocks (tfhe/examples/fhe_strings/ciphertext.rs:113-126: "should be larger than 4
to work with strings of length larger than 8 bits"): lengths, indices and counts as radix ciphertexts of num_blocks blocks, so
that find / len / trim / split / replace take strings beyond 255 chars.  CPU: the op-graph layer over the oracle-backed test
backend (n = 12); -m gpu: the same cases at PARAM_MESSAGE_2_CARRY_2_KS_PBS.  Checked like the reference's tests: decrypt and
compare with Rust's str semantics."""
import numpy as np
import pytest

from tests.test_host_ops_cpu import backend, decrypt_str, env  # noqa: F401  (fixtures)


def long_text(n, seed=7):
    rng = np.random.default_rng(seed)
    return "".join(rng.choice(list("abcdefgh "), size=n))


def run_cases(env, L):  # noqa: F811
    key = env.key
    s = long_text(L)
    s = s[:L - 30] + "needle" + s[L - 24:]
    es = env.encrypt_str(s)
    with pytest.raises(env.T.TfheHipError, match="num_blocks"):
        key.find(es, "needle")  # 4 blocks say 255 at most
    key.set_num_blocks(5)
    try:
        found, index = key.find(es, "needle")
        assert index.shape[0] == 5
        assert env.decrypt_bool(found) == 1 and env.decrypt_radix(index) == s.find("needle") == L - 30
        found, index = key.find(es, env.encrypt_str("zzz"))
        assert env.decrypt_bool(found) == 0 and env.decrypt_radix(index) == L   # find.rs:173-184: the index ran to the end
        found, index = key.rfind(es, "e")
        assert env.decrypt_bool(found) == 1 and env.decrypt_radix(index) == s.rfind("e")
        padded = s[:L - 9] + "\x00" * 9
        assert env.decrypt_radix(key.len(env.encrypt_str(padded), padding=1)) == L - 9
        content, length = key.trim(env.encrypt_str("   " + s[3:L - 2] + "  "))
        assert decrypt_str(env, content).replace("\x00", "") == ("   " + s[3:L - 2] + "  ").strip()
        assert env.decrypt_radix(length) == len(("   " + s[3:L - 2] + "  ").strip())
        # an encrypted n of the wider radix goes in as well (splitn counts) -- on a short string, for the time it takes
        parts, number = key.splitn(env.encrypt_str("a,b,c,d"), env.encrypt_radix(3, 5), ",")
        assert number.shape[0] == 5 and env.decrypt_radix(number) == 3
        assert [decrypt_str(env, p).replace("\x00", "") for p in parts[:3]] == ["a", "b", "c,d"]
    finally:
        key.set_num_blocks(4)
    with pytest.raises(env.T.TfheHipError, match="4 .. 16"):
        key.set_num_blocks(3)


def test_long_strings_with_five_blocks_cpu(env):  # noqa: F811
    run_cases(env, 300)


@pytest.fixture(scope="module")
def gpu_env(keyset_full):
    import tfhe_rs_string_amd as T

    ks = keyset_full
    ctx = T.Context(0)
    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
    sk = T.ServerKey(ctx, p, ks.bsk, ks.ksk)

    class E:
        pass

    e = E()
    e.T, e.ks, e.sk, e.key = T, ks, sk, T.StringServerKey(sk.handle, p)
    e.encrypt_str = lambda s: (ks.encrypt([(ord(ch) >> (2 * i)) & 3 for ch in s for i in range(4)]).reshape(len(s), 4, -1) if s
                               else np.zeros((0, 4, p.big_size), dtype=np.uint64))
    e.encrypt_radix = lambda v, nb: ks.encrypt([(v >> (2 * i)) & 3 for i in range(nb)])
    e.decrypt_radix = lambda blocks: sum(int(d % 4) << (2 * i) for i, d in enumerate(ks.decrypt(blocks)))
    e.decrypt_bool = lambda b: int(ks.decrypt(b.reshape(1, -1))[0])
    return e


@pytest.mark.gpu
def test_long_strings_with_five_blocks_gpu_env(gpu_env):
    run_cases(gpu_env, 300)
