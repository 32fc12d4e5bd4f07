"""The output contract of the string / integer ABI: the reference hands every result back as
Ciphertext{degree, noise_level = NOMINAL} (tfhe/src/shortint/ciphertext/mod.rs:263-270, server_key/mod.rs:855-856), so a
caller may sum 15 such blocks and bootstrap.  Op-graphs here that end in a leveled step (if_then_else / min / max:
the selected side; to_uppercase: the case-adjusted block) clean their results by default (`strict_outputs`); without,
tfhe_hip_host_key_output_metadata reports the higher level.  Both modes: feed the results of to_uppercase and
if_then_else into a 15-term sum + PBS (what tfhe-rs allows on NOMINAL blocks, scalar_comparison.rs:164-185).
CPU: the oracle-backed stand-in of the device entry points; -m gpu: the HIP path at full parameters."""
import numpy as np
import pytest

from tests.test_host_ops_cpu import Env, backend, decrypt_str  # noqa: F401  (backend: fixture)


def run_contract(env, shortint):
    key = env.key
    s = "pqrstuvwxyzpqrs"                       # 15 letters whose upper-case block 2 (bits 4-5) is 1, lower-case 3
    es = env.encrypt_str(s)
    cond1, cond0 = env.ks.encrypt([1])[0], env.ks.encrypt([0])[0]
    a, b = env.encrypt_radix(0xB7, 4), env.encrypt_radix(0x2E, 4)
    p = env.params
    for strict in (True, False):
        key.set_strict_outputs(strict)
        up = key.to_uppercase(es)
        deg_up, noise_up = key.output_metadata(up)
        assert decrypt_str(env, up) == s.upper()
        sel = key.if_then_else_parallelized(cond1, a, b)
        deg_sel, noise_sel = key.output_metadata(sel)
        assert env.decrypt_radix(sel) == 0xB7
        assert env.decrypt_radix(key.if_then_else_parallelized(cond0, a, b)) == 0x2E
        assert deg_up.max() <= 3 and deg_sel.max() <= 3
        noise_up = noise_up.reshape(len(s), 4)
        if strict:   # NOMINAL, like every result of the reference
            assert set(noise_up.ravel().tolist()) == {1} and set(noise_sel.tolist()) == {1}
        else:        # reported as they are: the sum of the two masked sides / block 2 -+ 2 * flag
            assert noise_sel.tolist() == [2, 2, 2, 2]
            assert noise_up[:, 2].tolist() == [3] * len(s) and noise_up[:, [0, 1, 3]].max() == 1
        # what tfhe-rs allows on NOMINAL blocks (are_all_comparisons_block_true, scalar_comparison.rs:164-185): 15 blocks
        # summed, one PBS x == 15 -- here the case-adjusted blocks themselves
        blocks = up[:, 2]
        total = np.zeros(p.big_size, dtype=np.uint64)
        for blk in blocks:
            total = total + blk
        lut_all, _ = env.T.generate_lookup_table(p, lambda x: int(x == 15))
        res, deg, noise = shortint.apply_lookup_table(total.reshape(1, -1), 15, int(noise_up[:, 2].sum()), lut_all, [1], 0)
        assert env.decrypt_bool(res[0]) == 1 and int(noise[0]) == 1 and int(deg[0]) == 1
        # and the bivariate packing 4 * hi + lo of two selected blocks, then a PBS
        packed = np.uint64(4) * sel[1] + sel[0]
        lut_id, _ = env.T.generate_lookup_table(p, lambda x: x)
        res, _, _ = shortint.apply_lookup_table(packed.reshape(1, -1), 15, 5 * int(noise_sel.max()), lut_id, [15], 0)
        assert int(env.ks.decrypt(res)[0]) == 4 * ((0xB7 >> 2) & 3) + (0xB7 & 3)
    key.set_strict_outputs(True)


def test_output_contract_cpu(oracle, backend, keyset_small):  # noqa: F811
    import tfhe_rs_string_amd as T

    env = Env(oracle, backend, keyset_small)
    shortint = T.ShortintServerKey(env.sk, env.params, lib=backend)
    probe = np.zeros(env.params.big_size, dtype=np.uint64)  # allocated first and kept alive: no output of the calls below can
    run_contract(env, shortint)                             # have had its address (the metadata is keyed by output address)
    with pytest.raises(T.TfheHipError, match="was not written"):
        env.key.output_metadata(probe)


@pytest.mark.gpu
def test_output_contract_gpu(keyset_full):
    import tfhe_rs_string_amd as T

    ks = keyset_full
    ctx = T.Context(0)
    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
    sk = T.ServerKey(ctx, p, ks.bsk, ks.ksk)

    class E:
        pass

    e = E()
    e.T, e.ks, e.params = T, ks, p
    e.key = T.StringServerKey(sk.handle, p)
    e.encrypt_str = lambda s: ks.encrypt([(ord(ch) >> (2 * i)) & 3 for ch in s for i in range(4)]).reshape(len(s), 4, -1)
    e.encrypt_radix = lambda v, nb: ks.encrypt([(v >> (2 * i)) & 3 for i in range(nb)])
    e.decrypt_radix = lambda blocks: sum(int(d % 4) << (2 * i) for i, d in enumerate(ks.decrypt(blocks)))
    e.decrypt_bool = lambda b: int(ks.decrypt(b.reshape(1, -1))[0])
    run_contract(e, T.ShortintServerKey(sk.handle, p))
