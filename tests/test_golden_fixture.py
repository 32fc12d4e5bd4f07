"""Known-answer fixture tests/golden/ks_pbs_n2.npz (made by tests/golden/make_golden.py): the CPU
oracle (here) and the HIP path (-m gpu) must both reproduce the committed vectors bit for bit, so a
silent change of the arithmetic specification (twiddle tables, FMA placement, rounding) in either
is caught even when the two still agree with each other."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "ks_pbs_n2.npz"))


def keyset_from(oracle, g):
    ks = oracle.KeySet.__new__(oracle.KeySet)
    ks.p = oracle.param_message_2_carry_2(lwe_dim=int(g["lwe_dim"]))
    ks.glwe_sk, ks.lwe_sk, ks.bsk, ks.ksk = (np.ascontiguousarray(g[k]) for k in ("glwe_sk", "lwe_sk", "bsk", "ksk"))
    ks.fft = oracle.Fft(ks.p.poly_size)
    ks.bsk_re = ks.bsk_im = None
    ks.to_fourier()
    return ks


def test_oracle_reproduces_golden(oracle, gold):
    ks = keyset_from(oracle, gold)
    tr, ti = ks.fft.twist()
    wr, wi = ks.fft.twiddles()
    assert np.array_equal(tr, gold["twist_re"]) and np.array_equal(ti, gold["twist_im"])
    assert np.array_equal(wr, gold["w_re"]) and np.array_equal(wi, gold["w_im"])
    assert np.array_equal(ks.bsk_re, gold["bsk_fourier_re"]) and np.array_equal(ks.bsk_im, gold["bsk_fourier_im"])
    assert np.array_equal(ks.keyswitch_batch(gold["cts"]), gold["expected_small"])
    assert np.array_equal(ks.pbs_batch(gold["expected_small"], gold["luts"], gold["lut_idx"]), gold["expected_pbs"])
    assert np.array_equal(ks.ks_pbs_batch(gold["cts"], gold["luts"], gold["lut_idx"]), gold["expected_ks_pbs"])
    assert np.array_equal(ks.pbs_batch(gold["edge_small"], gold["luts"], gold["lut_idx"]), gold["expected_edge_pbs"])
    funcs = [lambda x: x % 4, lambda x: x // 4, lambda x: int(x // 4 == x % 4), lambda x: (x * x) % 16]
    dec = ks.decrypt(gold["expected_ks_pbs"])
    assert [int(d) for d in dec] == [funcs[i](int(m)) for i, m in zip(gold["lut_idx"], gold["msgs"])]


def test_wave_emulation_reproduces_golden(oracle, gold):
    """the kernel's lane-level source, run by the host emulation, against the committed vectors"""
    import ctypes as C

    from tests.test_wave_emulation import CSRC, EMU_DIR  # noqa: F401
    import subprocess

    lib = os.path.join(EMU_DIR, "libwave_emu.so")
    if not os.path.exists(lib):
        pytest.skip("emulation library not built yet (tests/test_wave_emulation.py builds it)")
    emu = C.CDLL(lib)
    n = int(gold["lwe_dim"])
    bsk = np.empty(gold["bsk_fourier_re"].size * 2)
    bsk[0::2], bsk[1::2] = gold["bsk_fourier_re"], gold["bsk_fourier_im"]
    for i in range(gold["expected_small"].shape[0]):
        out = np.zeros(2049, dtype=np.uint64)
        lut = np.ascontiguousarray(gold["luts"][gold["lut_idx"][i]])
        ct = np.ascontiguousarray(gold["expected_small"][i])
        emu.wave_emu_blind_rotate(bsk.ctypes.data_as(C.c_void_p), ct.ctypes.data_as(C.c_void_p),
                                  lut.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), n)
        assert np.array_equal(out, gold["expected_pbs"][i])


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [5, 4, 3, 2, 1, 0, -1])
def test_hip_reproduces_golden(gold, variant):
    import tfhe_rs_string_amd as T

    ctx = T.Context(0)
    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=int(gold["lwe_dim"]))
    sk = T.ServerKey(ctx, p, gold["bsk"], gold["ksk"])
    sk.set_variant(variant)
    re, im = sk.fourier_bsk()
    assert np.array_equal(re, gold["bsk_fourier_re"]) and np.array_equal(im, gold["bsk_fourier_im"])
    assert np.array_equal(sk.keyswitch_batch(gold["cts"]), gold["expected_small"])
    assert np.array_equal(sk.pbs_batch(gold["expected_small"], gold["luts"], gold["lut_idx"]), gold["expected_pbs"])
    assert np.array_equal(sk.ks_pbs_batch(gold["cts"], gold["luts"], gold["lut_idx"]), gold["expected_ks_pbs"])
    assert np.array_equal(sk.pbs_batch(gold["edge_small"], gold["luts"], gold["lut_idx"]), gold["expected_edge_pbs"])
