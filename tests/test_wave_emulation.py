"""CPU check of the wave-per-ciphertext kernel's lane-level source (tfhe_rs_string_amd/csrc/
wave_pbs.hpp) through the host emulation in tests/emu: the exact phase functions the GPU runs are
executed lane by lane and must reproduce the oracle bit for bit (index maps, transposition,
half-swap, BSK lane order, rounding).  The GPU run of the same source is covered by -m gpu tests."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
CSRC = os.path.join(ROOT, "tfhe_rs_string_amd", "csrc")


@pytest.fixture(scope="module")
def emu():
    lib = os.path.join(EMU_DIR, "libwave_emu.so")
    srcs = [os.path.join(EMU_DIR, "wave_emu.cpp"), os.path.join(CSRC, "twiddle.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("wave_pbs.hpp", "pair_pbs.hpp", "block_pbs.hpp", "block_mb_pbs.hpp", "quad_pbs.hpp", "carry_chain.hpp", "wave_tables.hpp", "device_math.hpp")]
    if not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.check_call(["g++", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-std=c++17",
                               "-DTFHE_HIP_HOST_EMU", "-fPIC", "-shared", "-o", lib] + srcs)
    return C.CDLL(lib)


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def test_forward_transform_bitexact(emu, oracle):
    f = oracle.Fft(2048)
    rng = np.random.default_rng(5)
    for _ in range(3):
        d = rng.integers(-(2**22), 2**22 + 1, 2048).astype(np.int64)
        re, im = np.zeros(1024), np.zeros(1024)
        emu.wave_emu_forward(vp(d), vp(re), vp(im))
        ore, oim = f.forward_integer(d)
        assert np.array_equal(re, ore) and np.array_equal(im, oim)


@pytest.mark.parametrize("entry", ["wave_emu_blind_rotate", "pair_emu_blind_rotate", "block_emu_blind_rotate", "half_emu_blind_rotate",
                                   "quad_emu_blind_rotate"])
def test_blind_rotation_bitexact(emu, oracle, entry):
    ks = oracle.KeySet(oracle.param_message_2_carry_2(lwe_dim=6), seed=3)
    bsk = np.empty(ks.bsk_re.size * 2)
    bsk[0::2], bsk[1::2] = ks.bsk_re, ks.bsk_im
    luts = [ks.lut([(3 * x) % 16 for x in range(16)])[0], ks.lut([x // 4 for x in range(16)])[0]]
    small = ks.keyswitch_batch(ks.encrypt([5, 11, 0]))
    small[2, :3] = [0, 2**64 - 1, 1 << 63]  # a_hat = 0 (the reference skips it), 2N and N
    for i, ct in enumerate(small):
        lut = luts[i % 2]
        want = ks.pbs_batch(ct, lut, np.zeros(1, dtype=np.uint32))[0]
        out = np.zeros(2049, dtype=np.uint64)
        getattr(emu, entry)(vp(bsk), vp(np.ascontiguousarray(ct)), vp(lut), vp(out), 6)
        assert np.array_equal(out, want)


def test_quad_transposition_slots_and_lane_groups(emu):
    """variant 5: every 16-B slot of the transposition image written / read exactly once per transform, and the lane
    groups of ds_write_b128 / ds_read_b128 (MI355X_MICROARCH.md, LDS) hit distinct columns in both directions"""
    assert emu.quad_emu_check_transposition() == 0


@pytest.mark.parametrize("g", [2, 3])
def test_multi_bit_blind_rotation_bitexact(emu, oracle, g):
    """multi-bit blind rotation (block_mb_pbs.hpp) against the oracle's orc_mb_pbs, grouping factors 2 and 3"""
    P = oracle.param_multi_bit_message_2_carry_2_group_2 if g == 2 else oracle.param_multi_bit_message_2_carry_2_group_3
    n = 6
    ks = oracle.MultiBitKeySet(P(lwe_dim=n), g, seed=11)
    bsk = np.empty(ks.bsk_re.size * 2)
    bsk[0::2], bsk[1::2] = ks.bsk_re, ks.bsk_im
    luts = [ks.lut([(3 * x) % 16 for x in range(16)])[0], ks.lut([x // 4 for x in range(16)])[0]]
    small = ks.keyswitch_batch(ks.encrypt([5, 11, 0]))
    small[2, :3] = [0, 2**64 - 1, 1 << 63]
    for i, ct in enumerate(small):
        lut = luts[i % 2]
        want = ks.pbs_batch(ct, lut, 0)[0]
        out = np.zeros(2049, dtype=np.uint64)
        emu.block_mb_emu_blind_rotate(vp(bsk), vp(np.ascontiguousarray(ct)), vp(lut), vp(out), n, g)
        assert np.array_equal(out, want)
