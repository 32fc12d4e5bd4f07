#!/usr/bin/env python3
"""Generates tests/golden/ks_pbs_n2.npz: a tiny, self-contained known-answer set for the KS-PBS path
(PARAM_MESSAGE_2_CARRY_2_KS_PBS with the LWE dimension cut to 2: same N, k, bases and levels, so the
same code paths, but the keys are a few hundred KB).

The reference holds no golden ciphertexts for this path (its tests decrypt and compare), and it
cannot be executed in this image (Rust).  The vectors are therefore produced by the CPU oracle
(oracle/tfhe_oracle.c), whose integer pieces are pinned on the reference's doc-test constants
(tests/test_oracle_golden.py).  They freeze the engine's arithmetic specification:
both the oracle and the HIP path must keep reproducing them bit for bit.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402

FUNCS = [lambda x: x % 4, lambda x: x // 4, lambda x: int(x // 4 == x % 4), lambda x: (x * x) % 16]


def main():
    p = O.param_message_2_carry_2(lwe_dim=2)
    ks = O.KeySet(p, seed=0x601D)
    msgs = np.array([0, 5, 10, 15, 3, 12], dtype=np.uint64)
    cts = ks.encrypt(msgs)
    luts = np.stack([ks.lut([f(i) for i in range(16)])[0] for f in FUNCS])
    idx = np.array([0, 1, 2, 3, 3, 0], dtype=np.uint32)
    small = ks.keyswitch_batch(cts)
    # adversarial small ciphertexts: zero mask element, modulus switch to 2N and to N
    edge = small.copy()
    edge[0, 0] = 0
    edge[1, :] = np.uint64(2**64 - 1)
    edge[2, 1] = np.uint64(1) << np.uint64(63)
    out = {
        "lwe_dim": np.uint32(p.lwe_dim),
        "glwe_sk": ks.glwe_sk, "lwe_sk": ks.lwe_sk, "bsk": ks.bsk, "ksk": ks.ksk,
        "bsk_fourier_re": ks.bsk_re, "bsk_fourier_im": ks.bsk_im,
        "msgs": msgs, "cts": cts, "luts": luts, "lut_idx": idx,
        "expected_small": small,
        "expected_pbs": ks.pbs_batch(small, luts, idx),
        "expected_ks_pbs": ks.ks_pbs_batch(cts, luts, idx),
        "edge_small": edge, "expected_edge_pbs": ks.pbs_batch(edge, luts, idx),
        "twist_re": ks.fft.twist()[0], "twist_im": ks.fft.twist()[1],
        "w_re": ks.fft.twiddles()[0], "w_im": ks.fft.twiddles()[1],
    }
    path = os.path.join(ROOT, "tests", "golden", "ks_pbs_n2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
