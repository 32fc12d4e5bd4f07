"""Measures the noise of bootstrap outputs at PARAM_MESSAGE_2_CARRY_2_KS_PBS (supports DESIGN.md's noise-budget note):
standard deviation of decrypt(PBS_identity(enc(m))) - m * Delta, as a fraction of the torus."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import tfhe_rs_string_amd as T  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

ks = O.KeySet(O.param_message_2_carry_2(), seed=7)
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
sk = T.ServerKey(T.Context(0), p, ks.bsk, ks.ksk)
B = 2048
msgs = np.arange(B) % 16
cts = ks.encrypt(msgs)
lut, _ = T.generate_lookup_table(p, lambda x: x)
out = sk.ks_pbs_batch(cts, lut, 0)
delta = (1 << 63) // 16
raw = ks.decrypt_raw(out).astype(np.uint64)
err = (raw - msgs.astype(np.uint64) * np.uint64(delta)).astype(np.int64).astype(np.float64) / 2.0**64
print(f"bootstrap output noise: std {err.std():.3e} of the torus (2^{np.log2(err.std()):.1f}), max |e| {np.abs(err).max():.3e}; half slot = {1/64:.3e}")
fresh = ks.encrypt(msgs)
rawf = ks.decrypt_raw(fresh).astype(np.uint64)
errf = (rawf - msgs.astype(np.uint64) * np.uint64(delta)).astype(np.int64).astype(np.float64) / 2.0**64
print(f"fresh encryption noise (big key): std {errf.std():.3e}")
small = ks.keyswitch_batch(out)
raws = ks.decrypt_small_raw(small).astype(np.uint64)
errs = (raws - msgs.astype(np.uint64) * np.uint64(delta)).astype(np.int64).astype(np.float64) / 2.0**64
print(f"after keyswitch (small key): std {errs.std():.3e}; modulus switch adds ~ sqrt((n+1)/12)/(2N) = {np.sqrt(743/12)/4096:.3e}")
