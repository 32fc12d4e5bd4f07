import sys, time, numpy as np
sys.path.insert(0, '.')
import tfhe_rs_string_amd as T
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(7)
ctx = T.Context(0)
sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
key = T.StringServerKey(sk.handle, p)
hay = rng.integers(0, 2**64, (256, 4, p.big_size), dtype=np.uint64)
needle = rng.integers(0, 2**64, (8, 4, p.big_size), dtype=np.uint64)
for _ in range(2):
    key.contains(hay, needle)
print("--- timed", file=sys.stderr)
t0 = time.perf_counter(); key.contains(hay, needle); print((time.perf_counter() - t0) * 1e3, "ms", file=sys.stderr)
