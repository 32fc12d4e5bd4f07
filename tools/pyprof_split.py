import sys, time, numpy as np, cProfile, pstats
sys.path.insert(0, '.')
import tfhe_rs_string_amd as T
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(7)
ctx = T.Context(0)
sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
key = T.StringServerKey(sk.handle, p)
s = rng.integers(0, 2**64, (40, 4, p.big_size), dtype=np.uint64)
key.split(s, ",")
for rep in range(3):
    t0 = time.perf_counter(); r = key.split(s, ","); t1 = time.perf_counter(); del r; t2 = time.perf_counter()
    print(f"call {1e3*(t1-t0):.1f} ms, free {1e3*(t2-t1):.1f} ms")
pr = cProfile.Profile(); pr.enable(); r = key.split(s, ","); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(8)
