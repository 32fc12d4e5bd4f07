import sys, time, numpy as np
sys.path.insert(0, '.')
import tfhe_rs_string_amd as T
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(7)
ctx = T.Context(0)
sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
key = T.StringServerKey(sk.handle, p)
for L in (12, 40, 100):
    s = rng.integers(0, 2**64, (L, 4, p.big_size), dtype=np.uint64)
    for name, fn in (("len", lambda: key.len(s)), ("is_empty", lambda: key.is_empty(s))):
        fn(); key.stats(reset=True)
        t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
        st = key.stats()
        print(f"L={L:3d} {name:9s} {dt*1e3:6.1f} ms {st['pbs']:5d} PBS {st['launches']:2d} launches")
