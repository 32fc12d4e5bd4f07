# per-level host timing of FheString::eq (64/64) and FheUint32 mul on random keys (run with TFHE_HIP_HOST_TRACE=1)
import sys, time, numpy as np
sys.path.insert(0, '.')
import tfhe_rs_string_amd as T
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(7)
ctx = T.Context(0)
sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
key = T.StringServerKey(sk.handle, p)
s1 = rng.integers(0, 2**64, (64, 4, p.big_size), dtype=np.uint64)
s2 = rng.integers(0, 2**64, (64, 4, p.big_size), dtype=np.uint64)
a = rng.integers(0, 2**64, (16, p.big_size), dtype=np.uint64)
b = rng.integers(0, 2**64, (16, p.big_size), dtype=np.uint64)
for name, fn in [("eq 64/64", lambda: key.eq(s1, s2)), ("mul 16 blocks", lambda: key.mul_parallelized(a, b))]:
    for _ in range(3):
        fn()
    key.stats(reset=True)
    print(f"--- {name}", file=sys.stderr)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    st = key.stats()
    print(f"{name}: {[round(t * 1e3, 2) for t in ts]} ms, {st['pbs'] // 5} PBS, {st['launches'] // 5} launches", file=sys.stderr)
