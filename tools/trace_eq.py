import sys, time, numpy as np
sys.path.insert(0,'.')
import tfhe_rs_string_amd as T
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(7)
ctx = T.Context(0)
sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
key = T.StringServerKey(sk.handle, p)
s1 = rng.integers(0, 2**64, (64,4,p.big_size), dtype=np.uint64); s2 = rng.integers(0, 2**64, (64,4,p.big_size), dtype=np.uint64)
key.eq(s1,s2)
for _ in range(2):
    t0=time.perf_counter(); key.eq(s1,s2); print('eq total ms', (time.perf_counter()-t0)*1e3, file=sys.stderr)
a = rng.integers(0, 2**64, (16,p.big_size), dtype=np.uint64); b = rng.integers(0, 2**64, (16,p.big_size), dtype=np.uint64)
key.mul_parallelized(a,b)
t0=time.perf_counter(); key.mul_parallelized(a,b); print('mul total ms', (time.perf_counter()-t0)*1e3, file=sys.stderr)
print('--- with kernel profiling', file=sys.stderr)
sk.profiling(True)
import os
for _ in range(2):
    sk.profiling_read(0, reset=True)
    t0=time.perf_counter(); key.eq(s1,s2); dt=(time.perf_counter()-t0)*1e3
    ks,nk=sk.profiling_read(0); br,nb=sk.profiling_read(1)
    print(f'eq total {dt:.1f} ms; kernel sums: ks {ks:.2f} ms over {nk}, br {br:.2f} ms over {nb}', file=sys.stderr)
