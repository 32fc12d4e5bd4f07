import sys, time, numpy as np
sys.path.insert(0,'.')
import tfhe_rs_string_amd as T
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(7)
ctx = T.Context(0)
sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
lut,_ = T.generate_lookup_table(p, lambda x: x%4)
data = rng.integers(0, 2**64, (1024, p.big_size), dtype=np.uint64)
sk.profiling(True)
for B in (1, 16, 64, 256, 512, 1024):
    sk.ks_pbs_batch(data[:B], lut, 0)
    sk.profiling_read(0, reset=True); sk.profiling_read(1, reset=True)
    t0 = time.perf_counter()
    for _ in range(5): sk.ks_pbs_batch(data[:B], lut, 0)
    wall = (time.perf_counter() - t0) / 5
    ks,n0 = sk.profiling_read(0); br,n1 = sk.profiling_read(1)
    print(f"B={B:5d} keyswitch {ks/n0:7.3f} ms  blind-rotate {br/n1:7.3f} ms  host-to-host {wall*1e3:7.3f} ms", flush=True)
