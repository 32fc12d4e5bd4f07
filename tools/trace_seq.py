import sys, time, numpy as np
sys.path.insert(0,'.')
import tfhe_rs_string_amd as T
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(7)
ctx = T.Context(0)
sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
lut,_ = T.generate_lookup_table(p, lambda x: x%4)
luts = np.stack([lut, lut])
data = rng.integers(0, 2**64, (1024, p.big_size), dtype=np.uint64)
sk.ks_pbs_batch(data[:256], lut, 0)
sk.profiling(True)
for B, nl in [(256,1),(18,1),(18,2),(18,2),(256,1),(18,2),(2,2),(256,2),(114,2),(56,2),(1024,1),(18,1),(18,1)]:
    sk.profiling_read(0, reset=True)
    t0=time.perf_counter()
    sk.ks_pbs_batch(data[:B], luts[:nl], 0)
    dt=(time.perf_counter()-t0)*1e3
    ks,_=sk.profiling_read(0); br,_=sk.profiling_read(1)
    print(f"B={B:5d} luts={nl} wall {dt:7.2f} ms  ks {ks:6.2f}  br {br:6.2f}", flush=True)
