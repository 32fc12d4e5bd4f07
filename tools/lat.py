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
if len(sys.argv) > 1: sk.set_variant(int(sys.argv[1])); print('variant', sys.argv[1])
for B in (1, 16, 64, 256, 512, 1024):
    sk.ks_pbs_batch(data[:B], lut, 0)
    sk.profiling_read(0, reset=True)
    for _ in range(3): sk.ks_pbs_batch(data[:B], lut, 0)
    br,n=sk.profiling_read(1)
    print(f"B={B:5d} blind-rotate {br/n:7.3f} ms", flush=True)
