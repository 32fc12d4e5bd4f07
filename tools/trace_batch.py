import sys, time, numpy as np
sys.path.insert(0,'.')
import tfhe_rs_string_amd as T
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(7)
ctx = T.Context(0)
sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
key = T.StringServerKey(sk.handle, p)
col = [rng.integers(0, 2**64, (64, 4, p.big_size), dtype=np.uint64) for _ in range(64)]
tgt = rng.integers(0, 2**64, (64, 4, p.big_size), dtype=np.uint64)
def run():
    with key.batch():
        return [key.eq(c, tgt) for c in col]
run()
t0=time.perf_counter(); run(); print('batch total ms', (time.perf_counter()-t0)*1e3, file=sys.stderr)
