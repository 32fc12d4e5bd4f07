import sys, time, numpy as np
sys.path.insert(0,'.')
import tfhe_rs_string_amd as T
rng = np.random.default_rng(7)
ctx = T.Context(0)
for g, P in ((2, T.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS), (3, T.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS)):
    p = P()
    bsk_len = (p.lwe_dim // g) * (1 << g) * 4 * p.poly_size
    sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64), grouping_factor=g)
    lut,_ = T.generate_lookup_table(p, lambda x: x%4)
    data = rng.integers(0, 2**64, (1024, p.big_size), dtype=np.uint64)
    sk.profiling(True)
    for B in (1, 64, 256, 512, 1024):
        sk.ks_pbs_batch(data[:B], lut, 0)
        sk.profiling_read(0, reset=True); sk.profiling_read(1, reset=True)
        t0 = time.perf_counter()
        for _ in range(3): sk.ks_pbs_batch(data[:B], lut, 0)
        wall = (time.perf_counter() - t0) / 3
        ks,n0 = sk.profiling_read(0); br,n1 = sk.profiling_read(1)
        print(f"g={g} n={p.lwe_dim} B={B:5d} keyswitch {ks/n0:7.3f} ms  multi-bit blind-rotate {br/n1:7.3f} ms  host-to-host {wall*1e3:7.3f} ms  -> {B/(ks/n0+br/n1)*1e3:9.0f} PBS/s (kernels)", flush=True)
