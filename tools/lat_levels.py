# host-to-host time of one level (pinned buffers) for level sizes typical of the op-graphs
import sys, time, numpy as np
sys.path.insert(0,'.')
import tfhe_rs_string_amd as T
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(7)
ctx = T.Context(0)
sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
lut,_ = T.generate_lookup_table(p, lambda x: x%4)
pin_lut = T.PinnedArray(lut.shape); pin_lut.array[:] = lut
for B in (768, 1100, 1536, 2148, 3000, 4096, 8192):
    pin_in, pin_out = T.PinnedArray((B, p.big_size)), T.PinnedArray((B, p.big_size))
    pin_in.array[:] = rng.integers(0, 2**64, (B, p.big_size), dtype=np.uint64)
    sk.ks_pbs_batch(pin_in.array, pin_lut.array, 0, out=pin_out.array)
    t0 = time.perf_counter()
    for _ in range(3): sk.ks_pbs_batch(pin_in.array, pin_lut.array, 0, out=pin_out.array)
    dt = (time.perf_counter() - t0) / 3
    print(f"B={B:5d} host-to-host {dt*1e3:8.3f} ms -> {B/dt:9.0f} PBS/s", flush=True)
