import sys, time, numpy as np
sys.path.insert(0,'.')
import tfhe_rs_string_amd as T
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(7)
ctx = T.Context(0)
sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
lut,_ = T.generate_lookup_table(p, lambda x: x%4)
for B in (1024, 8192):
    data = rng.integers(0, 2**64, (B, p.big_size), dtype=np.uint64)
    pin_in, pin_out, pin_lut = T.PinnedArray((B, p.big_size)), T.PinnedArray((B, p.big_size)), T.PinnedArray(lut.shape)
    pin_in.array[:] = data; pin_lut.array[:] = lut
    for name, a, l, o in (("pageable", data, lut, None), ("pinned", pin_in.array, pin_lut.array, pin_out.array)):
        sk.ks_pbs_batch(a, l, 0, out=o)
        t0 = time.perf_counter()
        for _ in range(3): sk.ks_pbs_batch(a, l, 0, out=o)
        dt = (time.perf_counter() - t0) / 3
        print(f"B={B:5d} {name:9s} host-to-host {dt*1e3:8.3f} ms -> {B/dt:9.0f} PBS/s (PCIe-inclusive)", flush=True)
