import sys, time, os, numpy as np
sys.path.insert(0, '.')
import tfhe_rs_string_amd as T
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(7)
ctx = T.Context(0)
sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
key = T.StringServerKey(sk.handle, p)
for L in (8, 12, 16, 24):
    s = rng.integers(0, 2**64, (L, 4, p.big_size), dtype=np.uint64)
    for mode in ("rank", "network"):
        if mode == "network": os.environ["TFHE_HIP_COMPACT_NETWORK"] = "1"
        else: os.environ.pop("TFHE_HIP_COMPACT_NETWORK", None)
        key.push_padding_to_end(s); key.stats(reset=True)
        t0 = time.perf_counter(); r = key.push_padding_to_end(s); dt = time.perf_counter() - t0
        st = key.stats()
        print(f"L={L:3d} {mode:8s} {dt*1e3:6.1f} ms {st['pbs']:5d} PBS {st['launches']:2d} launches")
