"""blind-rotation time of the multi-bit PBS kernels (grouping factor 2 / 3) at their full parameter sets, synthetic keys.
  python3 tools/bench_multi_bit.py [batch]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import tfhe_rs_string_amd as T  # noqa: E402
import torch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = T.Context(0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
for g, p in ((2, T.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS()), (3, T.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS())):
    n_ggsw = (p.lwe_dim // g) << g
    bsk_len = n_ggsw * p.pbs_level * (p.glwe_dim + 1) ** 2 * p.poly_size
    sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64), grouping_factor=g)
    lut, _ = T.generate_lookup_table(p, lambda x: x % 4)
    d_in = torch.from_numpy(rng.integers(0, 2**64, (B, p.big_size), dtype=np.uint64).view(np.int64)).to(dev)
    d_luts = torch.from_numpy(lut.reshape(1, -1).view(np.int64)).to(dev)
    d_idx = torch.zeros(B, dtype=torch.int32, device=dev)
    d_out = torch.empty_like(d_in)
    for _ in range(3):
        sk.ks_pbs_batch_device(d_in, d_luts, d_idx, d_out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        sk.ks_pbs_batch_device(d_in, d_luts, d_idx, d_out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"grouping factor {g}, B = {B}: {dt * 1e3:.3f} ms per batch = {B / dt:.0f} PBS/s; status {sk.device_status()}", flush=True)
