"""KS-PBS throughput of every classic parameter set the library takes (T.CLASSIC_KS_PBS) at its full size, synthetic keys, inputs
resident in HBM: one JSON line per set (which kernel ran, ms per batch, PBS/s).  The 4-bit sets run on the wave / workgroup kernels
(the headline path), the others on the generic kernel (correct, not tuned).   python3 tools/bench_param_sets.py [batch]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import tfhe_rs_string_amd as T  # noqa: E402
import torch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = T.Context(0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(99)
for name in T.CLASSIC_KS_PBS:
    p = T.classic_params(name)
    sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
    lut, _ = T.generate_lookup_table(p, lambda x: x % p.message_modulus)
    d_in = torch.from_numpy(rng.integers(0, 2**64, (B, p.big_size), dtype=np.uint64).view(np.int64)).to(dev)
    d_luts = torch.from_numpy(lut.reshape(1, -1).view(np.int64)).to(dev)
    d_idx = torch.zeros(B, dtype=torch.int32, device=dev)
    d_out = torch.empty_like(d_in)
    for _ in range(2):
        sk.ks_pbs_batch_device(d_in, d_luts, d_idx, d_out)
    torch.cuda.synchronize()
    steps = 5
    t0 = time.perf_counter()
    for _ in range(steps):
        sk.ks_pbs_batch_device(d_in, d_luts, d_idx, d_out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fast = p.glwe_dim == 1 and p.poly_size == 2048 and p.pbs_level == 1 and p.pbs_base_log == 23
    print(json.dumps({"set": name, "lwe_dim": p.lwe_dim, "glwe_dim": p.glwe_dim, "poly_size": p.poly_size,
                      "pbs": f"{p.pbs_base_log}x{p.pbs_level}", "ks": f"{p.ks_base_log}x{p.ks_level}", "batch": B,
                      "kernel": "wave-pair (variant 2)" if fast else "generic (variant 0)", "ms_per_batch": round(dt * 1e3, 3),
                      "pbs_per_s": round(B / dt, 1), "device_status": sk.device_status(), "data": "synthetic"}), flush=True)
    del sk, d_in, d_out
