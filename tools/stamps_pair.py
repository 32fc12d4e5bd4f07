"""Per-phase clock sums of the wave-pair kernel (diagnostic build -DTFHE_PAIR_STAMPS: the kernel stores the s_memtime sums of
wavefronts 0 / 1 of workgroup 0 behind the last output row).  TFHE_HIP_LIB=<diagnostic .so> python3 tools/stamps_pair.py [B]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import tfhe_rs_string_amd as T  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = T.Context(0)
pf = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(1234)
sk = T.ServerKey(ctx, pf, rng.integers(0, 2**64, pf.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, pf.ksk_len, dtype=np.uint64))
sk.set_variant(2)
small = rng.integers(0, 2**64, (B, pf.small_size), dtype=np.uint64)
lut = T.generate_lookup_table(pf, lambda x: x)[0]
for _ in range(3):
    out = sk.pbs_batch(small, lut, 0)
names = ["mirror write", "rotate+decompose", "stages 1-2", "stage 3", "transpose+pass B", "exchange+wait", "pointwise", "inv pass B",
         "wait 2", "inv transpose+stage 3'", "stages 2'-1'+accumulate", "-"]
row = out[0]
tot = [0, 0]
for w in range(2):
    v = row[64 + 16 * w: 64 + 16 * w + 12].astype(np.int64)
    tot[w] = int(v.sum())
    print(f"wave {w}: total {tot[w] / 742:.0f} clocks per step (s_memtime: 100 MHz units x ?)")
    for k, nm in enumerate(names[:11]):
        print(f"   {nm:26s} {v[k] / 742:9.1f}  {100.0 * v[k] / max(tot[w], 1):5.1f} %")
