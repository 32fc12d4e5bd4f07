"""DIAGNOSTIC: per-phase clock sums of wavefront 0 of the quad kernel (library built by tools/build_stamps.sh with
TFHE_QUAD_STAMPS; run with TFHE_HIP_LIB pointing at it)."""
import sys
import numpy as np
sys.path.insert(0, '.')
import tfhe_rs_string_amd as T
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(7)
ctx = T.Context(0)
sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
sk.set_variant(5)
lut, _ = T.generate_lookup_table(p, lambda x: x % 4)
names = ['prefetch issue + mirror write', 'barrier 1', 'rotate+decompose+twist', 'stages 1,2,3a + transposition write', 'barrier 2',
         'pass B + exchange write', 'barrier 3', "pointwise + pass B' + transposition write", 'barrier 4', "pass A' + accumulate"]
for B in (1, 256, 512):
    data = rng.integers(0, 2**64, (B, p.small_size), dtype=np.uint64)
    out = sk.pbs_batch(data, lut, 0)
    st = out[0, 64:64 + 10].astype(np.float64)
    tot = st.sum()
    print(f"B={B} wavefront 0: total {tot / 742:.0f} ticks/step (s_memtime, 100 MHz: x 24 = clocks at 2.4 GHz)")
    for n_, v in zip(names, st):
        print(f"     {n_:44s} {v / 742:8.1f}  {100 * v / tot:5.1f}%")
