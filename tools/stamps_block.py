import sys, numpy as np
sys.path.insert(0,'.')
import tfhe_rs_string_amd as T
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(7)
ctx = T.Context(0)
sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
sk.set_variant(3)
lut,_ = T.generate_lookup_table(p, lambda x: x%4)
names=['rotate+stage1+store','barrier 1','fwd 2,3','fwd 4','barrier 2','stage5 x2+pointwise+5\'','inv 4,3,2','barrier 3','inv 1+accumulate+mirror','barrier 4']
for B in (1, 256):
    data = rng.integers(0, 2**64, (B, p.small_size), dtype=np.uint64)
    out = sk.pbs_batch(data, lut, 0)
    st = out[B-1, 64:64+10].astype(np.float64)
    tot = st.sum()
    print(f"B={B} wavefront 0: total {tot/742:.0f} clocks/iter (s_memtime, 100 MHz)")
    for n_,v in zip(names, st): print(f"     {n_:32s} {v/742:8.1f}  {100*v/tot:5.1f}%")
