import sys, numpy as np
sys.path.insert(0,'.')
import tfhe_rs_string_amd as T
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(7)
ctx = T.Context(0)
sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
lut,_ = T.generate_lookup_table(p, lambda x: x%4)
names=['mirror_write','rotate+digit+twist','fwd stage1,2','fwd stage3 (swaps)','transpose+passB','exchange+wait1','pointwise','inv passB','wait2','inv transpose+stage3','inv stage2,1+accumulate']
for B in (1, 1024):
    data = rng.integers(0, 2**64, (B, p.small_size), dtype=np.uint64)
    out = sk.pbs_batch(data, lut, 0)
    for w in range(2):
        st = out[B-1, 64+16*w:64+16*w+11].astype(np.float64)
        tot = st.sum()
        print(f"B={B} wave {w}: total {tot/742:.0f} cycles/iter")
        for n_,v in zip(names, st): print(f"     {n_:32s} {v/742:8.0f} cyc  {100*v/tot:5.1f}%")
