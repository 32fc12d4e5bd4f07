# experiment: consecutive independent batches on two alternating streams (keyswitch of batch i+1 under the blind rotation of batch i)
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import tfhe_rs_string_amd as T
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(7)
ctx = T.Context(0)
sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
dev = torch.device("cuda:0")
B = 1024
d_in = torch.from_numpy(rng.integers(0, 2**63, (B, p.big_size), dtype=np.int64)).to(dev)
lut = T.generate_lookup_table(p, lambda x: x % 4)[0]
d_luts = torch.from_numpy(lut.view(np.int64)).reshape(1, -1).to(dev)
d_idx = torch.zeros(B, dtype=torch.int32, device=dev)
outs = [torch.empty_like(d_in) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(nstreams, steps=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        st = streams[i % nstreams]
        sk.ks_pbs_batch_device(d_in, d_luts, d_idx, outs[i % 2], stream=st.cuda_stream)
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)
for n in (1, 2, 1, 2):
    run(n, 4)
    print(f"{n} stream(s): {run(n):.0f} PBS/s")
