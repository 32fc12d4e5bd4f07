# one-rank RCCL rehearsal of bench.py's exchange (the API calls, dtypes and stream semantics of the N > 1 path):
#   python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/rccl_rehearsal.py
import os, sys, time
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import tfhe_rs_string_amd as T

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29511")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=16)
rng = np.random.default_rng(1)
sk = T.ServerKey(T.Context(0), p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
B = 64
d_in = torch.from_numpy(rng.integers(0, 2**63, (B, p.big_size), dtype=np.int64)).to(dev)
lut = T.generate_lookup_table(p, lambda x: x % 4)[0]
d_luts = torch.from_numpy(lut.view(np.int64)).reshape(1, -1).to(dev)
d_idx = torch.zeros(B, dtype=torch.int32, device=dev)
outs = [torch.empty_like(d_in) for _ in range(2)]
g = T.ResultGather(dist, outs[0])
ref = None
for step in range(6):
    o = outs[step & 1]
    while len(g.pending) >= 2:
        g.pending.pop(0).wait()
    sk.ks_pbs_batch_device(d_in, d_luts, d_idx, o)
    glist = g.submit(o)
g.drain()
torch.cuda.synchronize()
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert torch.equal(glist[0], outs[1]) and float(t.item()) == 1.5
print("rccl rehearsal ok: gather(async) of int64 LWEs + barrier + all_reduce(MAX) on", torch.cuda.get_device_name(0))
dist.destroy_process_group()
