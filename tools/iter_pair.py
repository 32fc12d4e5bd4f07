"""GPU iteration helper for the blind-rotation kernels: bit-exact check of a variant against the oracle (n = 12, several
batch shapes, edge inputs from the golden set) and HIP-event timing of the blind rotation at full size.
  python3 tools/iter_pair.py [variant] [batch]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import tfhe_rs_string_amd as T  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ctx = T.Context(0)
ks = O.KeySet(O.param_message_2_carry_2(lwe_dim=12), seed=0x5EED0001)
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=12)
sk = T.ServerKey(ctx, p, ks.bsk, ks.ksk)
sk.set_variant(variant)
funcs = [lambda x: x % 4, lambda x: x // 4, lambda x: int(x // 4 == x % 4), lambda x: (x * x) % 16]
luts = np.stack([T.generate_lookup_table(p, f)[0] for f in funcs])
ok = True
for nb in (1, 5, 64, 300):
    cts = ks.encrypt(np.arange(nb) % 16)
    idx = (np.arange(nb) % 4).astype(np.uint32)
    got = sk.ks_pbs_batch(cts, luts, idx)
    want = ks.ks_pbs_batch(cts, luts, idx, threads=8)
    same = np.array_equal(got, want)
    ok &= same
    print(f"variant {variant} B={nb}: bit-exact {same}", flush=True)
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ks_pbs_n2.npz"))
sg = T.ServerKey(ctx, T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=int(g["lwe_dim"])), g["bsk"], g["ksk"])
sg.set_variant(variant)
same = np.array_equal(sg.pbs_batch(g["edge_small"], g["luts"], g["lut_idx"]), g["expected_edge_pbs"]) and \
    np.array_equal(sg.ks_pbs_batch(g["cts"], g["luts"], g["lut_idx"]), g["expected_ks_pbs"])
ok &= same
print("golden (edge rotations, all-zero masks):", same, flush=True)
# random small LWEs incl. zeros -> exercises the cold path (all-zero digit polynomials)
rng = np.random.default_rng(3)
small = rng.integers(0, 2**64, (16, p.small_size), dtype=np.uint64)
small[3, :-1] = 0
small[7] = 0
idx = (np.arange(16) % 4).astype(np.uint32)
same = np.array_equal(sk.pbs_batch(small, luts, idx), ks.pbs_batch(small, luts, idx, threads=8))
ok &= same
print("random / zero small LWEs:", same, flush=True)
# timing at full size
import torch  # noqa: E402

pf = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(1234)
skf = T.ServerKey(ctx, pf, rng.integers(0, 2**64, pf.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, pf.ksk_len, dtype=np.uint64))
skf.set_variant(variant)
dev = torch.device("cuda:0")
d_in = torch.from_numpy(rng.integers(0, 2**64, (B, pf.big_size), dtype=np.uint64).view(np.int64)).to(dev)
lf = np.stack([T.generate_lookup_table(pf, f)[0] for f in funcs])
d_luts = torch.from_numpy(lf.view(np.int64)).to(dev)
d_idx = torch.from_numpy((np.arange(B) % 4).astype(np.int32)).to(dev)
d_out = torch.empty_like(d_in)
for _ in range(2):
    skf.ks_pbs_batch_device(d_in, d_luts, d_idx, d_out)
torch.cuda.synchronize()
skf.profiling(True)
skf.profiling_read(0, reset=True)
t0 = time.perf_counter()
for _ in range(10):
    skf.ks_pbs_batch_device(d_in, d_luts, d_idx, d_out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
br, n = skf.profiling_read(1)
ksm, _ = skf.profiling_read(0)
print(f"variant {variant} B={B}: blind rotation {br / n:.3f} ms, keyswitch {ksm / n:.3f} ms, step {dt * 1e3:.3f} ms = {B / dt:.0f} PBS/s; status {skf.device_status()}")
sys.exit(0 if ok else 1)
