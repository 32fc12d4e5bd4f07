"""SURVEY 8f-4, throughput half: the exact-integer external product against the f64 one.
CPU (always): the oracle's two implementations on one thread.  GPU (if a device is present): csrc/exact_ntt.hip at B = 1024
against the production blind rotation's time per CMUX step (bench.py's kernel / 742).  Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import pyoracle as O  # noqa: E402

L = O.lib()
ks = O.KeySet(O.param_message_2_carry_2(lwe_dim=4), seed=3)
p = ks.p
N = p.poly_size
rng = np.random.default_rng(1)
ggsw = np.ascontiguousarray(ks.bsk[:4 * N])
t = L.orc_ntt_new(N)
prepared = np.zeros(L.orc_ntt_ggsw_len(C.byref(p)), dtype=np.uint64)
L.orc_ntt_ggsw_prepare(C.byref(p), t, ggsw, prepared)
glwe_in = rng.integers(0, 2**64, 2 * N, dtype=np.uint64)
acc = rng.integers(0, 2**64, 2 * N, dtype=np.uint64)
re, im = np.ascontiguousarray(ks.bsk_re[:4 * (N // 2)]), np.ascontiguousarray(ks.bsk_im[:4 * (N // 2)])
reps = 200
t0 = time.perf_counter()
for _ in range(reps):
    L.orc_ntt_add_external_product(C.byref(p), t, acc, prepared, glwe_in)
cpu_exact = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
for _ in range(reps):
    L.orc_add_external_product(C.byref(p), ks.fft.h, acc, re, im, glwe_in)
cpu_f64 = (time.perf_counter() - t0) / reps
out = {"experiment": "exact-integer (NTT mod 2^64 - 2^32 + 1, 3 x 22-bit limbs) external product vs the f64 FFT one",
       "cpu_exact_us": cpu_exact * 1e6, "cpu_f64_us": cpu_f64 * 1e6, "cpu_ratio": cpu_exact / cpu_f64}
try:
    import tfhe_rs_string_amd as T

    B = 1024
    gin = rng.integers(0, 2**64, (B, 2 * N), dtype=np.uint64)
    a0 = rng.integers(0, 2**64, (B, 2 * N), dtype=np.uint64)
    _, ms = T.exact_external_product_batch(ggsw, p.pbs_base_log, gin, a0, timing_reps=20)
    out.update({"gpu_exact_ms_per_1024_products": ms, "gpu_exact_products_per_s": B / ms * 1e3})
    prod = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r02_bench.json")))
    step_ms = prod["roofline"]["avg_launch_ms"] / 742
    out.update({"gpu_f64_ms_per_1024_products": step_ms, "gpu_ratio": ms / step_ms,
                "gpu_f64_source": "profiles/r02_bench.json roofline.avg_launch_ms / 742 CMUX steps"})
except Exception as e:  # no device here
    out["gpu"] = f"not measured: {e}"
print(json.dumps(out))
