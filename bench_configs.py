#!/usr/bin/env python3
"""Secondary measurements: the BASELINE.json configs other than the headline batch (which is
bench.py): latency of one KS-PBS, FheString::eq 64/64, FheString::contains 256/8 and FheUint32 mul,
each through the host op-graph layer (include/tfhe_hip_strings.h) on ONE GPU.  Synthetic data (random
keys / ciphertexts: work is data-independent; correctness is covered by tests/test_gpu_ops.py).
Prints one JSON line per config.  Usage: python bench_configs.py [--reps 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import tfhe_rs_string_amd as T

    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
    rng = np.random.default_rng(7)
    ctx = T.Context(0)
    sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
    key = T.StringServerKey(sk.handle, p)
    lut, _ = T.generate_lookup_table(p, lambda x: x % 4)

    def rnd(*shape):
        return rng.integers(0, 2**64, shape + (p.big_size,), dtype=np.uint64)

    def timed(fn):
        fn()  # warm-up
        key.stats(reset=True)
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            r = fn()  # kept until the clock is read: freeing a 100 MB result is not part of the operation
            ts.append(time.perf_counter() - t0)
            del r
        st = key.stats()
        return float(np.median(ts)) * 1e3, st["pbs"] // args.reps, st["launches"] // args.reps

    def emit(name, ms, pbs, launches, ref):
        print(json.dumps({"config": name, "ms": round(ms, 3), "pbs": pbs, "launches": launches, "n_gpus": 1,
                          "pbs_per_s": round(pbs / ms * 1e3, 1) if pbs else None, "reference": ref, "data": "synthetic"}), flush=True)

    one = rnd(1)
    sk.ks_pbs_batch(one, lut, 0)  # warm-up: the first call allocates the page-locked staging buffer
    ts = []
    for _ in range(max(20, args.reps)):
        t0 = time.perf_counter()
        sk.ks_pbs_batch(one, lut, 0)
        ts.append(time.perf_counter() - t0)
    emit("single shortint KS-PBS (latency, host buffers; median of %d)" % len(ts), float(np.median(ts)) * 1e3, 1, 1,
         "16.6 ms single thread m6i.metal (tfhe/docs/getting_started/benchmarks.md:42)")
    s1, s2 = rnd(64, 4), rnd(64, 4)
    emit("FheString::eq, two 64-char strings", *timed(lambda: key.eq(s1, s2)),
         "64-step sequential AND chain, >= 1.1 s critical path (fhe_strings/server_key/comparisons.rs:184-212)")
    hay, needle = rnd(256, 4), rnd(8, 4)
    emit("FheString::contains, 256-char haystack / 8-char needle", *timed(lambda: key.contains(hay, needle)),
         "256 sequential ORs + 8-deep AND chains, >= 4.5 s critical path (fhe_strings/server_key/contains.rs:77-91)")
    a, b = rnd(16), rnd(16)
    emit("FheUint32 mul (16 blocks)", *timed(lambda: key.mul_parallelized(a, b)),
         "333 ms on m6i.metal, all cores (tfhe/docs/getting_started/benchmarks.md:17)")
    # batch mode: independent operations staged together share their launches (throughput regime)
    col = [rnd(64, 4) for _ in range(64)]

    def eq_column():
        with key.batch():
            return [key.eq(c, s1) for c in col]

    ms, pbs, launches = timed(eq_column)
    emit("64 x FheString::eq (64-char strings) in one batch", ms, pbs, launches, "9.5 ms each when issued one by one")
    col2 = [rnd(32, 4) for _ in range(32)]

    def contains_column():
        with key.batch():
            return [key.contains(c, needle[:4]) for c in col2]

    ms, pbs, launches = timed(contains_column)
    emit("32 x FheString::contains (32-char strings, 4-char needle) in one batch", ms, pbs, launches, None)
    # beyond BASELINE.json: a few of the other string operations on the same key
    s40, pat = rnd(40, 4), rnd(3, 4)
    emit("FheString::le, two 40-char strings", *timed(lambda: key.le(s40, rnd(40, 4))), None)
    emit("FheString::find, 40-char string / 3-char pattern", *timed(lambda: key.find(s40, pat)), None)
    emit("FheString::to_uppercase, 40 chars", *timed(lambda: key.to_uppercase(s40)), None)
    emit("FheString::trim, 40 chars", *timed(lambda: key.trim(s40)), None)
    emit("FheString::push_padding_to_end, 40 chars", *timed(lambda: key.push_padding_to_end(s40)), None)
    emit("FheString::split, 40 chars by a clear char", *timed(lambda: key.split(s40, ",")),
         "41 sequential find + substring rounds (fhe_strings/server_key/split.rs:314-345)")
    emit("FheString::splitn(4), 40 chars by a clear 2-char pattern", *timed(lambda: key.splitn(s40, 4, ", ")), None)
    emit("FheString::split_ascii_whitespace, 40 chars", *timed(lambda: key.split_ascii_whitespace(s40)), None)
    emit("FheString::replace, 40 chars, clear 2-char pattern -> clear 3 chars", *timed(lambda: key.replace(s40, ", ", " - ")),
         "split + re-join of 41 parts (fhe_strings/server_key/replace.rs:20-52)")
    emit("FheString::substring with encrypted bounds, 40 chars", *timed(lambda: key.substring(s40, rnd(4), rnd(4))), None)


if __name__ == "__main__":
    main()
