"""Builds the HIP extension in-tree: tfhe_rs_string_amd/libtfhe_hip.so (gfx950 only).

hipcc cross-compiles without a GPU.  Device code is built with -ffp-contract=off: the engine's
bit-exactness against the oracle depends on every fused multiply-add being an explicit fma().
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libtfhe_hip.so")
OBJ = os.path.join(CSRC, "obj")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXX = os.environ.get("CXX", "g++")
COMMON = ["-O3", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-std=c++17"] + os.environ.get("TFHE_HIP_EXTRA_FLAGS", "").split()

# the latency kernels are short dependent phases between barriers: LLVM's max-ILP scheduling strategy interleaves their independent
# butterflies better than the default (2.62 -> 2.52 ms for 256 bootstraps, 4.49 -> 4.46 ms for 512, multi-bit 2.53 -> 2.47 / 3.12 ->
# 3.07 ms, alternating A/B on one box;
# no change for the throughput kernel, whose phases are explicit software pipelines)
MAX_ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]

SOURCES = [
    ("twiddle.cpp", [CXX] + COMMON),
    ("api.cpp", [HIPCC] + COMMON + ["-Wall", "-Wno-unused-value"]),
    ("kernels.hip", [HIPCC, "--offload-arch=gfx950"] + COMMON + ["-Wall", "-Wno-unused-function"]),
    ("keyswitch_mfma.hip", [HIPCC, "--offload-arch=gfx950"] + COMMON + ["-Wall", "-Wno-unused-function"]),
    ("blind_rotate_wave.hip", [HIPCC, "--offload-arch=gfx950"] + COMMON + ["-Wall", "-Wno-unused-function"]),
    ("blind_rotate_pair.hip", [HIPCC, "--offload-arch=gfx950"] + COMMON + ["-Wall", "-Wno-unused-function"]),
    ("blind_rotate_block.hip", [HIPCC, "--offload-arch=gfx950"] + COMMON + MAX_ILP + ["-Wall", "-Wno-unused-function"]),
    ("blind_rotate_quad.hip", [HIPCC, "--offload-arch=gfx950"] + COMMON + MAX_ILP + ["-Wall", "-Wno-unused-function"]),
    ("blind_rotate_block_mb.hip", [HIPCC, "--offload-arch=gfx950"] + COMMON + MAX_ILP + ["-Wall", "-Wno-unused-function"]),
    ("csprng.hip", [HIPCC, "--offload-arch=gfx950"] + COMMON + ["-Wall", "-Wno-unused-function"]),
    ("exact_ntt.hip", [HIPCC, "--offload-arch=gfx950"] + COMMON + ["-Wall", "-Wno-unused-function"]),
    ("host_api.cpp", [HIPCC] + COMMON + ["-Wall"]),
    ("wire.cpp", [HIPCC] + COMMON + ["-Wall"]),
]


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    deps += [os.path.join(HERE, "..", "include", "tfhe_hip.h")] + list(extra)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    objs = []
    rebuilt = False
    for name, cmd in SOURCES:
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            continue
        obj = os.path.join(OBJ, name + ".o")
        objs.append(obj)
        if force or _newer(src, obj):
            full = cmd + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(full), flush=True)
            subprocess.check_call(full)
            rebuilt = True
    if rebuilt or not os.path.exists(OUT):
        full = [HIPCC, "-shared", "-o", OUT] + objs + ["-fvisibility=hidden"]
        if verbose:
            print(" ".join(full), flush=True)
        subprocess.check_call(full)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
