// device_math.hpp -- scalar device helpers shared by every kernel (gfx950 only).
//
// Build rule: device code is compiled with -ffp-contract=off; every fused multiply-add is an
// explicit fma().  The arithmetic conventions below are the engine's FFT/rounding specification
// (DESIGN.md section 4); file:line references are to /root/reference/tfhe/src.
#pragma once
#include <stdint.h>
#if defined(TFHE_HIP_HOST_EMU)
// Host emulation build (tests/emu only): the same lane-level source runs lane by lane on the CPU
// so that the index maps and the arithmetic DAG can be checked against the oracle without a GPU.
#include <cmath>
#define TFHE_DEV inline
namespace tfhe_hip {
using std::floor;
using std::fma;
using std::rint;
} // namespace tfhe_hip
#else
#include <hip/hip_runtime.h>
#define TFHE_DEV __device__ __forceinline__
#endif

namespace tfhe_hip {

// ---- device-side failure reporting ---------------------------------------------------------------
// Bits of a server key's device-visible error word (page-locked host memory mapped into the device): a kernel
// that cannot produce a correct result sets a bit and carries on to the end of the launch (a trap would take
// the whole process down); every host entry point checks the word (api.cpp, check_device_status).
constexpr uint32_t ERR_PAIR_TIMEOUT = 1u; // a wave of the wave-pair kernel ran out of its hand-off spin
constexpr uint32_t ERR_LUT_IDX = 2u;      // lut_idx[ct] >= n_luts (the accumulator of table 0 was used instead)
#if !defined(TFHE_HIP_HOST_EMU)
TFHE_DEV void report_error(uint32_t *err, uint32_t code)
{
    if (err) __hip_atomic_fetch_or(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// lut_idx[ct], range-checked: an index past the tables would make the kernel read out of bounds
TFHE_DEV uint32_t checked_lut_idx(const uint32_t *lut_idx, int ct, uint32_t n_luts, uint32_t *err)
{
    const uint32_t v = lut_idx[ct];
    if (v < n_luts) return v;
    report_error(err, ERR_LUT_IDX);
    return 0;
}
#endif

struct alignas(16) c64 {
    double re, im;
};

// ---- integer pieces -------------------------------------------------------------------------

// core_crypto/commons/math/decomposition/decomposer.rs:97-118
TFHE_DEV uint64_t closest_representable(uint64_t x, int base_log, int level)
{
    const int shift = 64 - level * base_log - 1;
    uint64_t r = x >> shift;
    r += 1;
    r &= ~(uint64_t)1;
    return r << shift;
}

// decomposition/iter.rs:120-127 == fft64/math/decomposition.rs:78-86
TFHE_DEV int64_t decompose_one_level(int base_log, uint64_t &state, uint64_t mask)
{
    uint64_t res = state & mask;
    state >>= base_log;
    uint64_t carry = ((res - 1) | state) & res;
    carry >>= base_log - 1;
    state += carry;
    return (int64_t)(res - (carry << base_log));
}

// fft_impl/common.rs:26-43 ; result in [0, 2N]
TFHE_DEV uint32_t modswitch(uint64_t x, int log2N)
{
    uint64_t o = x >> (64 - log2N - 2);
    o += 1;
    o >>= 1;
    return (uint32_t)o;
}

// ---- f64 <-> integer ------------------------------------------------------------------------

// signed 64-bit -> f64, correctly rounded (one rounding in the fma), as cvtsi2sd does on the CPU
TFHE_DEV double i64_to_f64(int64_t x)
{
    const int32_t hi = (int32_t)(x >> 32);
    const uint32_t lo = (uint32_t)x;
    return fma((double)hi, 4294967296.0, (double)lo);
}

// integer-valued t, |t| <= 2^63 -> t mod 2^64 (fft64/math/fft/x86.rs:25-81 wraps +2^63 to i64::MIN)
TFHE_DEV uint64_t f64_to_u64_wrap(double t)
{
    t = (t >= 9223372036854775808.0) ? -9223372036854775808.0 : t;
    const double h = floor(t * 2.3283064365386962890625e-10); // 2^-32, exact
    const double l = fma(h, -4294967296.0, t);                // exact, in [0, 2^32)
    const uint32_t lo = (uint32_t)l;
    const int32_t hi = (int32_t)h;
    return ((uint64_t)(uint32_t)hi << 32) | (uint64_t)lo;
}

// commons/math/torus/mod.rs:72-78 in the rounding the x86 build executes (x86.rs:864-871):
// x - rne(x), * 2^64, rne, wrap to u64
TFHE_DEV uint64_t torus_from_f64(double x)
{
    double fr = x - rint(x);
    fr = rint(fr * 18446744073709551616.0);
    return f64_to_u64_wrap(fr);
}

// ---- complex kernels (operation order is part of the specification) ---------------------------

// y * w            (forward twist x86.rs:562-563, FFT twiddles)
TFHE_DEV c64 cmul(c64 y, c64 w)
{
    c64 r;
    r.re = fma(y.re, w.re, -(y.im * w.im));
    r.im = fma(y.re, w.im, y.im * w.re);
    return r;
}
// x * conj(w)      (backward twist x86.rs:842-861, inverse FFT twiddles)
TFHE_DEV c64 cmul_conj(c64 x, c64 w)
{
    c64 r;
    r.re = fma(x.re, w.re, x.im * w.im);
    r.im = fma(-x.re, w.im, x.im * w.re);
    return r;
}
// a * b + c        (update_with_fmadd, fft64/crypto/ggsw.rs:616-697; pulp's kernel is not in the tree)
TFHE_DEV c64 cmul_add(c64 a, c64 b, c64 c)
{
    c64 r;
    r.re = fma(a.re, b.re, -fma(a.im, b.im, -c.re));
    r.im = fma(a.re, b.im, fma(a.im, b.re, c.im));
    return r;
}

// radix-4 DIF butterfly (no twiddles): inputs a,b,c,d -> y0..y3
TFHE_DEV void bfly4_fwd(c64 &a, c64 &b, c64 &c, c64 &d)
{
    const double t0r = a.re + c.re, t0i = a.im + c.im, t1r = a.re - c.re, t1i = a.im - c.im;
    const double t2r = b.re + d.re, t2i = b.im + d.im, t3r = b.re - d.re, t3i = b.im - d.im;
    a.re = t0r + t2r; a.im = t0i + t2i;
    c.re = t0r - t2r; c.im = t0i - t2i;
    b.re = t1r + t3i; b.im = t1i - t3r;
    d.re = t1r - t3i; d.im = t1i + t3r;
}
// radix-4 DIT (inverse) butterfly (inputs already multiplied by conj twiddles)
TFHE_DEV void bfly4_inv(c64 &a, c64 &b, c64 &c, c64 &d)
{
    const double t0r = a.re + c.re, t0i = a.im + c.im, t1r = a.re - c.re, t1i = a.im - c.im;
    const double t2r = b.re + d.re, t2i = b.im + d.im, t3r = b.re - d.re, t3i = b.im - d.im;
    a.re = t0r + t2r; a.im = t0i + t2i;
    c.re = t0r - t2r; c.im = t0i - t2i;
    b.re = t1r - t3i; b.im = t1i + t3r;
    d.re = t1r + t3i; d.im = t1i - t3r;
}

} // namespace tfhe_hip
