// carry_chain.hpp -- integer primitives shared by the lane-level sources of the blind-rotation kernels (pair_pbs.hpp,
// block_pbs.hpp): 64-bit additions with a per-lane carry-in written out as v_addc_co_u32 chains, and small bit helpers.
// Device build: inline assembly (gfx950); host emulation build (tests/emu, -DTFHE_HIP_HOST_EMU): plain C++ with the same
// results, one lane at a time.
#pragma once
#include "device_math.hpp"

namespace tfhe_hip {
namespace cc {

TFHE_DEV uint32_t f64_hi(double x)
{
    uint64_t b;
    __builtin_memcpy(&b, &x, 8);
    return (uint32_t)(b >> 32);
}
TFHE_DEV uint32_t f64_lo(double x)
{
    uint64_t b;
    __builtin_memcpy(&b, &x, 8);
    return (uint32_t)b;
}
// a & b & c in one instruction (v_bitop3_b32, gfx950)
TFHE_DEV uint32_t and3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(TFHE_HIP_HOST_EMU)
    return a & b & c;
#else
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x80);
#endif
}
// ---- 64-bit additions with a per-lane carry-in, four at a time ---------------------------------------------------
// The compiler has no way to express "add with the carry-in taken from a lane mask" (it emits two adds and a mov), so
// the carry chains of the rotation (hi word of x + a + cin) and of the accumulation (a += v + cin) are written out:
// v_addc_co_u32 with the carry-in / carry-out in SGPR pairs, FOUR independent chains interleaved per block -- gfx950
// needs two wait states between a VALU write of an SGPR and a VALU read of it (LLVM: VALUWriteSGPRVALURead), which the
// other three chains provide (the hazard recogniser does not look inside inline assembly).
#if defined(TFHE_HIP_HOST_EMU)
TFHE_DEV uint32_t alignbit(uint32_t hi, uint32_t lo, int sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh); }
TFHE_DEV bool add_wraps(uint32_t a, uint32_t b, uint32_t &sum) { sum = a + b; return sum < a; } // -> lane_mask
using lane_mask = bool; // one lane at a time
TFHE_DEV lane_mask mask_of(bool b) { return b; }
// d[j] = hi word of ((mv[j] ^ -p[j]) + a[j] + p[j]): the conditional negation of the rotation (p = polarity, see below)
TFHE_DEV void hi_of_signed_sums4(const uint64_t mv[4], const uint64_t a[4], const lane_mask p[4], uint32_t d[4])
{
    for (int j = 0; j < 4; j++) d[j] = (uint32_t)(((p[j] ? ~mv[j] : mv[j]) + a[j] + (p[j] ? 1u : 0u)) >> 32);
}
// a[j] += (vh[j]:vl[j]) + (t[j] != 0)
TFHE_DEV void add_to4(uint64_t *a0, uint64_t *a1, uint64_t *a2, uint64_t *a3, const uint32_t vl[4], const uint32_t vh[4], const uint32_t t[4])
{
    uint64_t *a[4] = {a0, a1, a2, a3};
    for (int j = 0; j < 4; j++) *a[j] += (((uint64_t)vh[j] << 32) | vl[j]) + (t[j] ? 1u : 0u);
}
// q[j] = hi word of ((x[j] ^ -p[j]) + a[j] + !p[j]) = hi word of ~(+-v - acc) for x = v - 1, a = acc - 1 (see pair_pbs.hpp);
// `pbar` = the complement of the polarity: it is the carry-in
TFHE_DEV void hi_of_complement_sums4(const uint64_t x[4], const uint64_t a[4], const lane_mask pbar[4], uint32_t q[4])
{
    for (int j = 0; j < 4; j++) q[j] = (uint32_t)(((pbar[j] ? x[j] : ~x[j]) + a[j] + (pbar[j] ? 1u : 0u)) >> 32);
}
// a[j] += F[j] << 12, F = the 52-bit fraction field of g[j] in [1, 2): the fixed-point image (units of 2^-64) of g - 1
TFHE_DEV void add_fraction4(uint64_t *a0, uint64_t *a1, uint64_t *a2, uint64_t *a3, const double g[4])
{
    uint64_t *a[4] = {a0, a1, a2, a3};
    for (int j = 0; j < 4; j++) {
        uint64_t b;
        __builtin_memcpy(&b, &g[j], 8);
        *a[j] += b << 12;
    }
}
TFHE_DEV double fract(double x) { return x - floor(x); } // exact for the values of the fast path (|x| >= 2: ulp >= 2^-51)
#else
TFHE_DEV uint32_t alignbit(uint32_t hi, uint32_t lo, int sh) { return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)sh); }
// sum = a + b (b wave-uniform), returns the lanes whose addition wrapped (written out: the compiler turns the carry of
// __builtin_add_overflow into a 0 / 1 register and compares it again before it accepts it as a mask)
TFHE_DEV uint64_t add_wraps(uint32_t a, uint32_t b, uint32_t &sum)
{
    uint64_t w;
    asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(sum), "=s"(w) : "v"(a), "s"(b));
    return w;
}
using lane_mask = uint64_t; // an SGPR pair: one bit per lane
TFHE_DEV lane_mask mask_of(bool b) { return __builtin_amdgcn_ballot_w64(b); }
TFHE_DEV void hi_of_signed_sums4(const uint64_t mv[4], const uint64_t a[4], const lane_mask p[4], uint32_t d[4])
{
    uint64_t m0 = p[0], m1 = p[1], m2 = p[2], m3 = p[3];
    uint32_t t0, t1, t2, t3;
    // t = -p ; d = t ^ mv.hi ; t ^= mv.lo ; (t, carry) = t + a.lo + p ; d = d + a.hi + carry
    asm("v_cndmask_b32_e64 %4, 0, -1, %8\n\t"
        "v_cndmask_b32_e64 %5, 0, -1, %9\n\t"
        "v_cndmask_b32_e64 %6, 0, -1, %10\n\t"
        "v_cndmask_b32_e64 %7, 0, -1, %11\n\t"
        "v_xor_b32_e32 %0, %4, %16\n\t"
        "v_xor_b32_e32 %1, %5, %17\n\t"
        "v_xor_b32_e32 %2, %6, %18\n\t"
        "v_xor_b32_e32 %3, %7, %19\n\t"
        "v_xor_b32_e32 %4, %4, %12\n\t"
        "v_xor_b32_e32 %5, %5, %13\n\t"
        "v_xor_b32_e32 %6, %6, %14\n\t"
        "v_xor_b32_e32 %7, %7, %15\n\t"
        "v_addc_co_u32_e64 %4, %8, %4, %20, %8\n\t"
        "v_addc_co_u32_e64 %5, %9, %5, %21, %9\n\t"
        "v_addc_co_u32_e64 %6, %10, %6, %22, %10\n\t"
        "v_addc_co_u32_e64 %7, %11, %7, %23, %11\n\t"
        "v_addc_co_u32_e64 %0, %8, %0, %24, %8\n\t"
        "v_addc_co_u32_e64 %1, %9, %1, %25, %9\n\t"
        "v_addc_co_u32_e64 %2, %10, %2, %26, %10\n\t"
        "v_addc_co_u32_e64 %3, %11, %3, %27, %11"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "+s"(m0), "+s"(m1),
          "+s"(m2), "+s"(m3)
        : "v"((uint32_t)mv[0]), "v"((uint32_t)mv[1]), "v"((uint32_t)mv[2]), "v"((uint32_t)mv[3]), "v"((uint32_t)(mv[0] >> 32)),
          "v"((uint32_t)(mv[1] >> 32)), "v"((uint32_t)(mv[2] >> 32)), "v"((uint32_t)(mv[3] >> 32)), "v"((uint32_t)a[0]),
          "v"((uint32_t)a[1]), "v"((uint32_t)a[2]), "v"((uint32_t)a[3]), "v"((uint32_t)(a[0] >> 32)), "v"((uint32_t)(a[1] >> 32)),
          "v"((uint32_t)(a[2] >> 32)), "v"((uint32_t)(a[3] >> 32)));
}
TFHE_DEV void add_to4(uint64_t *a0, uint64_t *a1, uint64_t *a2, uint64_t *a3, const uint32_t vl[4], const uint32_t vh[4], const uint32_t t[4])
{
    uint64_t m0, m1, m2, m3;
    uint32_t l0 = (uint32_t)*a0, l1 = (uint32_t)*a1, l2 = (uint32_t)*a2, l3 = (uint32_t)*a3;
    uint32_t h0 = (uint32_t)(*a0 >> 32), h1 = (uint32_t)(*a1 >> 32), h2 = (uint32_t)(*a2 >> 32), h3 = (uint32_t)(*a3 >> 32);
    asm("v_cmp_ne_u32_e64 %8, 0, %20\n\t"
        "v_cmp_ne_u32_e64 %9, 0, %21\n\t"
        "v_cmp_ne_u32_e64 %10, 0, %22\n\t"
        "v_cmp_ne_u32_e64 %11, 0, %23\n\t"
        "v_addc_co_u32_e64 %0, %8, %0, %12, %8\n\t"
        "v_addc_co_u32_e64 %1, %9, %1, %13, %9\n\t"
        "v_addc_co_u32_e64 %2, %10, %2, %14, %10\n\t"
        "v_addc_co_u32_e64 %3, %11, %3, %15, %11\n\t"
        "v_addc_co_u32_e64 %4, %8, %4, %16, %8\n\t"
        "v_addc_co_u32_e64 %5, %9, %5, %17, %9\n\t"
        "v_addc_co_u32_e64 %6, %10, %6, %18, %10\n\t"
        "v_addc_co_u32_e64 %7, %11, %7, %19, %11"
        : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(vl[0]), "v"(vl[1]), "v"(vl[2]), "v"(vl[3]), "v"(vh[0]), "v"(vh[1]), "v"(vh[2]), "v"(vh[3]), "v"(t[0]), "v"(t[1]), "v"(t[2]),
          "v"(t[3]));
    *a0 = ((uint64_t)h0 << 32) | l0;
    *a1 = ((uint64_t)h1 << 32) | l1;
    *a2 = ((uint64_t)h2 << 32) | l2;
    *a3 = ((uint64_t)h3 << 32) | l3;
}
// q = hi word of ((x ^ -p) + a + !p), the masks given as pbar = !p: t = pbar ? 0 : -1 ; q = t ^ x.hi ; t ^= x.lo ;
// (t, carry) = t + a.lo + pbar ; q = q + a.hi + carry
TFHE_DEV void hi_of_complement_sums4(const uint64_t x[4], const uint64_t a[4], const lane_mask pbar[4], uint32_t q[4])
{
    uint64_t m0 = pbar[0], m1 = pbar[1], m2 = pbar[2], m3 = pbar[3];
    uint32_t t0, t1, t2, t3;
    asm("v_cndmask_b32_e64 %4, -1, 0, %8\n\t"
        "v_cndmask_b32_e64 %5, -1, 0, %9\n\t"
        "v_cndmask_b32_e64 %6, -1, 0, %10\n\t"
        "v_cndmask_b32_e64 %7, -1, 0, %11\n\t"
        "v_xor_b32_e32 %0, %4, %16\n\t"
        "v_xor_b32_e32 %1, %5, %17\n\t"
        "v_xor_b32_e32 %2, %6, %18\n\t"
        "v_xor_b32_e32 %3, %7, %19\n\t"
        "v_xor_b32_e32 %4, %4, %12\n\t"
        "v_xor_b32_e32 %5, %5, %13\n\t"
        "v_xor_b32_e32 %6, %6, %14\n\t"
        "v_xor_b32_e32 %7, %7, %15\n\t"
        "v_addc_co_u32_e64 %4, %8, %4, %20, %8\n\t"
        "v_addc_co_u32_e64 %5, %9, %5, %21, %9\n\t"
        "v_addc_co_u32_e64 %6, %10, %6, %22, %10\n\t"
        "v_addc_co_u32_e64 %7, %11, %7, %23, %11\n\t"
        "v_addc_co_u32_e64 %0, %8, %0, %24, %8\n\t"
        "v_addc_co_u32_e64 %1, %9, %1, %25, %9\n\t"
        "v_addc_co_u32_e64 %2, %10, %2, %26, %10\n\t"
        "v_addc_co_u32_e64 %3, %11, %3, %27, %11"
        : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "+s"(m0), "+s"(m1),
          "+s"(m2), "+s"(m3)
        : "v"((uint32_t)x[0]), "v"((uint32_t)x[1]), "v"((uint32_t)x[2]), "v"((uint32_t)x[3]), "v"((uint32_t)(x[0] >> 32)),
          "v"((uint32_t)(x[1] >> 32)), "v"((uint32_t)(x[2] >> 32)), "v"((uint32_t)(x[3] >> 32)), "v"((uint32_t)a[0]),
          "v"((uint32_t)a[1]), "v"((uint32_t)a[2]), "v"((uint32_t)a[3]), "v"((uint32_t)(a[0] >> 32)), "v"((uint32_t)(a[1] >> 32)),
          "v"((uint32_t)(a[2] >> 32)), "v"((uint32_t)(a[3] >> 32)));
}
// a += F << 12 with F = the fraction field of g in [1, 2): hi = alignbit(g.hi, g.lo, 20) (the exponent bits fall off the top),
// lo = g.lo << 12; four independent carry chains (the SGPR-write / VALU-read wait states, see above)
TFHE_DEV void add_fraction4(uint64_t *a0, uint64_t *a1, uint64_t *a2, uint64_t *a3, const double g[4])
{
    uint64_t m0, m1, m2, m3;
    uint32_t l0 = (uint32_t)*a0, l1 = (uint32_t)*a1, l2 = (uint32_t)*a2, l3 = (uint32_t)*a3;
    uint32_t h0 = (uint32_t)(*a0 >> 32), h1 = (uint32_t)(*a1 >> 32), h2 = (uint32_t)(*a2 >> 32), h3 = (uint32_t)(*a3 >> 32);
    uint32_t u0, u1, u2, u3, w0, w1, w2, w3;
    asm("v_alignbit_b32 %8, %21, %20, 20\n\t"
        "v_alignbit_b32 %9, %23, %22, 20\n\t"
        "v_alignbit_b32 %10, %25, %24, 20\n\t"
        "v_alignbit_b32 %11, %27, %26, 20\n\t"
        "v_lshlrev_b32_e32 %12, 12, %20\n\t"
        "v_lshlrev_b32_e32 %13, 12, %22\n\t"
        "v_lshlrev_b32_e32 %14, 12, %24\n\t"
        "v_lshlrev_b32_e32 %15, 12, %26\n\t"
        "v_add_co_u32_e64 %0, %16, %0, %12\n\t"
        "v_add_co_u32_e64 %1, %17, %1, %13\n\t"
        "v_add_co_u32_e64 %2, %18, %2, %14\n\t"
        "v_add_co_u32_e64 %3, %19, %3, %15\n\t"
        "v_addc_co_u32_e64 %4, %16, %4, %8, %16\n\t"
        "v_addc_co_u32_e64 %5, %17, %5, %9, %17\n\t"
        "v_addc_co_u32_e64 %6, %18, %6, %10, %18\n\t"
        "v_addc_co_u32_e64 %7, %19, %7, %11, %19"
        : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3),
          "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(f64_lo(g[0])), "v"(f64_hi(g[0])), "v"(f64_lo(g[1])), "v"(f64_hi(g[1])), "v"(f64_lo(g[2])), "v"(f64_hi(g[2])),
          "v"(f64_lo(g[3])), "v"(f64_hi(g[3])));
    *a0 = ((uint64_t)h0 << 32) | l0;
    *a1 = ((uint64_t)h1 << 32) | l1;
    *a2 = ((uint64_t)h2 << 32) | l2;
    *a3 = ((uint64_t)h3 << 32) | l3;
}
TFHE_DEV double fract(double x) { return __builtin_amdgcn_fract(x); }
#endif

// -digit of the one-level signed decomposition of d (base 2^BASE_LOG), from the high word of q = ~d:
// closest_representable (decomposer.rs:97-118) + the balanced digit (decomposition.rs:78-86) collapse to a rounding
// addition and a signed bit field: -digit = sbfe(q.hi + 2^(31 - B), 32 - B, B)
template <int BASE_LOG> TFHE_DEV double neg_digit_from_hi(uint32_t q_hi)
{
    static_assert(BASE_LOG >= 2 && BASE_LOG <= 30, "digit must fit 32 bits");
    constexpr int sh = 32 - BASE_LOG - 1;
    return (double)((int32_t)(q_hi + (1u << sh)) >> (32 - BASE_LOG));
}

} // namespace cc
} // namespace tfhe_hip
