// block_mb_pbs.hpp -- lane-level source of the MULTI-BIT blind rotation (workgroup per ciphertext).
//
// Multi-bit PBS (core_crypto/algorithms/lwe_multi_bit_programmable_bootstrapping.rs:18-545): the LWE mask
// is consumed G elements at a time (grouping factor G = 2 or 3); a group's 2^G Fourier GGSWs are combined
//     M = ggsw_0 + sum_{idx >= 1} factor_idx * (ggsw_idx * mono_idx)
// (prepare_multi_bit_ggsw_mem_optimized, :18-81) and the accumulator becomes  0 + M (x) acc  (:478-506).
// X^d has the Fourier image factor * mono with mono[pos] = W[(d' * freq(pos)) mod n], d' the folded degree,
// factor = +-(1 or i) * twist[d'] (fft/mod.rs:408-445); see oracle/tfhe_oracle.c (orc_mb_*) for the exact
// arithmetic, which this file reproduces bit for bit.
//
// Same thread layout, LDS images and transform phases as block_pbs.hpp (variant 3); what changes:
//   * no rotation: the digits come straight from the thread's own accumulator registers (no mirror image,
//     one workgroup barrier less per step);
//   * the external-product column is formed from the combined GGSW: per thread 8 values (2 rows x its 4
//     stage-5 positions), per non-trivial subset 8 bootstrap-key values + 4 table roots + a uniform factor;
//   * the result REPLACES the accumulator.
// n / G steps instead of n (409 / 296 instead of 742 at the 2_2 parameter sets).
#pragma once
#include "block_pbs.hpp"

namespace tfhe_hip {
namespace blkmb {

using namespace blk;

// Device BSK layout: [group][idx 2^G][column p][wavefront q][e = row * 4 + t][lane 64] c64,
// value = Fourier GGSW (group, idx), row, column p at DIF position 256 q + 4 lane + t
template <int G> TFHE_DEV constexpr size_t bsk_mb_index(size_t grp, int idx, int p, int q, int e, int lane)
{
    return (((((grp << G) + idx) * 2 + p) * 4 + q) * 8 + e) * 64 + lane;
}

struct Lane {
    uint64_t acc[8]; // acc[2t + hh] = coefficient (T + 256 t) + 1024 hh of polynomial p (plain, not complemented)
    c64 R[4], S[4];
    c64 tw[4];
    c64 w1[3], w2[3], w3[3], w4[3];
    c64 M[8];          // combined GGSW column p: rows 0 / 1 at the thread's stage-5 positions
    c64 P[12];         // one subset in flight: its 8 bootstrap-key values and 4 roots ...
    c64 Pf;            // ... and its factor
    uint32_t freq[4];  // natural frequency index of those positions (base-4 digit reversal)
};

TFHE_DEV uint32_t freq_of_pos(uint32_t pos)
{
    uint32_t f = 0;
#pragma unroll
    for (int s = 0; s < 5; s++) {
        f = (f << 2) | (pos & 3u);
        pos >>= 2;
    }
    return f;
}

// acc = lut * X^{-b_hat}; thread constants
TFHE_DEV void phase_init(Lane &s, const uint64_t *lut, const c64 *twist, const c64 *W, uint32_t b_hat, const Ids &d)
{
    const int rem = b_hat & (N - 1), cyc = b_hat >> LOG2N;
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int c = d.T + 256 * t + FN * hh;
            int src = c + rem, ng = cyc & 1;
            if (src >= N) { src -= N; ng ^= 1; }
            const uint64_t v = lut[d.p * N + src];
            s.acc[2 * t + hh] = ng ? (uint64_t)0 - v : v;
        }
        s.tw[t] = twist[d.T + 256 * t];
        s.freq[t] = freq_of_pos((uint32_t)(base5(d) + t));
    }
#pragma unroll
    for (int m = 1; m < 4; m++) {
        s.w1[m - 1] = W[d.T * m];
        s.w2[m - 1] = W[4 * d.l * m];
        s.w3[m - 1] = W[16 * (d.l & 15) * m];
        s.w4[m - 1] = W[64 * (d.l & 3) * m];
    }
}

// ggsw_0 of the group: the initial value of the combined GGSW
template <int G> TFHE_DEV void ggsw0_prefetch(Lane &s, const c64 *bsk_mb, size_t grp, const Ids &d)
{
    const c64 *g = bsk_mb + bsk_mb_index<G>(grp, 0, d.p, d.q, 0, d.l);
#pragma unroll
    for (int e = 0; e < 8; e++) s.M[e] = g[e * 64];
}

// one-digit signed decomposition of the thread's own coefficients, fold + forward twist, stage 1, store
template <int BASE_LOG> TFHE_DEV void phase_decompose_stage1(Lane &s, char *lds, const Ids &d)
{
#pragma unroll
    for (int t = 0; t < 4; t++) {
        c64 dg;
        dg.re = digit_from_hi<BASE_LOG>((uint32_t)(s.acc[2 * t] >> 32));
        dg.im = digit_from_hi<BASE_LOG>((uint32_t)(s.acc[2 * t + 1] >> 32));
        s.R[t] = cmul(dg, s.tw[t]);
    }
    bfly4_fwd(s.R[0], s.R[1], s.R[2], s.R[3]);
    char *f = lds + BUF_F + d.p * IMG_BYTES;
    st_c64(f, slot(d.T), s.R[0]);
#pragma unroll
    for (int m = 1; m < 4; m++) st_c64(f, slot(d.T + 256 * m), cmul(s.R[m], s.w1[m - 1]));
}

// folded degree and factor of X^degree (fft/mod.rs:408-445); `degree` already modulus-switched (< 2N or == 2N)
TFHE_DEV c64 monomial_factor(const c64 *twist, uint32_t degree, uint32_t &folded)
{
    const bool negate = (degree >> LOG2N) & 1u;
    uint32_t dd = degree & (N - 1);
    const bool imag = dd >= (uint32_t)FN;
    if (imag) dd -= FN;
    folded = dd;
    const c64 t = twist[dd];
    c64 r;
    r.re = imag ? -t.im : t.re;
    r.im = imag ? t.re : t.im;
    if (negate) { r.re = -r.re; r.im = -r.im; }
    return r;
}

// M += factor * (ggsw_idx * mono) for a non-trivial subset idx of the group's mask elements
// (prepare_multi_bit_ggsw_mem_optimized, :47-78): subset idx selects mask element m iff bit (G-1-m) of idx is set.
// Split into the issue of the loads and their use, so that the kernel can keep one subset in flight
// behind the transform phases (the combined GGSW of group grp + 1 is built between the external
// products of groups grp and grp + 1: slot K of mb_slot).
template <int G> TFHE_DEV void combine_issue(Lane &s, const c64 *bsk_mb, size_t grp, int idx, const uint64_t *mask_elems,
                                             const c64 *twist, const c64 *W, const Ids &d)
{
    uint64_t deg = 0;
#pragma unroll
    for (int m = 0; m < G; m++)
        if ((idx >> (G - 1 - m)) & 1) deg += mask_elems[m];
    uint32_t folded;
    s.Pf = monomial_factor(twist, modswitch(deg, LOG2N), folded);
    const c64 *g = bsk_mb + bsk_mb_index<G>(grp, idx, d.p, d.q, 0, d.l);
#pragma unroll
    for (int e = 0; e < 8; e++) s.P[e] = g[e * 64];
#pragma unroll
    for (int t = 0; t < 4; t++) s.P[8 + t] = W[(folded * s.freq[t]) & (uint32_t)(FN - 1)];
}
TFHE_DEV void combine_consume(Lane &s)
{
#pragma unroll
    for (int e = 0; e < 8; e++) s.M[e] = cmul_add(s.Pf, cmul(s.P[e], s.P[8 + (e & 3)]), s.M[e]);
}
// slot K (0..7) of the build of group grp's combined GGSW.  Step j fetches ggsw_0 and the first subset (j = 0) or
// uses subset j and fetches subset j + 1.  With 7 subsets (G = 3) every slot is a step; with 3 subsets (G = 2) only
// the even slots are, which leaves two transform phases between the issue of a subset's loads and their use.
template <int G, int K> TFHE_DEV void mb_slot(Lane &s, const c64 *bsk_mb, size_t grp, const uint64_t *lwe, const c64 *twist,
                                              const c64 *W, const Ids &d)
{
    constexpr int NSUB = (1 << G) - 1;
    constexpr int STRIDE = NSUB <= 3 ? 2 : 1;
    if (K % STRIDE != 0) return;
    constexpr int J = K / STRIDE;
    if (J == 0) ggsw0_prefetch<G>(s, bsk_mb, grp, d);
    if (J >= 1 && J <= NSUB) combine_consume(s);
    if (J + 1 <= NSUB) combine_issue<G>(s, bsk_mb, grp, J + 1, lwe + grp * G, twist, W, d);
}

// stage 5 of both polynomials, column p of the external product with the combined GGSW, stage 5', store
TFHE_DEV void phase_pointwise(Lane &s, char *lds, const Ids &d)
{
    const int b = base5(d);
#pragma unroll
    for (int t = 0; t < 4; t++) {
        s.R[t] = ld_c64(lds + BUF_F, slot(b + t));
        s.S[t] = ld_c64(lds + BUF_F + IMG_BYTES, slot(b + t));
    }
    bfly4_fwd(s.R[0], s.R[1], s.R[2], s.R[3]);
    bfly4_fwd(s.S[0], s.S[1], s.S[2], s.S[3]);
#pragma unroll
    for (int t = 0; t < 4; t++) s.R[t] = cmul_add(s.M[4 + t], s.S[t], cmul(s.M[t], s.R[t]));
    bfly4_inv(s.R[0], s.R[1], s.R[2], s.R[3]);
    char *m = lds + BUF_M + d.p * IMG_BYTES;
#pragma unroll
    for (int t = 0; t < 4; t++) st_c64(m, slot(b + t), s.R[t]);
}

// stage 1', backward twist, rounding to the torus: the result replaces the accumulator (dst = 0 + ...)
TFHE_DEV void phase_inv1_replace(Lane &s, char *lds, const Ids &d)
{
    inv_load<256>(s, lds + BUF_M + d.p * IMG_BYTES, d.T);
    inv_bfly(s, s.w1);
    c64 v[4];
    uint32_t sh[8], slow = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        v[t] = cmul_conj(s.R[t], s.tw[t]);
        sh[2 * t] = torus_shift<54>(v[t].re);
        sh[2 * t + 1] = torus_shift<54>(v[t].im);
        slow |= (sh[2 * t] > 63u) | (sh[2 * t + 1] > 63u);
    }
    if (__builtin_expect(slow != 0, 0)) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            s.acc[2 * t] = torus_slow(v[t].re);
            s.acc[2 * t + 1] = torus_slow(v[t].im);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            uint64_t n0 = ~(uint64_t)0, n1 = ~(uint64_t)0; // nacc_sub_fast works on the one's complement of 0
            nacc_sub_fast(n0, v[t].re, sh[2 * t]);
            nacc_sub_fast(n1, v[t].im, sh[2 * t + 1]);
            s.acc[2 * t] = ~n0;
            s.acc[2 * t + 1] = ~n1;
        }
    }
}

} // namespace blkmb
} // namespace tfhe_hip
