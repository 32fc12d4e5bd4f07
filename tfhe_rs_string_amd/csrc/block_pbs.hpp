// block_pbs.hpp -- lane-level source of the WORKGROUP-per-ciphertext blind rotation (variant 3).
//
// The latency variant: one 512-thread workgroup (8 wavefronts, two on every SIMD of a CU) per
// ciphertext, so a CMUX step of ONE ciphertext runs on a whole CU instead of on two wavefronts
// (variant 2) -- about a third of the latency for batches that do not fill the chip (<= 256
// ciphertexts = one per CU).  Wavefronts 0-3 own GLWE polynomial 0 (the mask), 4-7 polynomial 1
// (the body); thread T = 64 q + l of a group holds FOUR folded points, i.e. one radix-4 butterfly
// per stage:
//
//   stage 1 (L=1024)  thread T : positions T + 256 t            in registers, straight from the
//                                                               rotated accumulator
//   -- workgroup barrier: block m of 256 positions moves to wavefront q = m --
//   stage 2 (L=256)   lane l   : 256 q + l + 64 t
//   stage 3 (L=64)    lane l   : 256 q + 64 (l >> 4) + (l & 15) + 16 t       wave-local: 2 -> 3 by permlane
//   stage 4 (L=16)    lane l   : 256 q + 16 (l >> 2) + (l & 3) + 4 t         swaps, 3 -> 4 through LDS
//   -- workgroup barrier: both Fourier polynomials visible to both groups --
//   stage 5 (L=4)     lane l   : 256 q + 4 l + t    for BOTH polynomials (redundantly in the two
//                     groups: 32 additions instead of a third exchange), external product column
//                     p, stage 5' in registers
//   stages 4', 3', 2' wave-local; barrier; stage 1', backward twist, rounding, accumulate.
//
// Every stage works in place on thread-private position sets, so the only cross-lane hazards are
// between consecutive stages.  The accumulator stays in registers (8 coefficients per thread); its
// one's-complement mirror for the next rotation is a third, linear image.
// LDS: three 32 KiB images ([polynomial][1024] x 16 B); the two transform images are XOR-swizzled so
// that every access pattern above is conflict-free (lane groups of MI355X_MICROARCH.md, section LDS).
//
// The arithmetic DAG is exactly DESIGN.md section 4 (bit-exact against the oracle): the twiddle of
// butterfly j of stage s, output m, is the table entry W[(n/L) j m] for every j (the oracle skips
// the multiplication for j = 0, which only differs in the sign of zeros).
// The file is a sequence of lane-local phases like wave_pbs.hpp / pair_pbs.hpp, so tests/emu can
// run the same source thread by thread on the CPU.
#pragma once
#include "wave_pbs.hpp"
#include "carry_chain.hpp"

namespace tfhe_hip {
namespace blk {

using wave::FN;
using wave::LOG2N;
using wave::N;
using wave::digit_from_hi;
using wave::ld_c64;
using wave::nacc_sub_fast; // (block_mb_pbs.hpp keeps the one's-complement accumulator of wave_pbs.hpp)
using wave::st_c64;
using wave::torus_shift;
using wave::torus_slow;

// Integer conventions as in pair_pbs.hpp (exact identities): registers and mirror hold coefficient - 1, the digits come out
// negated, the Fourier key is stored multiplied by -2^-10, so that the backward twist yields the torus increment in
// revolutions and fract(w) + 1 carries it in its fraction bits (carry_chain.hpp).
constexpr double BSK_PRESCALE = -0.0009765625;
#if defined(TFHE_HIP_HOST_EMU)
inline uint64_t torus_slow_scaled(double w) { return torus_from_f64(w); }
#else
__device__ __attribute__((noinline)) uint64_t torus_slow_scaled(double w) { return torus_from_f64(w); }
#endif

constexpr int THREADS = 512;
constexpr int IMG_BYTES = FN * 16;     // one polynomial image: 1024 slots of 16 B
constexpr int BUF_F = 0;               // forward transform [p 2][1024] c64
constexpr int BUF_M = 2 * IMG_BYTES;   // inverse transform [p 2][1024] c64
constexpr int BUF_MIR = 4 * IMG_BYTES; // accumulator mirror [p 2][1024] x (~coef P, ~coef P + 1024), LINEAR: the
                                       // rotation reads 64 consecutive slots from an arbitrary offset
constexpr int LDS_BYTES = 6 * IMG_BYTES; // 98 304
// PACKED layout (blind_rotate_half_kernel, two workgroups per CU): the mirror lives in the 16-B slots of the inverse image
// -- a thread writes the new accumulator words into exactly the slots it has just read its stage-1' inputs from; the
// rotation of the next step reads them (swizzled addresses) before that image is written again behind the second
// barrier -- so a ciphertext needs four images = 64 KiB.
constexpr int LDS_BYTES_PACKED = 4 * IMG_BYTES; // 65 536

// slot of position P in the transform images: bits 3, 4, 5 of P are folded into the low four bits so that
// every access pattern of the header is conflict-free (ds_read_b128 lane groups and the 128-byte store
// rows of MI355X_MICROARCH.md, section LDS; checked by enumeration)
TFHE_DEV constexpr int swz(int P)
{
    const int b = (P >> 4) & 3;
    return P ^ ((((0 - b) & 3) << 2) | b) ^ (((P >> 3) & 1) * 3);
}
TFHE_DEV constexpr int slot(int P) { return swz(P) << 4; }
// byte offset of the mirror slot of position P of polynomial p
template <bool PACKED> TFHE_DEV int mirror_off(int p, int P)
{
    return PACKED ? BUF_M + p * IMG_BYTES + slot(P) : BUF_MIR + p * IMG_BYTES + P * 16;
}

// Device BSK layout: [i][column p][wavefront q][e = row * 4 + t][lane 64] c64,
// value = Fourier GGSW i, row, column p at DIF position 256 q + 4 lane + t
TFHE_DEV constexpr size_t bsk_block_index(size_t i, int p, int q, int e, int lane)
{
    return ((((i * 2 + p) * 4 + q) * 8 + e) * 64) + lane;
}

struct Ids {
    int p, q, l, T; // polynomial / column, wavefront in the group, lane, thread in the group
};
TFHE_DEV Ids make_ids(int tid)
{
    Ids d;
    d.p = tid >> 8;
    d.q = (tid >> 6) & 3;
    d.l = tid & 63;
    d.T = tid & 255;
    return d;
}

struct Lane {
    uint64_t acc[8]; // acc[2t + hh] = coefficient (T + 256 t) + 1024 hh of polynomial p, MINUS ONE
    c64 R[4], S[4];  // working set (S: the other polynomial in stage 5)
    c64 tw[4];       // twist of positions T + 256 t
    c64 w1[3], w2[3], w3[3], w4[3]; // this thread's twiddles of stages 1-4, outputs m = 1..3
    c64 G[8];        // bootstrap-key prefetch: rows 0 / 1 of column p at the thread's stage-5 positions
};

struct u64x2 {
    uint64_t lo, hi;
};
TFHE_DEV u64x2 ld_u64x2(const char *base, int off)
{
    const c64 v = ld_c64(base, off);
    u64x2 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}
TFHE_DEV void st_u64x2(char *base, int off, uint64_t lo, uint64_t hi)
{
    c64 v;
    const u64x2 r{lo, hi};
    __builtin_memcpy(&v, &r, 16);
    st_c64(base, off, v);
}

// stage bases (position of t = 0) of a thread
TFHE_DEV int base2(const Ids &d) { return 256 * d.q + d.l; }
TFHE_DEV int base3(const Ids &d) { return 256 * d.q + 64 * (d.l >> 4) + (d.l & 15); }
TFHE_DEV int base4(const Ids &d) { return 256 * d.q + 16 * (d.l >> 2) + (d.l & 3); }
TFHE_DEV int base5(const Ids &d) { return 256 * d.q + 4 * d.l; }

// ---- phases -----------------------------------------------------------------------------------

// acc = lut * X^{-b_hat} (polynomial_algorithms.rs:315-354), twiddles of the thread, first mirror image
template <bool PACKED = false>
TFHE_DEV void phase_init(Lane &s, char *lds, const uint64_t *lut, const c64 *twist, const c64 *W, uint32_t b_hat,
                         const Ids &d)
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
            s.acc[2 * t + hh] = (ng ? (uint64_t)0 - v : v) - 1; // registers hold acc - 1
        }
        s.tw[t] = twist[d.T + 256 * t];
        st_u64x2(lds, mirror_off<PACKED>(d.p, d.T + 256 * t), s.acc[2 * t], s.acc[2 * t + 1]);
    }
#pragma unroll
    for (int m = 1; m < 4; m++) {
        s.w1[m - 1] = W[d.T * m];
        s.w2[m - 1] = W[4 * d.l * m];
        s.w3[m - 1] = W[16 * (d.l & 15) * m];
        s.w4[m - 1] = W[64 * (d.l & 3) * m];
    }
}

TFHE_DEV void ggsw_prefetch(Lane &s, const c64 *bsk_b, size_t i, const Ids &d)
{
    const c64 *g = bsk_b + bsk_block_index(i, d.p, d.q, 0, d.l);
#pragma unroll
    for (int e = 0; e < 8; e++) s.G[e] = g[e * 64];
}

// ct1 = acc * X^{a_hat} - acc (polynomial_algorithms.rs:425-490), one-digit signed decomposition
// (ggsw.rs:515-517), fold + forward twist (x86.rs:562-563), stage 1, store for the block exchange.
// Coefficients c and c + 1024 of the rotated polynomial come from ONE mirror slot: with
// u = (c - a_hat) mod 2N, slot u mod 1024 holds source coefficients (u mod 1024, u mod 1024 + 1024);
// bit 10 of u says which of the two is c's, bit 11 is the negacyclic sign.
template <int BASE_LOG, bool PACKED = false> TFHE_DEV void phase_rotate_stage1(Lane &s, char *lds, uint32_t a_hat, const Ids &d)
{
    u64x2 v[4];
    uint32_t u[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        u[t] = ((uint32_t)(d.T + 256 * t) - a_hat) & (2 * N - 1);
        v[t] = ld_u64x2(lds, mirror_off<PACKED>(d.p, (int)(u[t] & (FN - 1))));
    }
    // registers and mirror hold x = coefficient - 1: ~d = ~(+-v - acc) = (x_v ^ -p) + x_acc + !p with p = !sign (carry_chain.hpp)
    uint32_t dhi[8];
#pragma unroll
    for (int g = 0; g < 2; g++) {
        uint64_t x[4], a[4];
        cc::lane_mask pbar[4];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int t = 2 * g + j;
            const bool hs = (u[t] >> 10) & 1u, neg0 = ((u[t] >> 11) & 1u) != 0;
            x[2 * j] = hs ? v[t].hi : v[t].lo;
            x[2 * j + 1] = hs ? v[t].lo : v[t].hi;
            a[2 * j] = s.acc[2 * t];
            a[2 * j + 1] = s.acc[2 * t + 1];
            pbar[2 * j] = cc::mask_of(neg0);
            pbar[2 * j + 1] = cc::mask_of(neg0 != hs); // sign1 = sign0 ^ hs
        }
        cc::hi_of_complement_sums4(x, a, pbar, dhi + 4 * g);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        c64 dg; // the NEGATED digits (the key's prescale carries the sign)
        dg.re = cc::neg_digit_from_hi<BASE_LOG>(dhi[2 * t]);
        dg.im = cc::neg_digit_from_hi<BASE_LOG>(dhi[2 * t + 1]);
        s.R[t] = cmul(dg, s.tw[t]);
    }
    bfly4_fwd(s.R[0], s.R[1], s.R[2], s.R[3]);
    char *f = lds + BUF_F + d.p * IMG_BYTES;
    st_c64(f, slot(d.T), s.R[0]);
#pragma unroll
    for (int m = 1; m < 4; m++) st_c64(f, slot(d.T + 256 * m), cmul(s.R[m], s.w1[m - 1]));
}

// one forward stage on positions base + stride * t: load / butterfly + twiddles / store
template <int STRIDE, class LaneT> TFHE_DEV void fwd_load(LaneT &s, const char *img, int base)
{
#pragma unroll
    for (int t = 0; t < 4; t++) s.R[t] = ld_c64(img, slot(base + STRIDE * t));
}
template <class LaneT> TFHE_DEV void fwd_bfly(LaneT &s, const c64 *w)
{
    bfly4_fwd(s.R[0], s.R[1], s.R[2], s.R[3]);
#pragma unroll
    for (int m = 1; m < 4; m++) s.R[m] = cmul(s.R[m], w[m - 1]);
}
template <int STRIDE, class LaneT> TFHE_DEV void store4(const LaneT &s, char *img, int base)
{
#pragma unroll
    for (int t = 0; t < 4; t++) st_c64(img, slot(base + STRIDE * t), s.R[t]);
}
// stage 2 leaves its outputs (positions base2 + 64 m) in registers; the 4 x 4 transposition between
// the register index m and lane bits (5,4) -- two v_permlane32_swap / v_permlane16_swap rounds, no
// LDS -- turns them into the inputs of stage 3 (positions base3 + 16 t)
template <class LaneT> TFHE_DEV void phase_fwd2(LaneT &s, char *lds, const Ids &d)
{
    fwd_load<64>(s, lds + BUF_F + d.p * IMG_BYTES, base2(d));
    fwd_bfly(s, s.w2);
}
//   [transpose54]
template <class LaneT> TFHE_DEV void phase_fwd3(LaneT &s, char *lds, const Ids &d)
{
    fwd_bfly(s, s.w3);
    store4<16>(s, lds + BUF_F + d.p * IMG_BYTES, base3(d));
}
template <class LaneT> TFHE_DEV void phase_fwd4(LaneT &s, char *lds, const Ids &d)
{
    char *img = lds + BUF_F + d.p * IMG_BYTES;
    fwd_load<4>(s, img, base4(d));
    fwd_bfly(s, s.w4);
    store4<4>(s, img, base4(d));
}

// stage 5 of both polynomials, column p of the external product (ggsw.rs:477-598: row 0 `mul`, row 1
// `mul_add`), stage 5' of the result, stored into the inverse image
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
    for (int t = 0; t < 4; t++) s.R[t] = cmul_add(s.G[4 + t], s.S[t], cmul(s.G[t], s.R[t]));
    bfly4_inv(s.R[0], s.R[1], s.R[2], s.R[3]);
    char *m = lds + BUF_M + d.p * IMG_BYTES;
#pragma unroll
    for (int t = 0; t < 4; t++) st_c64(m, slot(b + t), s.R[t]);
}

// The same for a thread that holds the positions of BOTH groups (blind_rotate_half_kernel): stage 5 of the two Fourier
// polynomials once, both columns of the external product, stage 5' of each, stored into the two inverse images
TFHE_DEV void phase_pointwise_both(Lane &s0, Lane &s1, char *lds, const Ids &d0)
{
    const int b = base5(d0);
    c64 F0[4], F1[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        F0[t] = ld_c64(lds + BUF_F, slot(b + t));
        F1[t] = ld_c64(lds + BUF_F + IMG_BYTES, slot(b + t));
    }
    bfly4_fwd(F0[0], F0[1], F0[2], F0[3]);
    bfly4_fwd(F1[0], F1[1], F1[2], F1[3]);
#pragma unroll
    for (int t = 0; t < 4; t++) {
        s0.R[t] = cmul_add(s0.G[4 + t], F1[t], cmul(s0.G[t], F0[t]));
        s1.R[t] = cmul_add(s1.G[4 + t], F1[t], cmul(s1.G[t], F0[t]));
    }
    bfly4_inv(s0.R[0], s0.R[1], s0.R[2], s0.R[3]);
    bfly4_inv(s1.R[0], s1.R[1], s1.R[2], s1.R[3]);
#pragma unroll
    for (int t = 0; t < 4; t++) {
        st_c64(lds + BUF_M, slot(b + t), s0.R[t]);
        st_c64(lds + BUF_M + IMG_BYTES, slot(b + t), s1.R[t]);
    }
}

// one inverse stage on positions base + stride * t
template <int STRIDE, class LaneT> TFHE_DEV void inv_load(LaneT &s, const char *img, int base)
{
#pragma unroll
    for (int t = 0; t < 4; t++) s.R[t] = ld_c64(img, slot(base + STRIDE * t));
}
template <class LaneT> TFHE_DEV void inv_bfly(LaneT &s, const c64 *w)
{
#pragma unroll
    for (int m = 1; m < 4; m++) s.R[m] = cmul_conj(s.R[m], w[m - 1]);
    bfly4_inv(s.R[0], s.R[1], s.R[2], s.R[3]);
}
template <class LaneT> TFHE_DEV void phase_inv4(LaneT &s, char *lds, const Ids &d)
{
    char *img = lds + BUF_M + d.p * IMG_BYTES;
    inv_load<4>(s, img, base4(d));
    inv_bfly(s, s.w4);
    store4<4>(s, img, base4(d));
}
template <class LaneT> TFHE_DEV void phase_inv3(LaneT &s, char *lds, const Ids &d)
{
    inv_load<16>(s, lds + BUF_M + d.p * IMG_BYTES, base3(d));
    inv_bfly(s, s.w3);
}
//   [transpose54]
template <class LaneT> TFHE_DEV void phase_inv2(LaneT &s, char *lds, const Ids &d)
{
    inv_bfly(s, s.w2);
    store4<64>(s, lds + BUF_M + d.p * IMG_BYTES, base2(d));
}

// stage 1', backward twist (x86.rs:842-861), rounding to the torus, accumulate; the new accumulator
// words go to the mirror image for the next rotation
template <bool PACKED = false> TFHE_DEV void phase_inv1_accumulate(Lane &s, char *lds, const Ids &d)
{
    char *m = lds + BUF_M + d.p * IMG_BYTES;
    inv_load<256>(s, m, d.T);
    inv_bfly(s, s.w1);
    // w = x * conj(tw) carries the key's prescale: the torus increment in revolutions; fast path |w| >= 2: see pair_pbs.hpp
    c64 v[4];
    uint32_t ok = 0xFFFFFFFFu;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        v[t] = cmul_conj(s.R[t], s.tw[t]);
        ok = cc::and3(ok, cc::f64_hi(v[t].re), cc::f64_hi(v[t].im));
    }
    if (__builtin_expect((ok & 0x40000000u) == 0, 0)) { // some |w| < 2: the reference formula, out of line
#pragma unroll
        for (int t = 0; t < 4; t++) {
            s.acc[2 * t] += torus_slow_scaled(v[t].re);
            s.acc[2 * t + 1] += torus_slow_scaled(v[t].im);
        }
    } else {
#pragma unroll
        for (int g = 0; g < 2; g++) {
            double f[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int e = 4 * g + j;
                f[j] = cc::fract((e & 1) ? v[e >> 1].im : v[e >> 1].re) + 1.0;
            }
            cc::add_fraction4(&s.acc[4 * g], &s.acc[4 * g + 1], &s.acc[4 * g + 2], &s.acc[4 * g + 3], f);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) st_u64x2(lds, mirror_off<PACKED>(d.p, d.T + 256 * t), s.acc[2 * t], s.acc[2 * t + 1]);
}

// sample extraction at X^0 (glwe_sample_extraction.rs:125-146): group 0 writes the mask, group 1 the body
// mode 0: the registers hold the coefficients; 2: the coefficients minus one
template <class LaneT> TFHE_DEV void phase_sample_extract_plain(const LaneT &s, uint64_t *out, const Ids &d, int mode)
{
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int c = d.T + 256 * t + FN * hh;
            const uint64_t v = mode == 2 ? s.acc[2 * t + hh] + 1 : s.acc[2 * t + hh];
            if (d.p == 0) out[c == 0 ? 0 : N - c] = (c == 0) ? v : (uint64_t)0 - v;
            else if (c == 0) out[N] = v;
        }
}
TFHE_DEV void phase_sample_extract(const Lane &s, uint64_t *out, const Ids &d) { phase_sample_extract_plain(s, out, d, 2); }

} // namespace blk
} // namespace tfhe_hip
