// wave_pbs.hpp -- lane-level source of the wave-per-ciphertext blind rotation (variant 1).
//
// One 64-lane wavefront owns one ciphertext for the whole CMUX chain; the accumulator and all FFT
// data live in its registers, the only shared-memory traffic is one transposition per FFT and the
// staging of the accumulator for the monomial rotation.  No workgroup barrier is needed after
// start-up (everything a wave touches in LDS is private to it), so waves never wait for each other.
//
// Lane layout: lane = 32*h + l.  Half h (0/1) owns GLWE polynomial h (mask / body): the same
// instruction stream transforms both polynomials at once, 32 folded complex points per lane.
//
//   pass A (natural order)   lane l holds points q = l + 32 r, r = 0..31   (registers R[r])
//        stage 1 (L=1024), stage 2 (L=256), first radix-2 layer of stage 3 (pairs 32 apart)
//   transposition through LDS
//   pass B                   lane l' = 2*blk + par holds two 16-point blocks of 64-block `blk`
//        second layer of stage 3 (+ stage-3 twiddles), stage 4 (L=16), stage 5 (L=4)
//
// The arithmetic DAG is exactly the radix-4 DIF/DIT schedule of DESIGN.md section 4 (the oracle's
// fft_fwd / fft_inv): only where values live differs.  The file is written as a sequence of
// lane-local "phases" separated by the points where lanes exchange data (LDS direction changes,
// permlane swaps); on the GPU a wave runs the phases back to back, the host emulation
// (tests/emu, -DTFHE_HIP_HOST_EMU) runs each phase for lanes 0..63 in turn.
//
// Reference functions restated (paths relative to /root/reference/tfhe/src):
//   fft64/crypto/bootstrap.rs:242-331 blind_rotate_assign, ggsw.rs:477-598 add_external_product_assign,
//   polynomial_algorithms.rs:315-354, 425-490, fft64/math/fft/mod.rs:398-405, 487-557 (+x86.rs twins),
//   glwe_sample_extraction.rs:91-147.
#pragma once
#include "device_math.hpp"

#if defined(TFHE_HIP_HOST_EMU) || defined(TFHE_NO_SCHED_FENCE) // (the latter: scheduling experiments only)
#define TFHE_SCHED_FENCE() ((void)0)
#else
// Pins the position of the loads issued before it relative to the arithmetic after it: with one
// wave per SIMD nothing else hides LDS / L2 latency, so every phase below is written as an explicit
// software pipeline (issue the next chunk's loads, fence, compute the current chunk).
#define TFHE_SCHED_FENCE() asm volatile("" ::: "memory")
#endif

namespace tfhe_hip {
namespace wave {

constexpr int N = 2048;  // polynomial size
constexpr int FN = 1024; // folded complex points
constexpr int LOG2N = 11;
#ifndef TFHE_GRING
#define TFHE_GRING 4
#endif
constexpr int GRING = TFHE_GRING; // bootstrap-key prefetch ring depth, in Fourier slots (4 x 16 B per slot and lane)

// ---- LDS layout (bytes) ---------------------------------------------------------------------
// per wave: two polynomial scratch areas; each holds either the u64 mirror of one accumulator
// polynomial (2048 * 8 B) or the transposition buffer (1024 padded 16-B entries).
constexpr int POLY_BYTES = 16768;
constexpr int WAVE_BYTES = 2 * POLY_BYTES;
constexpr int WAVES_PER_WG = 4;
// tables first: their offsets then fit the 16-bit immediate offset field of the DS instructions
constexpr int TAB_TW = 0;                          // twist   [r 32][l 32]            16 KiB
constexpr int TAB_T1 = TAB_TW + 32 * 32 * 16;      // stage 1 [c 8][t 3][l 32]        12 KiB
constexpr int TAB_T3 = TAB_T1 + 8 * 3 * 32 * 16;   // stage 3 [par 2][j 32]            1 KiB
constexpr int TAB_BYTES = TAB_T3 + 2 * 32 * 16;    // 29 696
constexpr int WAVE0 = TAB_BYTES;                   // per-wave scratch areas follow
constexpr int LDS_BYTES = WAVE0 + WAVES_PER_WG * WAVE_BYTES; // 163 840 - 1 024 = 162 816
static_assert(LDS_BYTES <= 163840, "LDS budget of one CU");

// global-memory copy of the tables (same byte layout as the LDS image, followed by T2 and WC)
constexpr int GTAB_T2 = TAB_BYTES;                 // stage 2 [c 2][t 3][l 32] c64      3 KiB
constexpr int GTAB_WC = GTAB_T2 + 2 * 3 * 32 * 16; // W[64 m], m < 16                   256 B
constexpr int GTAB_BYTES = GTAB_WC + 16 * 16;

// transposition entry (16-B units) of (64-block, kind t0..t3, index in 16-block)
TFHE_DEV constexpr int tr_entry(int blk64, int kind, int jj) { return blk64 * 65 + kind * 16 + jj; }

// Oracle (DIF position order) index of Fourier slot (lane l' , register j) after pass B
TFHE_DEV constexpr int fourier_pos(int lp, int j) { return 64 * (lp >> 1) + 16 * ((lp & 1) + 2 * (j >> 4)) + (j & 15); }
// Device BSK layout: [i][e = j*4 + (row*2+col)][lane 64] c64 ; lane = 32*h' + l' handles slots j + 16 h'
TFHE_DEV constexpr int bsk_src_pos(int lane, int jloc) { return fourier_pos(lane & 31, jloc + 16 * (lane >> 5)); }

// ---- memory helpers -------------------------------------------------------------------------
// Returns x unchanged but opaque to the optimiser (no instruction is emitted).  Used on per-lane
// LDS base offsets: without it the compiler hoists one address register per unrolled access out of
// the CMUX loop (hundreds of VGPRs -> scratch spills) instead of using base + immediate offsets.
TFHE_DEV int opaque(int x)
{
#if !defined(TFHE_HIP_HOST_EMU)
    asm volatile("" : "+v"(x));
#endif
    return x;
}

TFHE_DEV c64 ld_c64(const char *base, int off) { return *reinterpret_cast<const c64 *>(base + off); }
TFHE_DEV void st_c64(char *base, int off, c64 v) { *reinterpret_cast<c64 *>(base + off) = v; }
TFHE_DEV uint64_t ld_u64(const char *base, int off) { return *reinterpret_cast<const uint64_t *>(base + off); }
TFHE_DEV void st_u64(char *base, int off, uint64_t v) { *reinterpret_cast<uint64_t *>(base + off) = v; }
TFHE_DEV double ld_f64(const char *base, int off) { return *reinterpret_cast<const double *>(base + off); }

// wave-uniform constants, passed by value as kernel arguments so that they live in SGPRs:
// W[64], W[128], W[192], W[384] -- the twiddles of stage 4 / 4'
struct WaveConsts {
    double w[8];
};

struct Lane {
    uint64_t acc[64]; // acc[2r + hh] = ONE'S COMPLEMENT of coefficient (l + 32 r) + 1024 hh of polynomial h
    c64 R[32];        // FFT working set
    c64 G[4 * GRING]; // bootstrap-key prefetch ring (GRING Fourier slots x 4 GGSW polynomials)
    int t2_off[2];    // LDS byte offset of this lane's stage-2 twiddles inside the stage-1 table
};

// ---- radix-4 stages that are uniform across lanes (pass B, inside one 16-point block) ---------
TFHE_DEV c64 mul_mi(c64 y) // y * (0, -1) == y * W[256]
{
    c64 r; r.re = y.im; r.im = -y.re; return r;
}
TFHE_DEV c64 mul_pi(c64 y) // y * conj(W[256]) = y * (0, +1)
{
    c64 r; r.re = -y.im; r.im = y.re; return r;
}
TFHE_DEV c64 neg(c64 y)
{
    c64 r; r.re = -y.re; r.im = -y.im; return r;
}
TFHE_DEV c64 wc(const WaveConsts &k, int i)
{
    c64 r; r.re = k.w[2 * i]; r.im = k.w[2 * i + 1]; return r;
}

// stage 4 (L=16, Q=4) + stage 5 (L=4, Q=1) of the forward transform on B[0..15]
template <int OFF, class LaneT> TFHE_DEV void fwd_block16(LaneT &s, const WaveConsts &k)
{
    c64 *B = s.R + OFF;
    const c64 w1 = wc(k, 0), w2 = wc(k, 1), w3 = wc(k, 2), w6 = wc(k, 3); // W[64], W[128], W[192], W[384]
    // jj = 0: no twiddles
    bfly4_fwd(B[0], B[4], B[8], B[12]);
    // jj = 1: W[64], W[128], W[192]
    bfly4_fwd(B[1], B[5], B[9], B[13]);
    B[5] = cmul(B[5], w1); B[9] = cmul(B[9], w2); B[13] = cmul(B[13], w3);
    // jj = 2: W[128], W[256], W[384]
    bfly4_fwd(B[2], B[6], B[10], B[14]);
    B[6] = cmul(B[6], w2); B[10] = mul_mi(B[10]); B[14] = cmul(B[14], w6);
    // jj = 3: W[192], W[384], W[576] = -W[64]
    bfly4_fwd(B[3], B[7], B[11], B[15]);
    B[7] = cmul(B[7], w3); B[11] = cmul(B[11], w6); B[15] = neg(cmul(B[15], w1));
#pragma unroll
    for (int g = 0; g < 4; g++) bfly4_fwd(B[4 * g], B[4 * g + 1], B[4 * g + 2], B[4 * g + 3]);
}

template <int OFF, class LaneT> TFHE_DEV void inv_block16(LaneT &s, const WaveConsts &k)
{
    c64 *B = s.R + OFF;
    const c64 w1 = wc(k, 0), w2 = wc(k, 1), w3 = wc(k, 2), w6 = wc(k, 3);
#pragma unroll
    for (int g = 0; g < 4; g++) bfly4_inv(B[4 * g], B[4 * g + 1], B[4 * g + 2], B[4 * g + 3]);
    bfly4_inv(B[0], B[4], B[8], B[12]);
    B[5] = cmul_conj(B[5], w1); B[9] = cmul_conj(B[9], w2); B[13] = cmul_conj(B[13], w3);
    bfly4_inv(B[1], B[5], B[9], B[13]);
    B[6] = cmul_conj(B[6], w2); B[10] = mul_pi(B[10]); B[14] = cmul_conj(B[14], w6);
    bfly4_inv(B[2], B[6], B[10], B[14]);
    B[7] = cmul_conj(B[7], w3); B[11] = cmul_conj(B[11], w6); B[15] = neg(cmul_conj(B[15], w1));
    bfly4_inv(B[3], B[7], B[11], B[15]);
}

// signed decomposition to ONE digit of base 2^BASE_LOG (level count 1), from the high word only:
// closest_representable (decomposer.rs:97-118) + decompose_one_level (decomposition.rs:78-86) collapse
// to 32-bit arithmetic because only bits >= 64 - BASE_LOG - 1 of the input matter.
template <int BASE_LOG> TFHE_DEV double digit_from_hi(uint32_t d_hi)
{
    static_assert(BASE_LOG >= 2 && BASE_LOG <= 30, "digit must fit 32 bits");
    // state = closest_representable(d) >> (64 - BASE_LOG) = (d_hi + 2^sh) >> (sh + 1), sh = 31 - BASE_LOG;
    // res = state mod B; digit = res - B * [res > B/2]  ==  ((state + K) mod B) - K with K = B/2 - 1.
    // Adding K << (sh + 1) before the shift commutes with it, so:
    constexpr int sh = 32 - BASE_LOG - 1;
    constexpr uint32_t K = (1u << (BASE_LOG - 1)) - 1u;
    constexpr uint32_t C = (1u << sh) + (K << (sh + 1));
    const uint32_t x = ((d_hi + C) >> (sh + 1)) & ((1u << BASE_LOG) - 1u);
    return (double)((int32_t)x - (int32_t)K);
}

// Subtracts RNE-int(y * 2^SH) (mod 2^64) from the one's-complemented accumulator word `nacc`
// (nacc = ~acc, so acc += t  <=>  nacc -= t).  Equals `acc += torus_from_f64(y * 2^(SH-64))`:
// (y*2^(SH-64) - rne(y*2^(SH-64))) * 2^64 rounded == RNE-int(y * 2^SH) minus a multiple of 2^64.
// Fast path: y = +-M * 2^(e-52) with 53-bit M, so y * 2^SH = +-(M << s), s = e - 52 + SH, exact for
// 0 <= s <= 63 (bits shifted past 2^64 drop out of the torus).  `s` from torus_shift().
template <int SH> TFHE_DEV uint32_t torus_shift(double y)
{
    uint64_t bits;
    __builtin_memcpy(&bits, &y, 8);
    return (((uint32_t)(bits >> 32) >> 20) & 0x7FFu) - (uint32_t)(1075 - SH);
}
// cold path (|y| < 1/4 or absurdly large): reference formula, kept out of line so that the unrolled
// hot loop stays inside the instruction cache
#if defined(TFHE_HIP_HOST_EMU)
inline uint64_t torus_slow(double y) { return torus_from_f64(y * 0.0009765625); }
#else
__device__ __attribute__((noinline)) uint64_t torus_slow(double y) { return torus_from_f64(y * 0.0009765625); }
#endif
TFHE_DEV void nacc_sub_fast(uint64_t &nacc, double y, uint32_t s)
{
    uint64_t bits;
    __builtin_memcpy(&bits, &y, 8);
    const uint64_t m = (bits & 0x000FFFFFFFFFFFFFull) | 0x0010000000000000ull;
    const uint64_t v = m << s;
    nacc += ((int64_t)bits < 0) ? v : (uint64_t)0 - v;
}

// =============================================================================================
// Phases.  `poly` = this lane's polynomial scratch (lds_wave + h * POLY_BYTES), `tab` = LDS base
// (table offsets TAB_* are absolute), `ggsw` = lane-ordered Fourier GGSW of this CMUX iteration.
// =============================================================================================

// acc = lut * X^{-b_hat}   (polynomial_algorithms.rs:315-354)
TFHE_DEV void phase_init(Lane &s, const uint64_t *lut, uint32_t b_hat, int lane)
{
    const int h = lane >> 5, l = lane & 31;
    const int rem = b_hat & (N - 1), cyc = b_hat >> LOG2N;
#pragma unroll
    for (int r = 0; r < 32; r++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int c = l + 32 * r + FN * hh;
            int src = c + rem, ng = cyc & 1;
            if (src >= N) { src -= N; ng ^= 1; }
            const uint64_t v = lut[h * N + src];
            s.acc[2 * r + hh] = ~(ng ? (uint64_t)0 - v : v); // registers hold the one's complement
        }
    // stage-2 twiddles W[4 (l + 32 c)(t+1)] are entries of the stage-1 table of another lane:
    // 4 (l + 32 c) = l2 + 32 c2 with l2 = 4l mod 32, c2 = 4l / 32 + 4c
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int l2 = (4 * l) & 31, c2 = ((4 * l) >> 5) + 4 * c;
        s.t2_off[c] = TAB_T1 + (c2 * 3 * 32 + l2) * 16;
    }
}

// issue the bootstrap-key loads of Fourier slots j0 .. j0+cnt-1 into the prefetch ring
template <int J0, int CNT> TFHE_DEV void ggsw_prefetch(Lane &s, const c64 *ggsw, int lane)
{
#pragma unroll
    for (int j = J0; j < J0 + CNT; j++)
#pragma unroll
        for (int rc = 0; rc < 4; rc++) s.G[(j % GRING) * 4 + rc] = ggsw[(j * 4 + rc) * 64 + lane];
}

// stage the accumulator polynomial in LDS for the rotation
TFHE_DEV void phase_mirror_write(const Lane &s, char *lds, int poly_off, int lane)
{
    char *p = lds + opaque(poly_off + (lane & 31) * 8);
#pragma unroll
    for (int r = 0; r < 32; r++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) st_u64(p, (32 * r + FN * hh) * 8, s.acc[2 * r + hh]);
}

// ct1 = acc * X^{a_hat} - acc (polynomial_algorithms.rs:425-490); one-digit signed decomposition
// (ggsw.rs:515-517, level count 1); fold + forward twist (convert_forward_integer, x86.rs:562-563).
// Software pipeline over 8 chunks of 4 points: loads of chunk k+1 are in flight while chunk k computes.
template <int BASE_LOG> TFHE_DEV void phase_rotate_decompose_twist(Lane &s, const char *lds, int poly_off, int lane,
                                                                   uint32_t a_hat)
{
    const int l = lane & 31;
    const char *poly = lds + poly_off;
    const char *tab = lds + opaque(l * 16);
    uint64_t rot[2][8];
    c64 tw[2][4];
    // src(c) = (c - a_hat) mod 2N with c = l + 32 t' (t' = r + 32 hh): the lane reads 64 consecutive
    // 32-coefficient blocks starting at block bq; byte address = ((b6 + t') mod 64) * 256 + br * 8 and
    // the sign flips with bit 6 of bq + t'.  Bit 14 of `a0 + t' * 256` is exactly that sign.
    const uint32_t base = ((uint32_t)l - a_hat) & (2 * N - 1);
    const uint32_t a0 = ((base >> 5) << 8) | ((base & 31) << 3); // (b6 << 8 | br << 3) + (s6 << 14)
    auto issue = [&](int k, int slot) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 4 * k + i;
#pragma unroll
            for (int hh = 0; hh < 2; hh++)
                rot[slot][2 * i + hh] = ld_u64(poly, (int)((a0 + (r + 32 * hh) * 256) & 0x3FFFu));
            tw[slot][i] = ld_c64(tab, TAB_TW + r * 32 * 16);
        }
    };
    issue(0, 0);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k < 7) issue(k + 1, (k + 1) & 1);
        TFHE_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 4 * k + i;
            c64 dg;
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                // mirror and registers hold one's complements (nv = ~v, nacc = ~acc):
                //   v - acc = ~nv + nacc + 1 ;  -v - acc = nv + nacc + 2
                const uint32_t ng = ((a0 + (r + 32 * hh) * 256) >> 14) & 1u;
                const uint64_t nv = rot[k & 1][2 * i + hh];
                const uint64_t d = (nv ^ ((uint64_t)ng - 1)) + s.acc[2 * r + hh] + (uint64_t)(1 + ng);
                (hh ? dg.im : dg.re) = digit_from_hi<BASE_LOG>((uint32_t)(d >> 32));
            }
            s.R[r] = cmul(dg, tw[k & 1][i]);
        }
    }
}

// stages 1, 2 and the first layer of stage 3; R[] then holds (t0|t2) in even r, (t1|t3) in odd r
TFHE_DEV void fwd_pass_a(Lane &s, const char *lds, int lane)
{
    const char *tab = lds + opaque((lane & 31) * 16);
    c64 w[2][6];
    auto issue1 = [&](int k, int slot) { // stage-1 twiddles of butterflies r0 = 2k, 2k+1
#pragma unroll
        for (int e = 0; e < 6; e++) w[slot][e] = ld_c64(tab, TAB_T1 + ((2 * k) * 3 + e) * 32 * 16);
    };
    issue1(0, 0);
#pragma unroll
    for (int k = 0; k < 4; k++) { // stage 1: L = 1024, Q = 256, jj = l + 32 r0
        if (k < 3) issue1(k + 1, (k + 1) & 1);
        else {
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int t = 0; t < 3; t++) w[0][c * 3 + t] = ld_c64(lds + s.t2_off[c], t * 512);
        }
        TFHE_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r0 = 2 * k + i;
            bfly4_fwd(s.R[r0], s.R[r0 + 8], s.R[r0 + 16], s.R[r0 + 24]);
            s.R[r0 + 8] = cmul(s.R[r0 + 8], w[k & 1][i * 3 + 0]);
            s.R[r0 + 16] = cmul(s.R[r0 + 16], w[k & 1][i * 3 + 1]);
            s.R[r0 + 24] = cmul(s.R[r0 + 24], w[k & 1][i * 3 + 2]);
        }
    }
#pragma unroll
    for (int blk = 0; blk < 4; blk++) // stage 2: L = 256, Q = 64, jj = l + 32 c ; twiddles in w[0]
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int b = 8 * blk + c;
            bfly4_fwd(s.R[b], s.R[b + 2], s.R[b + 4], s.R[b + 6]);
            s.R[b + 2] = cmul(s.R[b + 2], w[0][c * 3 + 0]);
            s.R[b + 4] = cmul(s.R[b + 4], w[0][c * 3 + 1]);
            s.R[b + 6] = cmul(s.R[b + 6], w[0][c * 3 + 2]);
        }
#pragma unroll
    for (int m = 0; m < 16; m++) { // stage 3, first layer: pairs 32 apart
        const c64 lo = s.R[2 * m], hi = s.R[2 * m + 1];
        s.R[2 * m].re = lo.re + hi.re; s.R[2 * m].im = lo.im + hi.im;
        s.R[2 * m + 1].re = lo.re - hi.re; s.R[2 * m + 1].im = lo.im - hi.im;
    }
}

TFHE_DEV void fwd_transpose_write(const Lane &s, char *lds, int poly_off, int lane)
{
    const int l = lane & 31;
    char *p = lds + opaque(poly_off + tr_entry(0, 2 * (l >> 4), l & 15) * 16);
#pragma unroll
    for (int r = 0; r < 32; r++) st_c64(p, tr_entry(r >> 1, r & 1, 0) * 16, s.R[r]);
}

// read pass-B inputs, second layer of stage 3 + its twiddles, stages 4 and 5
TFHE_DEV void fwd_pass_b(Lane &s, const char *lds, int poly_off, int lane, const WaveConsts &k)
{
    const int lp = lane & 31, par = lp & 1;
    const int swp = par * 8; // odd lanes read t3 with re/im swapped (t3 * -i = (im, -re))
    const int base = poly_off + tr_entry(lp >> 1, par, 0) * 16;
    const char *px = lds + opaque(base);
    const char *pr = lds + opaque(base + swp), *pi = lds + opaque(base + 8 - swp);
    const char *t3 = lds + opaque(TAB_T3 + par * 32 * 16);
    const double sg = par ? -1.0 : 1.0;
    constexpr int CH = 2; // points per pipeline chunk
    c64 X[2][CH], U[2][CH], wl[2][CH], wh[2][CH];
    auto issue = [&](int q, int slot) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int j = CH * q + i;
            X[slot][i] = ld_c64(px, j * 16);
            U[slot][i].re = ld_f64(pr, tr_entry(0, 2, j) * 16);
            U[slot][i].im = ld_f64(pi, tr_entry(0, 2, j) * 16);
            wl[slot][i] = ld_c64(t3, j * 16);
            wh[slot][i] = ld_c64(t3, (16 + j) * 16);
        }
    };
    issue(0, 0);
#pragma unroll
    for (int q = 0; q < 16 / CH; q++) {
        if (q < 16 / CH - 1) issue(q + 1, (q + 1) & 1);
        TFHE_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int j = CH * q + i;
            const c64 x = X[q & 1][i], u = U[q & 1][i];
            c64 lo, hi;
            lo.re = x.re + u.re; lo.im = fma(u.im, sg, x.im);
            hi.re = x.re - u.re; hi.im = fma(u.im, -sg, x.im);
            s.R[j] = cmul(lo, wl[q & 1][i]);
            s.R[16 + j] = cmul(hi, wh[q & 1][i]);
        }
    }
    fwd_block16<0>(s, k);
    fwd_block16<16>(s, k);
}

// out_c = ggsw[0][c] * F_0 (+)= ggsw[1][c] * F_1 (ggsw.rs:616-697); after the half swap a lane holds
// (F_0, F_1) of its 16 slots in (R[j], R[16 + j]); result (out_0, out_1) in the same registers.
// Slots 0..GRING-1 were prefetched by the caller; slot j+GRING is requested as soon as slot j's ring entry frees.
TFHE_DEV void phase_pointwise(Lane &s, const c64 *ggsw, int lane)
{
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const c64 f0 = s.R[j], f1 = s.R[16 + j];
        const c64 g00 = s.G[(j % GRING) * 4 + 0], g01 = s.G[(j % GRING) * 4 + 1];
        const c64 g10 = s.G[(j % GRING) * 4 + 2], g11 = s.G[(j % GRING) * 4 + 3];
        s.R[j] = cmul_add(g10, f1, cmul(g00, f0));
        s.R[16 + j] = cmul_add(g11, f1, cmul(g01, f0));
        if (j + GRING < 16) {
#pragma unroll
            for (int rc = 0; rc < 4; rc++) s.G[(j % GRING) * 4 + rc] = ggsw[((j + GRING) * 4 + rc) * 64 + lane];
            TFHE_SCHED_FENCE();
        }
    }
}

// inverse stages 5', 4', stage-3' twiddles and first layer
TFHE_DEV void inv_pass_b(Lane &s, const char *lds, int lane, const WaveConsts &k)
{
    const int par = lane & 1;
    const char *t3 = lds + opaque(TAB_T3 + par * 32 * 16);
    c64 wl[2][4], wh[2][4];
    auto issue = [&](int q, int slot) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            wl[slot][i] = ld_c64(t3, (4 * q + i) * 16);
            wh[slot][i] = ld_c64(t3, (16 + 4 * q + i) * 16);
        }
    };
    issue(0, 0);
    TFHE_SCHED_FENCE();
    inv_block16<0>(s, k);
    inv_block16<16>(s, k);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (q < 3) issue(q + 1, (q + 1) & 1);
        TFHE_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int j = 4 * q + i;
            const c64 lo = cmul_conj(s.R[j], wl[q & 1][i]);
            const c64 hi = cmul_conj(s.R[16 + j], wh[q & 1][i]);
            s.R[j].re = lo.re + hi.re; s.R[j].im = lo.im + hi.im;           // t0 (even lanes) / t2 (odd)
            s.R[16 + j].re = lo.re - hi.re; s.R[16 + j].im = lo.im - hi.im; // t1 (even lanes) / t3 (odd)
        }
    }
}

TFHE_DEV void inv_transpose_write(const Lane &s, char *lds, int poly_off, int lane)
{
    const int lp = lane & 31, par = lp & 1;
    char *p = lds + opaque(poly_off + tr_entry(lp >> 1, 2 * par, 0) * 16);
#pragma unroll
    for (int j = 0; j < 16; j++) {
        st_c64(p, j * 16, s.R[j]);
        st_c64(p, tr_entry(0, 1, j) * 16, s.R[16 + j]);
    }
}

// read pass-A' inputs; second layer of stage 3', stages 2', 1'; backward twist, rounding, accumulate
TFHE_DEV void inv_pass_a(Lane &s, const char *lds, int poly_off, int lane)
{
    const int l = lane & 31, up = l >> 4;
    const int swp = up * 8; // lanes >= 16 read t3 with re/im swapped
    const int base = poly_off + tr_entry(0, up, l & 15) * 16;
    const char *px = lds + opaque(base);
    const char *pr = lds + opaque(base + swp), *pi = lds + opaque(base + 8 - swp);
    const char *tab = lds + opaque(l * 16);
    const double sg = up ? -1.0 : 1.0;
    {
        c64 X[2][4], U[2][4];
        auto issue = [&](int q, int slot) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int m = 4 * q + i;
                X[slot][i] = ld_c64(px, tr_entry(m, 0, 0) * 16);
                U[slot][i].re = ld_f64(pr, tr_entry(m, 2, 0) * 16);
                U[slot][i].im = ld_f64(pi, tr_entry(m, 2, 0) * 16);
            }
        };
        issue(0, 0);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (q < 3) issue(q + 1, (q + 1) & 1);
            TFHE_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int m = 4 * q + i;
                const c64 x = X[q & 1][i], u = U[q & 1][i];
                s.R[2 * m].re = fma(u.re, sg, x.re);      s.R[2 * m].im = x.im + u.im;
                s.R[2 * m + 1].re = fma(u.re, -sg, x.re); s.R[2 * m + 1].im = x.im - u.im;
            }
        }
    }
    c64 w[2][6];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int t = 0; t < 3; t++) w[1][c * 3 + t] = ld_c64(lds + s.t2_off[c], t * 512);
    auto issue1 = [&](int k, int slot) {
#pragma unroll
        for (int e = 0; e < 6; e++) w[slot][e] = ld_c64(tab, TAB_T1 + ((2 * k) * 3 + e) * 32 * 16);
    };
    issue1(0, 0);
    TFHE_SCHED_FENCE();
#pragma unroll
    for (int blk = 0; blk < 4; blk++) // stage 2' (twiddles in w[1])
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int b = 8 * blk + c;
            s.R[b + 2] = cmul_conj(s.R[b + 2], w[1][c * 3 + 0]);
            s.R[b + 4] = cmul_conj(s.R[b + 4], w[1][c * 3 + 1]);
            s.R[b + 6] = cmul_conj(s.R[b + 6], w[1][c * 3 + 2]);
            bfly4_inv(s.R[b], s.R[b + 2], s.R[b + 4], s.R[b + 6]);
        }
    c64 tw[2][4];
    auto issue_tw = [&](int q, int slot) {
#pragma unroll
        for (int i = 0; i < 4; i++) tw[slot][i] = ld_c64(tab, TAB_TW + (4 * q + i) * 32 * 16);
    };
#pragma unroll
    for (int k = 0; k < 4; k++) { // stage 1'
        if (k < 3) issue1(k + 1, (k + 1) & 1);
        else issue_tw(0, 0);
        TFHE_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r0 = 2 * k + i;
            s.R[r0 + 8] = cmul_conj(s.R[r0 + 8], w[k & 1][i * 3 + 0]);
            s.R[r0 + 16] = cmul_conj(s.R[r0 + 16], w[k & 1][i * 3 + 1]);
            s.R[r0 + 24] = cmul_conj(s.R[r0 + 24], w[k & 1][i * 3 + 2]);
            bfly4_inv(s.R[r0], s.R[r0 + 8], s.R[r0 + 16], s.R[r0 + 24]);
        }
    }
    // convert_add_backward_torus (x86.rs:823-874): x * conj(w / n), from_torus, wrapping add.
    // The 1/n = 2^-10 scale is folded into the conversion (exact power of two), so the result is
    // bit-identical to scaling w first.  Registers hold ~acc, hence the subtraction.
#pragma unroll
    for (int q = 0; q < 8; q++) {
        if (q < 7) issue_tw(q + 1, (q + 1) & 1);
        TFHE_SCHED_FENCE();
        c64 v[4];
        uint32_t sh[8], slow = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            v[i] = cmul_conj(s.R[4 * q + i], tw[q & 1][i]);
            sh[2 * i] = torus_shift<54>(v[i].re);
            sh[2 * i + 1] = torus_shift<54>(v[i].im);
            slow |= (sh[2 * i] > 63u) | (sh[2 * i + 1] > 63u);
        }
        if (__builtin_expect(slow != 0, 0)) { // |value| < 1/4 (e.g. an all-zero digit polynomial) or huge
#pragma unroll
            for (int i = 0; i < 4; i++) {
                s.acc[2 * (4 * q + i)] -= torus_slow(v[i].re);
                s.acc[2 * (4 * q + i) + 1] -= torus_slow(v[i].im);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                nacc_sub_fast(s.acc[2 * (4 * q + i)], v[i].re, sh[2 * i]);
                nacc_sub_fast(s.acc[2 * (4 * q + i) + 1], v[i].im, sh[2 * i + 1]);
            }
        }
    }
}

// extract_lwe_sample_from_glwe_ciphertext at X^0 (glwe_sample_extraction.rs:125-146)
TFHE_DEV void phase_sample_extract(const Lane &s, uint64_t *out, int lane)
{
    const int h = lane >> 5, l = lane & 31;
#pragma unroll
    for (int r = 0; r < 32; r++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int c = l + 32 * r + FN * hh;
            const uint64_t v = ~s.acc[2 * r + hh];
            if (h == 0) out[c == 0 ? 0 : N - c] = (c == 0) ? v : (uint64_t)0 - v;
            else if (c == 0) out[N] = v;
        }
}

// one CMUX iteration's lane-local phases in order; X marks the points where lanes exchange data
//   mirror_write X rotate_decompose_twist, fwd_pass_a X(write) fwd_transpose_write X fwd_pass_b
//   SWAP pointwise SWAP inv_pass_b X(write) inv_transpose_write X inv_pass_a

} // namespace wave
} // namespace tfhe_hip
