// quad_pbs.hpp -- lane-level source of the QUAD blind rotation (variant 5): four wavefronts per ciphertext.
//
// A 256-thread workgroup per ciphertext, two workgroups per CU.  Wavefronts 0-1 own GLWE polynomial 0 (the mask),
// 2-3 polynomial 1 (the body); a thread holds EIGHT folded points of its polynomial, so that a transform crosses LDS
// once (variants 3 / 4: four times) and the two-lane butterflies are done with v_permlane32_swap / v_permlane16_swap:
//
//   pass A   thread (w, l): positions q = 512 i1 + 256 i0 + 128 i2 + 64 l[5] + 32 l[4] + 16 w + l[3:0], register
//            i = i0 + 2 i1 + 4 i2.  Stage 1 (position bits 9, 8) in registers; swap32 trades the finished bit 8 for
//            lane bit 5 (= bit 6): stage 2 (bits 7, 6); swap16 trades bit 9 for lane bit 4 (= bit 5): first layer of
//            stage 3 (bit 5).
//   -- one transposition through LDS (XOR-swizzled, conflict-free both ways), workgroup barrier --
//   pass B   registers j = b4 + 2 b3 + 4 b2, lane bits (5, 4) = (b1, b0), the other five lane / wave bits say WHICH
//            block of 32: second layer of stage 3 (bit 4; the +-i branch is chosen by the first-layer output d, which the
//            transposition makes the wavefront index), stage 4 (bits 3, 2), two swaps, stage 5 (bits 1, 0).
//   A radix-4 output m of a stage split like that stays in the thread that computed it; the Fourier order of this
//   kernel is therefore its own (fourier_pos below), and the bootstrap key is stored in that order.
//   The two groups exchange their Fourier polynomials through LDS; each forms its column of the external product and
//   transforms it back (the mirror image of the above).
//
// LDS per ciphertext: two 16 KiB images per polynomial -- X: accumulator mirror (read by the rotation), then the
// Fourier exchange image; Y: the transposition image of both transforms -- and 4 KiB of twiddle tables (stages 2, 3, 4;
// the registers hold accumulator, working set, twist, stage-1 twiddles and the step's key values): 68 KiB, four
// workgroup barriers per CMUX step.  Stored bytes per thread and step: 32 x 16 B (variant 4: 56 x 16 B).
// Measured (profiles/r03_quad_kernel.md): 4.47 ms for 257-512 bootstraps (variant 4: 4.82), 2.92 ms for <= 256
// (variant 3: 2.60): the automatic dispatch uses it for the two-per-CU band only.
//
// Arithmetic: exactly DESIGN.md section 4 (FFT schedule v1) with the integer identities of pair_pbs.hpp; bit-exact
// against the oracle.  The file is a sequence of lane-local phases (tests/emu runs the same source on the CPU).
#pragma once
#include "wave_pbs.hpp"
#include "carry_chain.hpp"

namespace tfhe_hip {
namespace quad {

using wave::FN;
using wave::LOG2N;
using wave::N;
using wave::ld_c64;
using wave::ld_u64;
using wave::opaque;
using wave::st_c64;
using wave::st_u64;

using cc::add_fraction4;
using cc::add_wraps;
using cc::alignbit;
using cc::and3;
using cc::f64_hi;
using cc::hi_of_complement_sums4;
using cc::lane_mask;
using cc::mask_of;
using cc::neg_digit_from_hi;

constexpr double BSK_PRESCALE = -0.0009765625; // see pair_pbs.hpp
#if defined(TFHE_HIP_HOST_EMU)
inline uint64_t torus_slow_scaled(double w) { return torus_from_f64(w); }
#else
__device__ __attribute__((noinline)) uint64_t torus_slow_scaled(double w) { return torus_from_f64(w); }
#endif

// keeps the two bodies of a wave-uniform branch apart (the compiler would otherwise turn them into one body with
// v_cndmask selects or register moves on one side)
#if defined(TFHE_HIP_HOST_EMU)
#define TFHE_BRANCH_MARK(text) ((void)0)
#else
#define TFHE_BRANCH_MARK(text) asm volatile("; " text ::: "memory")
#endif

constexpr int THREADS = 256;
constexpr int REGION = 16384;       // one image of one polynomial; 16 KiB-aligned (rotation addresses, pair_pbs.hpp)
constexpr int BUF_X = 0;            // [p 2]: accumulator mirror u64[2048] (linear) / exchange image [reg 8][T 128] c64
constexpr int BUF_Y = 2 * REGION;   // [p 2]: transposition image, slots of 16 B, swizzled (tr_slot)
constexpr int TAB_T3 = 4 * REGION;  // stage-3 twiddles of pass B [reg 8][class 8] c64 (the 8 classes of a read: 32 distinct banks)
constexpr int TAB_W2 = TAB_T3 + 64 * 16;     // stage-2 twiddles [m 3][c0 mod 64] c64
constexpr int TAB_W4 = TAB_W2 + 3 * 64 * 16; // stage-4 twiddles [2 l[5] + l[4]][m 3] c64
constexpr int LDS_BYTES = TAB_W4 + 4 * 3 * 16; // 69 824
#ifndef TFHE_QGHEAD
#define TFHE_QGHEAD 4
#endif
constexpr int QGHEAD = TFHE_QGHEAD; // bootstrap-key positions fetched at the top of a step (the rest in front of the exchange)

struct Ids {
    int p, w, l, T; // polynomial / column, wavefront in the group, lane, thread in the group
};
TFHE_DEV Ids make_ids(int tid)
{
    Ids d;
    d.p = tid >> 7;
    d.w = (tid >> 6) & 1;
    d.l = tid & 63;
    d.T = tid & 127;
    return d;
}

// ---- index maps ---------------------------------------------------------------------------------
// pass A: folded position of register i of thread (w, l) = reg_q(i) + lane_c0(w, l)
TFHE_DEV constexpr int lane_c0(int w, int l) { return 64 * (l >> 5) + 32 * ((l >> 4) & 1) + 16 * w + (l & 15); }
TFHE_DEV constexpr int reg_q(int i) { return 256 * (i & 1) + 512 * ((i >> 1) & 1) + 128 * (i >> 2); }
// DIF position (oracle order: 256 m1 + 64 m2 + 16 m3 + 4 m4 + m5, m_s = output index of stage s) of the Fourier value
// in register `reg` of thread (w, l) after pass B
TFHE_DEV constexpr int fourier_pos(int w, int l, int reg)
{
    const int m1 = 2 * (l & 1) + ((l >> 3) & 1), m2 = 2 * ((l >> 2) & 1) + ((l >> 1) & 1), m3 = 2 * (l >> 5) + w;
    const int m4 = 2 * ((l >> 4) & 1) + (reg >> 2), m5 = 2 * (reg & 1) + ((reg >> 1) & 1);
    return 256 * m1 + 64 * m2 + 16 * m3 + 4 * m4 + m5;
}
// Device BSK layout: [i][column p][wavefront w][reg 8][row 2][lane 64] c64
TFHE_DEV constexpr size_t bsk_quad_index(size_t i, int p, int w, int reg, int row, int lane)
{
    return (((((i * 2 + p) * 2 + w) * 8 + reg) * 2 + row) * 64) + lane;
}
// Transposition image.  Pass-B thread (w', l'), register j owns slot j * 128 + 64 w' + l' with w' = d (the first-layer
// output of stage 3: the second layer's branch is then wave-uniform), l' = 32 b1 + 16 b0 + 8 m1[0] + 4 m2[1] + 2 m2[0] + m1[1];
// bits (j[1], j[2], l'[5], l'[4]) = position bits (3, 2, 1, 0) are XORed into the low four bits, which makes the 8 / 16-lane
// groups of both directions hit distinct 16-B columns (MI355X_MICROARCH.md, LDS: ds_write_b128 8 x 8 lanes, ds_read_b128
// 4 x 16 lanes; tests/emu checks every pattern).
// Pass-A side: register r = 4 b7 + b6 + 2 d of thread (w, l) [lane bits (5,4) = position bits (8,9) at that point]
TFHE_DEV int tr_base_a(int w, int l)
{
    const int lo = l & 15, b3 = lo >> 3, b2 = (lo >> 2) & 1, b1 = (lo >> 1) & 1, b0 = lo & 1;
    const int slot = (w + 2 * b3 + 4 * b2) * 128 + 32 * b1 + 16 * b0 + 8 * ((l >> 5) ^ b3) + 4 * b2 + 2 * b1 + (((l >> 4) & 1) ^ b0);
    return slot << 4;
}
// register r of pass A: d -> w' (slot bit 6), (b7, b6) -> slot bits (2, 1)
TFHE_DEV constexpr int tr_xor_a(int r) { return (64 * ((r >> 1) & 1) + 4 * (r >> 2) + 2 * (r & 1)) << 4; }
// pass-B side: four bases (j[2:1]) + j * 2048
TFHE_DEV int tr_base_b(int w, int l) { return (64 * w + (l ^ (2 * (l >> 5) + ((l >> 4) & 1)))) << 4; }
TFHE_DEV constexpr int tr_xor_b(int j) { return (4 * (j >> 2) + 8 * ((j >> 1) & 1)) << 4; }

struct Lane {
    uint64_t acc[16]; // acc[2i + hh] = coefficient (reg_q(i) + lane_c0) + 1024 hh of polynomial p, MINUS ONE
    c64 R[8];         // FFT working set
    c64 tw[8];        // twist of the pass-A positions
    c64 w1[6];        // stage 1: [i2][m - 1] = W[(128 i2 + c0) m]
    c64 G[16];        // bootstrap-key values of the step: rows 0 / 1 of column p at the thread's 8 Fourier positions
                      // (positions 0-3 fetched at the top of the step, 4-7 in front of the exchange: register pressure)
};

// ---- phases -------------------------------------------------------------------------------------

TFHE_DEV void phase_init(Lane &s, char *lds, const uint64_t *lut, const c64 *twist, const c64 *W, uint32_t b_hat,
                         const Ids &d, int tid)
{
    const int rem = b_hat & (N - 1), cyc = b_hat >> LOG2N;
    const int c0 = lane_c0(d.w, d.l);
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int c = reg_q(i) + c0 + FN * hh;
            int src = c + rem, ng = cyc & 1;
            if (src >= N) { src -= N; ng ^= 1; }
            const uint64_t v = lut[d.p * N + src];
            s.acc[2 * i + hh] = (ng ? (uint64_t)0 - v : v) - 1; // registers hold acc - 1
        }
        s.tw[i] = twist[reg_q(i) + c0];
    }
#pragma unroll
    for (int m = 1; m < 4; m++) {
        s.w1[m - 1] = W[c0 * m];
        s.w1[3 + m - 1] = W[(128 + c0) * m];
    }
    if (tid < 192) st_c64(lds, TAB_W2 + tid * 16, W[4 * (tid & 63) * ((tid >> 6) + 1)]);
    if (tid < 12) st_c64(lds, TAB_W4 + tid * 16, W[64 * (tid / 3) * (tid % 3 + 1)]);
    if (tid < 64) { // class c = 4 d + 2 b1 + b0 (d = pass-B wavefront), register j = b4 + 2 b3 + 4 b2: W[16 (2 b4 + d) (8 b3 + 4 b2 + 2 b1 + b0)]
        const int j = tid >> 3, c = tid & 7;
        const int m = 2 * (j & 1) + (c >> 2), jj = 8 * ((j >> 1) & 1) + 4 * (j >> 2) + (c & 3);
        st_c64(lds, TAB_T3 + tid * 16, W[16 * m * jj]);
    }
}

template <int J0, int CNT> TFHE_DEV void ggsw_prefetch(Lane &s, const c64 *col, int lane)
{
#pragma unroll
    for (int j = J0; j < J0 + CNT; j++) {
        s.G[j * 2 + 0] = col[(j * 2 + 0) * 64 + lane];
        s.G[j * 2 + 1] = col[(j * 2 + 1) * 64 + lane];
    }
}

// The mirror holds coefficient c at index c with bits 4 and 5 exchanged: the 32 lanes of a ds_read_b64 group hold
// coefficients (l[4], l[3:0]) = position bits (5, 3..0) -- two runs of 16 that are 32 apart and would share their banks in
// a linear image; with the exchange they are 16 apart (32 banks), whatever the rotation.
TFHE_DEV constexpr uint32_t swap45(uint32_t x) { return (x & ~48u) | ((x & 16u) << 1) | ((x & 32u) >> 1); }
TFHE_DEV void phase_mirror_write(const Lane &s, char *lds, const Ids &d)
{
    char *p = lds + opaque(BUF_X + d.p * REGION + (int)swap45((uint32_t)lane_c0(d.w, d.l)) * 8);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) st_u64(p, (reg_q(i) + FN * hh) * 8, s.acc[2 * i + hh]);
}

// rotation by X^{a_hat}, subtraction, one-digit decomposition (negated digits), fold + forward twist: pair_pbs.hpp's
// scheme (block number in the top bits of `A`, the carry-out of the block addition is the negacyclic wrap)
template <int BASE_LOG> TFHE_DEV void phase_rotate_decompose_twist(Lane &s, const char *lds, const Ids &d, uint32_t a_hat)
{
    uint64_t rot[2][8];
    lane_mask pol[2][8];
    const uint32_t base = ((uint32_t)lane_c0(d.w, d.l) - a_hat) & (2 * N - 1);
    const uint32_t A = ((base >> 6) << 27) | (swap45(base & 63) << 21);
    const lane_mask p0 = mask_of(((base >> 11) & 1u) != 0);
    const uint32_t region = (uint32_t)(BUF_X + d.p * REGION) >> 14;
    auto issue = [&](int k, int slot) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 4 * k + i;
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                uint32_t sum;
                const lane_mask wrap = add_wraps(A, (uint32_t)((reg_q(r) >> 6) + 16 * hh) << 27, sum);
                rot[slot][2 * i + hh] = ld_u64(lds, (int)alignbit(region, sum, 18));
                pol[slot][2 * i + hh] = wrap ^ p0;
            }
        }
    };
    issue(0, 0);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (k < 1) issue(k + 1, (k + 1) & 1);
        TFHE_SCHED_FENCE();
        uint32_t dhi[8];
#pragma unroll
        for (int g = 0; g < 2; g++) {
            uint64_t a[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int e = 4 * g + j;
                a[j] = s.acc[2 * (4 * k + (e >> 1)) + (e & 1)];
            }
            hi_of_complement_sums4(&rot[k & 1][4 * g], a, &pol[k & 1][4 * g], dhi + 4 * g);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            c64 dg;
            dg.re = neg_digit_from_hi<BASE_LOG>(dhi[2 * i]);
            dg.im = neg_digit_from_hi<BASE_LOG>(dhi[2 * i + 1]);
            s.R[4 * k + i] = cmul(dg, s.tw[4 * k + i]);
        }
    }
}

// stage 1 (L = 1024): positions j + 256 m, m = i0 + 2 i1
TFHE_DEV void fwd_stage1(Lane &s)
{
#pragma unroll
    for (int i2 = 0; i2 < 2; i2++) {
        const int b = 4 * i2;
        bfly4_fwd(s.R[b], s.R[b + 1], s.R[b + 2], s.R[b + 3]);
#pragma unroll
        for (int m = 1; m < 4; m++) s.R[b + m] = cmul(s.R[b + m], s.w1[3 * i2 + m - 1]);
    }
}
//   [swap32 of (R[2k], R[2k+1])]: lane bit 5 = position bit 8, register bit 0 = position bit 6
// stage 2 (L = 256): positions j + 64 m, m = b6 + 2 b7 = register bits (0, 2)
// the thread's stage-2 twiddles W[4 (c0 mod 64) m] (at that point lane bit 4 is still position bit 5)
struct Tw3 {
    c64 w[3];
};
TFHE_DEV Tw3 load_w2(const char *lds, const Ids &d)
{
    const char *t = lds + opaque(TAB_W2 + (lane_c0(d.w, d.l) & 63) * 16);
    Tw3 r;
#pragma unroll
    for (int m = 0; m < 3; m++) r.w[m] = ld_c64(t, m * 1024);
    return r;
}
// ... and stage-4 twiddles W[64 (2 l[5] + l[4]) m]
TFHE_DEV Tw3 load_w4(const char *lds, const Ids &d)
{
    const char *t = lds + opaque(TAB_W4 + (2 * (d.l >> 5) + ((d.l >> 4) & 1)) * 48);
    Tw3 r;
#pragma unroll
    for (int m = 0; m < 3; m++) r.w[m] = ld_c64(t, m * 16);
    return r;
}
TFHE_DEV void fwd_stage2(Lane &s, const Tw3 &w2)
{
#pragma unroll
    for (int i1 = 0; i1 < 2; i1++) {
        const int b = 2 * i1;
        bfly4_fwd(s.R[b], s.R[b + 1], s.R[b + 4], s.R[b + 5]);
        s.R[b + 1] = cmul(s.R[b + 1], w2.w[0]);
        s.R[b + 4] = cmul(s.R[b + 4], w2.w[1]);
        s.R[b + 5] = cmul(s.R[b + 5], w2.w[2]);
    }
}
//   [swap16 of (R[x], R[x+2]), x = 0, 1, 4, 5]: lane bit 4 = position bit 9, register bit 1 = position bit 5
// stage 3 (L = 64), first layer: (t0, t1) in wavefront 0 (position bit 4 = 0), (t2, t3) in wavefront 1
TFHE_DEV void fwd_stage3_layer_a(Lane &s)
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = (k & 1) + 4 * (k >> 1);
        const c64 lo = s.R[x], hi = s.R[x + 2];
        s.R[x].re = lo.re + hi.re; s.R[x].im = lo.im + hi.im;
        s.R[x + 2].re = lo.re - hi.re; s.R[x + 2].im = lo.im - hi.im;
    }
}
TFHE_DEV void transpose_write_a(const Lane &s, char *lds, const Ids &d)
{
    const int a = opaque(BUF_Y + d.p * REGION + tr_base_a(d.w, d.l));
#pragma unroll
    for (int r = 0; r < 8; r++) st_c64(lds + (a ^ tr_xor_a(r)), 0, s.R[r]);
}
TFHE_DEV void transpose_read_a(Lane &s, const char *lds, const Ids &d)
{
    const int a = opaque(BUF_Y + d.p * REGION + tr_base_a(d.w, d.l));
#pragma unroll
    for (int r = 0; r < 8; r++) s.R[r] = ld_c64(lds + (a ^ tr_xor_a(r)), 0);
}
TFHE_DEV void transpose_read_b(Lane &s, const char *lds, const Ids &d)
{
    int a[4];
    a[0] = opaque(BUF_Y + d.p * REGION + tr_base_b(d.w, d.l));
#pragma unroll
    for (int c = 1; c < 4; c++) a[c] = opaque(a[0] ^ tr_xor_b(2 * c));
#pragma unroll
    for (int j = 0; j < 8; j++) s.R[j] = ld_c64(lds + a[j >> 1], j * 2048);
}
TFHE_DEV void transpose_write_b(const Lane &s, char *lds, const Ids &d)
{
    int a[4];
    a[0] = opaque(BUF_Y + d.p * REGION + tr_base_b(d.w, d.l));
#pragma unroll
    for (int c = 1; c < 4; c++) a[c] = opaque(a[0] ^ tr_xor_b(2 * c));
#pragma unroll
    for (int j = 0; j < 8; j++) st_c64(lds + a[j >> 1], j * 2048, s.R[j]);
}

// the thread's pass-B twiddles: stage 3 (class 4 d + 2 b1 + b0, per register) and stage 4.  The kernel issues these loads
// ahead of the workgroup barrier / the swaps in front of their use.
struct TwB {
    c64 t3[8];
    Tw3 w4;
};
TFHE_DEV TwB load_twb(const char *lds, const Ids &d)
{
    const int cls = 4 * d.w + 2 * (d.l >> 5) + ((d.l >> 4) & 1);
    const char *t3 = lds + opaque(TAB_T3 + cls * 16);
    TwB r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.t3[j] = ld_c64(t3, j * 128);
    r.w4 = load_w4(lds, d);
    return r;
}
// pass B, first part: second layer of stage 3 (+ its twiddles), stage 4
TFHE_DEV void fwd_pass_b1(Lane &s, const TwB &tb, const Ids &d)
{
    const c64 *T3 = tb.t3;
    const Tw3 &w4 = tb.w4;
    if (d.w == 0) { // (t0, t2) -> y0, y2
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const c64 x = s.R[2 * k], y = s.R[2 * k + 1];
            c64 lo, hi;
            lo.re = x.re + y.re; lo.im = x.im + y.im;
            hi.re = x.re - y.re; hi.im = x.im - y.im;
            s.R[2 * k] = cmul(lo, T3[2 * k]);         // y0 (W^0)
            s.R[2 * k + 1] = cmul(hi, T3[2 * k + 1]); // y2 W[32 jj]
        }
        TFHE_BRANCH_MARK("stage 3, second layer: even outputs");
    } else { // (t1, t3) -> y1 = t1 - i t3, y3 = t1 + i t3
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const c64 x = s.R[2 * k], y = s.R[2 * k + 1];
            c64 lo, hi;
            lo.re = x.re + y.im; lo.im = x.im - y.re;
            hi.re = x.re - y.im; hi.im = x.im + y.re;
            s.R[2 * k] = cmul(lo, T3[2 * k]);         // y1 W[16 jj]
            s.R[2 * k + 1] = cmul(hi, T3[2 * k + 1]); // y3 W[48 jj]
        }
        TFHE_BRANCH_MARK("stage 3, second layer: odd outputs");
    }
#pragma unroll
    for (int j0 = 0; j0 < 2; j0++) { // stage 4 (L = 16): positions j + 4 m, m = b2 + 2 b3 = register bits (2, 1)
        bfly4_fwd(s.R[j0], s.R[j0 + 4], s.R[j0 + 2], s.R[j0 + 6]);
        s.R[j0 + 4] = cmul(s.R[j0 + 4], w4.w[0]);
        s.R[j0 + 2] = cmul(s.R[j0 + 2], w4.w[1]);
        s.R[j0 + 6] = cmul(s.R[j0 + 6], w4.w[2]);
    }
}
//   [swap32 of (R[2k], R[2k+1])] [swap16 of (R[x], R[x+2])]: register bits (0, 1) = position bits (1, 0)
TFHE_DEV void fwd_stage5(Lane &s)
{
#pragma unroll
    for (int j2 = 0; j2 < 2; j2++) bfly4_fwd(s.R[4 * j2], s.R[4 * j2 + 2], s.R[4 * j2 + 1], s.R[4 * j2 + 3]);
}

// publish this group's Fourier polynomial for the other group: [reg][T]
TFHE_DEV void exchange_write(const Lane &s, char *lds, const Ids &d)
{
    char *p = lds + opaque(BUF_X + d.p * REGION + d.T * 16);
#pragma unroll
    for (int j = 0; j < 8; j++) st_c64(p, j * 2048, s.R[j]);
}

// out_p = ggsw[0][p] * F_0 (+)= ggsw[1][p] * F_1 (ggsw.rs:477-598); own polynomial in R, the other group's from LDS
template <int P> TFHE_DEV void phase_pointwise(Lane &s, const char *lds, const Ids &d)
{
    const char *p = lds + opaque(BUF_X + (1 - P) * REGION + d.T * 16);
    c64 F[8];
#pragma unroll
    for (int j = 0; j < 8; j++) F[j] = ld_c64(p, j * 2048);
    TFHE_SCHED_FENCE();
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const c64 f0 = P ? F[j] : s.R[j], f1 = P ? s.R[j] : F[j];
        const c64 g0 = s.G[j * 2 + 0], g1 = s.G[j * 2 + 1];
        s.R[j] = cmul_add(g1, f1, cmul(g0, f0));
    }
}

TFHE_DEV void inv_stage5(Lane &s)
{
#pragma unroll
    for (int j2 = 0; j2 < 2; j2++) bfly4_inv(s.R[4 * j2], s.R[4 * j2 + 2], s.R[4 * j2 + 1], s.R[4 * j2 + 3]);
}
//   [swap16] [swap32]
// stage 4', twiddles and first layer of stage 3'
TFHE_DEV void inv_pass_b1(Lane &s, const TwB &tb)
{
    const c64 *T3 = tb.t3;
    const Tw3 &w4 = tb.w4;
#pragma unroll
    for (int j0 = 0; j0 < 2; j0++) {
        s.R[j0 + 4] = cmul_conj(s.R[j0 + 4], w4.w[0]);
        s.R[j0 + 2] = cmul_conj(s.R[j0 + 2], w4.w[1]);
        s.R[j0 + 6] = cmul_conj(s.R[j0 + 6], w4.w[2]);
        bfly4_inv(s.R[j0], s.R[j0 + 4], s.R[j0 + 2], s.R[j0 + 6]);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { // (y0, y2) -> (t0, t1) in even lanes, (y1, y3) -> (t2, t3) in odd lanes
        const c64 lo = cmul_conj(s.R[2 * k], T3[2 * k]), hi = cmul_conj(s.R[2 * k + 1], T3[2 * k + 1]);
        s.R[2 * k].re = lo.re + hi.re; s.R[2 * k].im = lo.im + hi.im;
        s.R[2 * k + 1].re = lo.re - hi.re; s.R[2 * k + 1].im = lo.im - hi.im;
    }
}
// pass A': second layer of stage 3': wavefront 0 holds (t0, t2) -> (x0, x2), wavefront 1 (t1, t3) -> (x1, x3)
TFHE_DEV void inv_stage3_layer_b(Lane &s, const Ids &d)
{
    if (d.w == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int xi = (k & 1) + 4 * (k >> 1);
            const c64 x = s.R[xi], y = s.R[xi + 2];
            s.R[xi].re = x.re + y.re;     s.R[xi].im = x.im + y.im;     // x0
            s.R[xi + 2].re = x.re - y.re; s.R[xi + 2].im = x.im - y.im; // x2
        }
        TFHE_BRANCH_MARK("stage 3', second layer: even inputs");
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int xi = (k & 1) + 4 * (k >> 1);
            const c64 x = s.R[xi], y = s.R[xi + 2];
            s.R[xi].re = x.re - y.im;     s.R[xi].im = x.im + y.re;     // x1 = t1 + i t3
            s.R[xi + 2].re = x.re + y.im; s.R[xi + 2].im = x.im - y.re; // x3 = t1 - i t3
        }
        TFHE_BRANCH_MARK("stage 3', second layer: odd inputs");
    }
}
//   [swap16]
TFHE_DEV void inv_stage2(Lane &s, const Tw3 &w2)
{
#pragma unroll
    for (int i1 = 0; i1 < 2; i1++) {
        const int b = 2 * i1;
        s.R[b + 1] = cmul_conj(s.R[b + 1], w2.w[0]);
        s.R[b + 4] = cmul_conj(s.R[b + 4], w2.w[1]);
        s.R[b + 5] = cmul_conj(s.R[b + 5], w2.w[2]);
        bfly4_inv(s.R[b], s.R[b + 1], s.R[b + 4], s.R[b + 5]);
    }
}
//   [swap32]
// stage 1', backward twist (x86.rs:842-861), rounding to the torus, accumulate (integer identities of pair_pbs.hpp)
TFHE_DEV void inv_stage1_accumulate(Lane &s)
{
#pragma unroll
    for (int i2 = 0; i2 < 2; i2++) {
        const int b = 4 * i2;
#pragma unroll
        for (int m = 1; m < 4; m++) s.R[b + m] = cmul_conj(s.R[b + m], s.w1[3 * i2 + m - 1]);
        bfly4_inv(s.R[b], s.R[b + 1], s.R[b + 2], s.R[b + 3]);
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        c64 v[4];
        uint32_t ok = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            v[i] = cmul_conj(s.R[4 * q + i], s.tw[4 * q + i]);
            ok = and3(ok, f64_hi(v[i].re), f64_hi(v[i].im));
        }
        if (__builtin_expect((ok & 0x40000000u) == 0, 0)) { // some |w| < 2: the reference formula, out of line
#pragma unroll
            for (int i = 0; i < 4; i++) {
                s.acc[2 * (4 * q + i)] += torus_slow_scaled(v[i].re);
                s.acc[2 * (4 * q + i) + 1] += torus_slow_scaled(v[i].im);
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
                const int b = 2 * (4 * q + 2 * g);
                add_fraction4(&s.acc[b], &s.acc[b + 1], &s.acc[b + 2], &s.acc[b + 3], f);
            }
        }
    }
}

// sample extraction at X^0 (glwe_sample_extraction.rs:125-146): group 0 writes the mask, group 1 the body
TFHE_DEV void phase_sample_extract(const Lane &s, uint64_t *out, const Ids &d)
{
    const int c0 = lane_c0(d.w, d.l);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int c = reg_q(i) + c0 + FN * hh;
            const uint64_t v = s.acc[2 * i + hh] + 1;
            if (d.p == 0) out[c == 0 ? 0 : N - c] = (c == 0) ? v : (uint64_t)0 - v;
            else if (c == 0) out[N] = v;
        }
}

} // namespace quad
} // namespace tfhe_hip
