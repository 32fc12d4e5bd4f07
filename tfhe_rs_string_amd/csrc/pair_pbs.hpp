// pair_pbs.hpp -- lane-level source of the wave-PAIR blind rotation (variant 2).
//
// Two wavefronts per ciphertext, one per GLWE polynomial (wave h owns polynomial h: its half of the
// accumulator, its digit polynomial, its output column of the external product).  64 lanes x 16
// folded points per lane, so a wave needs <= 256 registers and two waves fit on every SIMD; a
// ciphertext's CMUX step runs on two SIMDs at once (half the latency of variant 1).
//
//   pass A   lane l holds points q = l + 64 r, r < 16: stages 1 (L=1024) and 2 (L=256) in registers;
//            stage 3 (L=64) pairs lanes 32 and 16 apart: done with v_permlane32_swap / v_permlane16_swap,
//            which turn a butterfly between two lanes into a butterfly between two registers
//            (no redundant arithmetic, no LDS)
//   transposition through LDS (one per transform, XOR-swizzled, conflict-free)
//   pass B   lane l' holds the 16 consecutive positions 16 l' .. 16 l' + 15: stages 4 and 5
//   the two waves of a ciphertext exchange their Fourier polynomials through LDS (two workgroup
//   barriers per CMUX step); each then forms its own output column and transforms it back.
//
// The arithmetic DAG is again exactly DESIGN.md section 4 (bit-exact against the oracle); only the
// placement of values differs from wave_pbs.hpp (variant 1) and kernels.hip (variant 0).
// Like wave_pbs.hpp the file is a sequence of lane-local phases; the points where lanes / waves
// exchange data (permlane swaps, LDS direction changes, barriers) are between phases, so the host
// emulation (tests/emu) can run the same source lane by lane.
#pragma once
#include "wave_pbs.hpp"
#include "carry_chain.hpp"

namespace tfhe_hip {
namespace pair {

using wave::FN;
using wave::LOG2N;
using wave::N;
using wave::WaveConsts;
using wave::digit_from_hi;
using wave::fwd_block16;
using wave::inv_block16;
using wave::ld_c64;
using wave::ld_u64;
using wave::nacc_sub_fast;
using wave::opaque;
using wave::st_c64;
using wave::st_u64;
using wave::torus_shift;
using wave::torus_slow;

using cc::and3;
using cc::f64_hi;
using cc::f64_lo;
// cold path of the accumulation: the reference formula (from_torus, commons/math/torus/mod.rs:72-78 in the rounding of
// x86.rs:864-871) on w = the torus increment in revolutions (see BSK_PRESCALE)
#if defined(TFHE_HIP_HOST_EMU)
inline uint64_t torus_slow_scaled(double w) { return torus_from_f64(w); }
#else
__device__ __attribute__((noinline)) uint64_t torus_slow_scaled(double w) { return torus_from_f64(w); }
#endif

constexpr int WAVES_PER_WG = 8; // 4 ciphertexts x 2 polynomials
#ifndef TFHE_PGRING
#define TFHE_PGRING 4
#endif
constexpr int PGRING = TFHE_PGRING; // bootstrap-key prefetch ring depth (Fourier slots)

// ---- LDS layout (bytes) ---------------------------------------------------------------------
// Wave regions first, each 16 KiB and 16 KiB-ALIGNED (the rotation forms its addresses with one v_alignbit_b32:
// region index : 14-bit offset), the shared tables behind them (their offsets are relative to TAB0; lane bases
// include TAB0, so the per-access immediates stay below 64 KiB).
constexpr int WAVE0 = 0;
constexpr int WAVE_BYTES = 16384; // per wave: accumulator mirror / transposition buffer / exchange buffer
constexpr int TAB0 = WAVE0 + WAVES_PER_WG * WAVE_BYTES; // 131 072
constexpr int TAB_TW = 0;                       // twist   [r 16][l 64]        16 KiB
constexpr int TAB_T1 = TAB_TW + 16 * 64 * 16;   // stage 1 [c 4][t 3][l 64]    12 KiB
constexpr int TAB_T3 = TAB_T1 + 4 * 3 * 64 * 16; // stage 3 [slot 2][l 64]      2 KiB
constexpr int TAB_BYTES = TAB_T3 + 2 * 64 * 16;
constexpr int FLAGS = TAB0 + TAB_BYTES; // one progress counter (16 B slot) per wave
constexpr int LDS_BYTES = FLAGS + WAVES_PER_WG * 16;     // 161 920
static_assert(LDS_BYTES <= 163840, "LDS budget of one CU");
static_assert(WAVE0 % 16384 == 0 && WAVE_BYTES == 16384, "wave regions must be 16 KiB aligned (rotation addresses)");
constexpr int GTAB_BYTES = TAB_BYTES; // global image == LDS table image

// Integer conventions of this kernel (all exact identities: the results are the reference's bits):
//   * registers and mirror hold A1 = coefficient - 1.  With s the negacyclic sign of a rotated source coefficient v and
//     p = !s, the one's complement of the difference d = +-v - acc is ~d = (A1_v ^ -p) + A1_acc + !p: one conditional
//     complement and a carry chain (carry_chain.hpp), and the signed digit of d comes out NEGATED as a plain signed bit
//     field: -digit = sbfe(~d.hi + 2^(31 - B), 32 - B, B) -- the closest_representable rounding (decomposer.rs:97-118)
//     and the balanced digit (decomposition.rs:78-86) in two instructions.
//   * every digit being negated, the Fourier key is stored multiplied by BSK_PRESCALE = -2^-10 (an exact power-of-two
//     scaling: every product and sum of the external product scales exactly, nothing rounds differently): the backward
//     twist then yields w = the torus increment in REVOLUTIONS (the 1/n of convert_add_backward_torus included).
//   * acc += RNE(w 2^64) mod 2^64 (from_torus): for |w| >= 2 (ulp >= 2^-51) fract(w) + 1 is exact and its 52 fraction bits
//     ARE the increment in units of 2^-52 -- whatever the sign of w: v_fract_f64, v_add_f64, one v_alignbit_b32 and a
//     shift make the two words, a carry chain adds them.  |w| < 2 (probability ~2^-23 per coefficient, and the all-zero
//     digit polynomials of the edge cases) takes the reference formula out of line.
constexpr double BSK_PRESCALE = -0.0009765625;

// Slot of lane l' in a 64-entry row of the stage-1 table.  Stage 1 reads the row with l' = lane, stage 2 reads entries
// l' = 4 lane mod 64 of it (W[4 l (t+1)] is a stage-1 entry): with the low four bits XOR-ed by l' >> 4 both patterns hit 16
// distinct 16-B columns in every ds_read_b128 lane group (the plain row gave the stage-2 reads a four-way bank conflict:
// all of the kernel's SQ_LDS_BANK_CONFLICT, 72 cycles per step and wavefront).
#if defined(TFHE_HIP_HOST_EMU)
inline constexpr int t1_slot(int l) { return (l & 48) | ((l & 15) ^ (l >> 4)); }
#else
__host__ __device__ inline constexpr int t1_slot(int l) { return (l & 48) | ((l & 15) ^ (l >> 4)); } // (the host builds the table image)
#endif

// pass-B lane that holds the value of pass-A lane `l`, register pair `m`, slot `s` after stage 3
TFHE_DEV constexpr int dest_lane(int l, int m, int s) { return 8 * m + 2 * s + 4 * (l >> 5) + ((l >> 4) & 1); }
// Device BSK layout: [i][column h][slot j 16][row 2][lane 64] c64 (lane l' holds positions 16 l' + j)
TFHE_DEV constexpr size_t bsk_pair_index(size_t i, int h, int j, int row, int lane)
{
    return ((((i * 2 + h) * 16 + j) * 2 + row) * 64) + lane;
}

// Pairwise hand-off between the two waves of a ciphertext (instead of workgroup barriers, which
// would couple the four ciphertexts of a workgroup): each wave owns a monotonically increasing
// progress counter in LDS.  The counter is published with a workgroup-scope RELEASE store (everything
// the wave wrote to / read from LDS before it is complete) and polled with workgroup-scope ACQUIRE
// loads; both waves of a pair are in the same workgroup, hence co-resident, and run the same number
// of steps.  The spin is bounded, and running out of it is an ERROR, not a fall-through: the wave
// records ERR_PAIR_TIMEOUT in the key's device-visible error word (the host turns it into rc 1,
// api.cpp) and stops waiting for the rest of the launch, so that the grid still drains.
#if defined(TFHE_HIP_HOST_EMU)
TFHE_DEV void pair_signal(char *, int, uint32_t) {}
TFHE_DEV bool pair_wait(const char *, int, uint32_t, bool) { return true; }
#else
constexpr int PAIR_SPIN_LIMIT = 1 << 22; // x >= 64 clocks per poll: > 0.1 s, five orders of magnitude above a CMUX step
TFHE_DEV void pair_signal(char *lds, int flag_off, uint32_t value)
{
    __hip_atomic_store(reinterpret_cast<uint32_t *>(lds + flag_off), value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// returns false when the partner did not get there in time (or `alive` was already false)
TFHE_DEV bool pair_wait(const char *lds, int flag_off, uint32_t want, bool alive)
{
    if (!alive) return false;
    for (int spin = 0; spin < PAIR_SPIN_LIMIT; spin++) {
        const uint32_t v = __hip_atomic_load(reinterpret_cast<const uint32_t *>(lds + flag_off), __ATOMIC_ACQUIRE,
                                             __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((int32_t)(v - want) >= 0) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}
#endif

using cc::add_fraction4;
using cc::add_wraps;
using cc::alignbit;
using cc::hi_of_complement_sums4;
using cc::lane_mask;
using cc::mask_of;
using cc::neg_digit_from_hi;

struct Lane {
    uint64_t acc[32]; // acc[2r + hh] = coefficient (l + 64 r) + 1024 hh of polynomial h, MINUS ONE
    c64 R[16];        // FFT working set
    c64 G[2 * PGRING]; // bootstrap-key prefetch ring (rows 0 / 1 of this wave's GGSW column)
    int t2_off;       // LDS byte offset of this lane's stage-2 twiddles W[4 l (t+1)] inside the stage-1 table
};

// ---- phases -----------------------------------------------------------------------------------

TFHE_DEV void phase_init(Lane &s, const uint64_t *lut, int h, uint32_t b_hat, int lane, const char *gtab)
{
    const int rem = b_hat & (N - 1), cyc = b_hat >> LOG2N;
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int c = lane + 64 * r + FN * hh;
            int src = c + rem, ng = cyc & 1;
            if (src >= N) { src -= N; ng ^= 1; }
            const uint64_t v = lut[h * N + src];
            s.acc[2 * r + hh] = (ng ? (uint64_t)0 - v : v) - 1; // registers hold acc - 1
        }
    // W[4 l (t+1)] = stage-1 table entry of lane l2 = 4l mod 64, block c2 = 4l / 64 (+ t * 64 * 16)
    const int l2 = (4 * lane) & 63, c2 = (4 * lane) >> 6;
    s.t2_off = TAB0 + TAB_T1 + (c2 * 3 * 64 + t1_slot(l2)) * 16;
    (void)gtab;
}

template <int J0, int CNT> TFHE_DEV void ggsw_prefetch(Lane &s, const c64 *col, int lane)
{
#pragma unroll
    for (int j = J0; j < J0 + CNT; j++) {
        s.G[(j % PGRING) * 2 + 0] = col[(j * 2 + 0) * 64 + lane];
        s.G[(j % PGRING) * 2 + 1] = col[(j * 2 + 1) * 64 + lane];
    }
}

TFHE_DEV void phase_mirror_write(const Lane &s, char *lds, int poly_off, int lane)
{
    char *p = lds + opaque(poly_off + lane * 8);
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) st_u64(p, (64 * r + FN * hh) * 8, s.acc[2 * r + hh]);
}

// rotation by X^{a_hat}, subtraction, one-digit decomposition (NEGATED digits: see above), fold + forward twist.
// The lane reads 32 consecutive 64-coefficient blocks of the mirror starting at block bq = base >> 6; the 14-bit byte
// offset (block mod 32 : lane part) sits in the TOP bits of `A`, so that adding the block count t' << 27 wraps by
// itself and the carry-out of that addition IS the wrap -- which flips the sign; the LDS address is one
// v_alignbit_b32 (region index : offset).
template <int BASE_LOG> TFHE_DEV void phase_rotate_decompose_twist(Lane &s, const char *lds, int poly_off, int lane,
                                                                   uint32_t a_hat)
{
    const char *tab = lds + opaque(TAB0 + lane * 16);
    uint64_t rot[2][8];
    lane_mask pol[2][8];
    c64 tw[2][4];
    const uint32_t base = ((uint32_t)lane - a_hat) & (2 * N - 1);
    const uint32_t A = ((base >> 6) << 27) | ((base & 63) << 21); // bit 5 of the block number falls off the top: it is s0
    const lane_mask p0 = mask_of(((base >> 11) & 1u) != 0);        // !polarity (polarity p = !sign) of the first block
    const uint32_t region = (uint32_t)poly_off >> 14;
    auto issue = [&](int k, int slot) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 4 * k + i;
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                uint32_t sum;
                const lane_mask wrap = add_wraps(A, (uint32_t)(r + 16 * hh) << 27, sum);
                rot[slot][2 * i + hh] = ld_u64(lds, (int)alignbit(region, sum, 18));
                pol[slot][2 * i + hh] = wrap ^ p0;
            }
            tw[slot][i] = ld_c64(tab, TAB_TW + r * 64 * 16);
        }
    };
    issue(0, 0);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (k < 3) issue(k + 1, (k + 1) & 1);
        TFHE_SCHED_FENCE();
        uint32_t dhi[8];
#pragma unroll
        for (int g = 0; g < 2; g++) { // coefficients 4g .. 4g+3 of the chunk = points 2g, 2g+1
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
            s.R[4 * k + i] = cmul(dg, tw[k & 1][i]);
        }
    }
}

// stages 1 and 2 (in registers)
TFHE_DEV void fwd_stage12(Lane &s, const char *lds, int lane)
{
    const char *tab1 = lds + opaque(TAB0 + t1_slot(lane) * 16);
    c64 w[2][6];
    auto issue1 = [&](int k, int slot) { // twiddles of butterflies r0 = 2k, 2k+1: W[(l + 64 r0)(t+1)]
#pragma unroll
        for (int e = 0; e < 6; e++) w[slot][e] = ld_c64(tab1, TAB_T1 + ((2 * k) * 3 + e) * 64 * 16);
    };
    issue1(0, 0);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (k < 1) issue1(k + 1, (k + 1) & 1);
        else {
#pragma unroll
            for (int t = 0; t < 3; t++) w[0][t] = ld_c64(lds + s.t2_off, t * 64 * 16); // stage-2 twiddles
        }
        TFHE_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r0 = 2 * k + i;
            bfly4_fwd(s.R[r0], s.R[r0 + 4], s.R[r0 + 8], s.R[r0 + 12]);
            s.R[r0 + 4] = cmul(s.R[r0 + 4], w[k & 1][i * 3 + 0]);
            s.R[r0 + 8] = cmul(s.R[r0 + 8], w[k & 1][i * 3 + 1]);
            s.R[r0 + 12] = cmul(s.R[r0 + 12], w[k & 1][i * 3 + 2]);
        }
    }
#pragma unroll
    for (int blk = 0; blk < 4; blk++) { // stage 2: L = 256, Q = 64, jj = l ; twiddles in w[0][0..2]
        const int b = 4 * blk;
        bfly4_fwd(s.R[b], s.R[b + 1], s.R[b + 2], s.R[b + 3]);
        s.R[b + 1] = cmul(s.R[b + 1], w[0][0]);
        s.R[b + 2] = cmul(s.R[b + 2], w[0][1]);
        s.R[b + 3] = cmul(s.R[b + 3], w[0][2]);
    }
}
// --- stage 3, forward.  Pair m = registers (X, Y) = (R[2m], R[2m+1]).
//   [swap32(X, Y)]   X.upper <-> Y.lower : lanes < 32 hold (x@l, x@l+32) of register 2m, lanes >= 32 of 2m+1
TFHE_DEV void fwd_stage3_layer_a(Lane &s)
{
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const c64 lo = s.R[2 * m], hi = s.R[2 * m + 1];
        s.R[2 * m].re = lo.re + hi.re; s.R[2 * m].im = lo.im + hi.im;         // t0 (rows 0,2) / t2 (rows 1,3)
        s.R[2 * m + 1].re = lo.re - hi.re; s.R[2 * m + 1].im = lo.im - hi.im; // t1 / t3
    }
}
//   [swap16(X, Y)]   X.row1 <-> Y.row0, X.row3 <-> Y.row2 : even rows hold (t0, t2), odd rows (t1, t3)
TFHE_DEV void fwd_stage3_layer_b(Lane &s, const char *lds, int lane)
{
    const bool odd = (lane >> 4) & 1;
    const double sg = odd ? -1.0 : 1.0;
    const char *t3 = lds + opaque(TAB0 + TAB_T3 + lane * 16);
    const c64 T3x = ld_c64(t3, 0), T3y = ld_c64(t3, 64 * 16);
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const c64 x = s.R[2 * m], y = s.R[2 * m + 1];
        c64 u; // odd rows: t3 * -i = (im, -re): re/im swapped here, the sign is in sg
        u.re = odd ? y.im : y.re;
        u.im = odd ? y.re : y.im;
        c64 lo, hi;
        lo.re = x.re + u.re; lo.im = fma(u.im, sg, x.im);
        hi.re = x.re - u.re; hi.im = fma(u.im, -sg, x.im);
        s.R[2 * m] = cmul(lo, T3x);     // y0 (even rows, W^0) / y1 * W[16 jj]
        s.R[2 * m + 1] = cmul(hi, T3y); // y2 * W[32 jj]       / y3 * W[48 jj]
    }
}

// per-lane base addresses of the 4 (m & 1, s) classes of the swizzled transposition buffer
struct TrBases {
    int b[4];
};
TFHE_DEV TrBases tr_bases_a(int poly_off, int lane)
{
    const int jj = lane & 15, lp = 4 * (lane >> 5) + ((lane >> 4) & 1); // lane part of the destination row
    const int wb = poly_off + lp * 256 + ((jj ^ lp) << 4);
    TrBases t;
#pragma unroll
    for (int c = 0; c < 4; c++) t.b[c] = opaque(wb ^ ((8 * (c >> 1) + 2 * (c & 1)) << 4)); // c = 2 (m & 1) + s
    return t;
}
// byte offset added to the class base for pair m, slot s: row 8m + 2s
TFHE_DEV constexpr int tr_imm_a(int m, int s) { return (8 * m + 2 * s) * 256; }

TFHE_DEV void fwd_transpose_write(const Lane &s, char *lds, int poly_off, int lane)
{
    const TrBases t = tr_bases_a(poly_off, lane);
#pragma unroll
    for (int m = 0; m < 8; m++)
#pragma unroll
        for (int sl = 0; sl < 2; sl++) st_c64(lds + t.b[2 * (m & 1) + sl], tr_imm_a(m, sl), s.R[2 * m + sl]);
}

// pass B: lane l' reads positions 16 l' + j (entry (l', j) at row l', column j ^ (l' & 15)); stages 4, 5
TFHE_DEV void fwd_pass_b(Lane &s, const char *lds, int poly_off, int lane, const WaveConsts &k)
{
    const int a = opaque(poly_off + lane * 256 + ((lane & 15) << 4));
#pragma unroll
    for (int j = 0; j < 16; j++) s.R[j] = ld_c64(lds + (a ^ (j << 4)), 0);
    fwd_block16<0>(s, k);
}

// publish this wave's Fourier polynomial for the partner wave: [j][lane]
TFHE_DEV void exchange_write(const Lane &s, char *lds, int poly_off, int lane)
{
    char *p = lds + opaque(poly_off + lane * 16);
#pragma unroll
    for (int j = 0; j < 16; j++) st_c64(p, j * 1024, s.R[j]);
}

// out_h = ggsw[0][h] * F_0 (+)= ggsw[1][h] * F_1 ; own polynomial in R, the partner's read from LDS
template <int H> TFHE_DEV void phase_pointwise(Lane &s, const char *lds, int partner_off, const c64 *col, int lane)
{
    const char *p = lds + opaque(partner_off + lane * 16);
    c64 P[2][4];
    auto issue = [&](int q, int slot) {
#pragma unroll
        for (int i = 0; i < 4; i++) P[slot][i] = ld_c64(p, (4 * q + i) * 1024);
    };
    issue(0, 0);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (q < 3) issue(q + 1, (q + 1) & 1);
        TFHE_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int j = 4 * q + i;
            const c64 f0 = H ? P[q & 1][i] : s.R[j], f1 = H ? s.R[j] : P[q & 1][i];
            const c64 g0 = s.G[(j % PGRING) * 2 + 0], g1 = s.G[(j % PGRING) * 2 + 1];
            s.R[j] = cmul_add(g1, f1, cmul(g0, f0));
            if (j + PGRING < 16) {
                s.G[(j % PGRING) * 2 + 0] = col[((j + PGRING) * 2 + 0) * 64 + lane];
                s.G[(j % PGRING) * 2 + 1] = col[((j + PGRING) * 2 + 1) * 64 + lane];
                TFHE_SCHED_FENCE();
            }
        }
    }
}

TFHE_DEV void inv_pass_b(Lane &s, const WaveConsts &k) { inv_block16<0>(s, k); }
TFHE_DEV void inv_transpose_write(Lane &s, char *lds, int poly_off, int lane)
{
    const int a = opaque(poly_off + lane * 256 + ((lane & 15) << 4));
#pragma unroll
    for (int j = 0; j < 16; j++) st_c64(lds + (a ^ (j << 4)), 0, s.R[j]);
}

// pass A': read the pair layout, stage-3' twiddles and first layer
TFHE_DEV void inv_stage3_read_layer_a(Lane &s, const char *lds, int poly_off, int lane)
{
    const TrBases t = tr_bases_a(poly_off, lane);
    const char *t3 = lds + opaque(TAB0 + TAB_T3 + lane * 16);
    const c64 T3x = ld_c64(t3, 0), T3y = ld_c64(t3, 64 * 16);
    c64 X[2][4], Y[2][4];
    auto issue = [&](int q, int slot) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int m = 4 * q + i;
            X[slot][i] = ld_c64(lds + t.b[2 * (m & 1) + 0], tr_imm_a(m, 0));
            Y[slot][i] = ld_c64(lds + t.b[2 * (m & 1) + 1], tr_imm_a(m, 1));
        }
    };
    issue(0, 0);
#pragma unroll
    for (int q = 0; q < 2; q++) {
        if (q < 1) issue(q + 1, (q + 1) & 1);
        TFHE_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int m = 4 * q + i;
            const c64 lo = cmul_conj(X[q & 1][i], T3x), hi = cmul_conj(Y[q & 1][i], T3y);
            s.R[2 * m].re = lo.re + hi.re; s.R[2 * m].im = lo.im + hi.im;         // t0 (even rows) / t2 (odd)
            s.R[2 * m + 1].re = lo.re - hi.re; s.R[2 * m + 1].im = lo.im - hi.im; // t1 / t3
        }
    }
}
//   [swap16(X, Y)] : even rows hold (t0, t2), odd rows (t1, t3)
TFHE_DEV void inv_stage3_layer_b(Lane &s, int lane)
{
    const bool odd = (lane >> 4) & 1;
    const double sg = odd ? -1.0 : 1.0;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const c64 x = s.R[2 * m], y = s.R[2 * m + 1];
        c64 u; // odd rows: i * t3 = (-im, re): re/im swapped here, the sign is in sg
        u.re = odd ? y.im : y.re;
        u.im = odd ? y.re : y.im;
        s.R[2 * m].re = fma(u.re, sg, x.re);      s.R[2 * m].im = x.im + u.im;     // x0 / x1
        s.R[2 * m + 1].re = fma(u.re, -sg, x.re); s.R[2 * m + 1].im = x.im - u.im; // x2 / x3
    }
}
//   [swap32(X, Y)] : natural pass-A layout again
TFHE_DEV void inv_stage21_accumulate(Lane &s, const char *lds, int lane)
{
    const char *tab = lds + opaque(TAB0 + lane * 16);
    const char *tab1 = lds + opaque(TAB0 + t1_slot(lane) * 16);
    c64 w[2][6];
    auto issue1 = [&](int k, int slot) {
#pragma unroll
        for (int e = 0; e < 6; e++) w[slot][e] = ld_c64(tab1, TAB_T1 + ((2 * k) * 3 + e) * 64 * 16);
    };
#pragma unroll
    for (int t = 0; t < 3; t++) w[1][t] = ld_c64(lds + s.t2_off, t * 64 * 16);
    issue1(0, 0);
    TFHE_SCHED_FENCE();
#pragma unroll
    for (int blk = 0; blk < 4; blk++) { // stage 2' (twiddles in w[1][0..2])
        const int b = 4 * blk;
        s.R[b + 1] = cmul_conj(s.R[b + 1], w[1][0]);
        s.R[b + 2] = cmul_conj(s.R[b + 2], w[1][1]);
        s.R[b + 3] = cmul_conj(s.R[b + 3], w[1][2]);
        bfly4_inv(s.R[b], s.R[b + 1], s.R[b + 2], s.R[b + 3]);
    }
    c64 tw[2][4];
    auto issue_tw = [&](int q, int slot) {
#pragma unroll
        for (int i = 0; i < 4; i++) tw[slot][i] = ld_c64(tab, TAB_TW + (4 * q + i) * 64 * 16);
    };
#pragma unroll
    for (int k = 0; k < 2; k++) { // stage 1'
        if (k < 1) issue1(k + 1, (k + 1) & 1);
        else issue_tw(0, 0);
        TFHE_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r0 = 2 * k + i;
            s.R[r0 + 4] = cmul_conj(s.R[r0 + 4], w[k & 1][i * 3 + 0]);
            s.R[r0 + 8] = cmul_conj(s.R[r0 + 8], w[k & 1][i * 3 + 1]);
            s.R[r0 + 12] = cmul_conj(s.R[r0 + 12], w[k & 1][i * 3 + 2]);
            bfly4_inv(s.R[r0], s.R[r0 + 4], s.R[r0 + 8], s.R[r0 + 12]);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) { // backward twist + integer-domain rounding
        if (q < 3) issue_tw(q + 1, (q + 1) & 1);
        TFHE_SCHED_FENCE();
        // w = x * conj(tw) carries the key's prescale: the torus increment in revolutions.  Fast path: every |w| >= 2 (bit 30 of
        // the high words): fract(w) + 1 is exact and its fraction bits are the increment.
        c64 v[4];
        uint32_t ok = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            v[i] = cmul_conj(s.R[4 * q + i], tw[q & 1][i]);
            ok = and3(ok, f64_hi(v[i].re), f64_hi(v[i].im));
        }
        if (__builtin_expect((ok & 0x40000000u) == 0, 0)) { // some exponent below 1024
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

// sample extraction at X^0: wave 0 writes the mask, wave 1 the body
TFHE_DEV void phase_sample_extract(const Lane &s, uint64_t *out, int h, int lane)
{
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int c = lane + 64 * r + FN * hh;
            const uint64_t v = s.acc[2 * r + hh] + 1;
            if (h == 0) out[c == 0 ? 0 : N - c] = (c == 0) ? v : (uint64_t)0 - v;
            else if (c == 0) out[N] = v;
        }
}

} // namespace pair
} // namespace tfhe_hip
