// wave_emu.cpp -- HOST EMULATION of the wave-per-ciphertext kernel (test infrastructure only).
//
// Compiles tfhe_rs_string_amd/csrc/wave_pbs.hpp with -DTFHE_HIP_HOST_EMU and runs its lane-local
// phases for lanes 0..63 in turn, with LDS as a plain byte array and the permlane32 swap done by
// hand.  Lets the CPU test-suite check the kernel's index maps and arithmetic DAG bit-for-bit
// against the oracle.  It is not part of the product library and is never used as a fallback.
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include "../../tfhe_rs_string_amd/csrc/wave_tables.hpp"

using namespace tfhe_hip;
using namespace tfhe_hip::wave;

extern "C" void tfhe_hip_make_tables(uint32_t n, double *tw, double *w);

static WaveConsts make_consts(const double *w)
{
    WaveConsts K;
    const int idx[4] = {64, 128, 192, 384};
    for (int i = 0; i < 4; i++) {
        K.w[2 * i] = w[2 * idx[i]];
        K.w[2 * i + 1] = w[2 * idx[i] + 1];
    }
    return K;
}

static void swap_phase(std::vector<Lane> &L)
{
    for (int l = 0; l < 32; l++)
        for (int j = 0; j < 16; j++) std::swap(L[l + 32].R[j], L[l].R[16 + j]);
}

// bsk_f: oracle order [n][4 (row*2+col)][FN] interleaved (re, im); lwe_small n+1; lut 2N; out N+1
extern "C" int wave_emu_blind_rotate(const double *bsk_f, const uint64_t *lwe_small, const uint64_t *lut,
                                     uint64_t *lwe_out, int n)
{
    std::vector<double> tw(2 * FN), w(2 * FN);
    tfhe_hip_make_tables(FN, tw.data(), w.data());
    std::vector<char> gtab = build_tables(tw.data(), w.data());
    std::vector<char> lds(LDS_BYTES, 0);
    std::memcpy(lds.data() + TAB_TW, gtab.data(), TAB_BYTES);
    // lane-ordered BSK, same map as bsk_permute_wave_kernel
    std::vector<c64> bsk_w((size_t)n * 4 * FN);
    const c64 *src = reinterpret_cast<const c64 *>(bsk_f);
    for (size_t i = 0; i < (size_t)n; i++)
        for (int e = 0; e < 64; e++)
            for (int lane = 0; lane < 64; lane++)
                bsk_w[(i * 64 + e) * 64 + lane] = src[(i * 4 + (e & 3)) * FN + bsk_src_pos(lane, e >> 2)];

    std::vector<Lane> L(64);
    auto poly = [&](int lane) { return WAVE0 + (lane >> 5) * POLY_BYTES; }; // wave 0
#define FOR_LANES for (int lane = 0; lane < 64; lane++)
    const WaveConsts K = make_consts(w.data());
    FOR_LANES phase_init(L[lane], lut, modswitch(lwe_small[n], LOG2N), lane);
    for (int i = 0; i < n; i++) {
        const uint32_t a_hat = modswitch(lwe_small[i], LOG2N);
        const c64 *ggsw = bsk_w.data() + (size_t)i * 4 * FN;
        FOR_LANES ggsw_prefetch<0, GRING>(L[lane], ggsw, lane);
        FOR_LANES phase_mirror_write(L[lane], lds.data(), poly(lane), lane);
        FOR_LANES phase_rotate_decompose_twist<23>(L[lane], lds.data(), poly(lane), lane, a_hat);
        FOR_LANES fwd_pass_a(L[lane], lds.data(), lane);
        FOR_LANES fwd_transpose_write(L[lane], lds.data(), poly(lane), lane);
        FOR_LANES fwd_pass_b(L[lane], lds.data(), poly(lane), lane, K);
        swap_phase(L);
        FOR_LANES phase_pointwise(L[lane], ggsw, lane);
        swap_phase(L);
        FOR_LANES inv_pass_b(L[lane], lds.data(), lane, K);
        FOR_LANES inv_transpose_write(L[lane], lds.data(), poly(lane), lane);
        FOR_LANES inv_pass_a(L[lane], lds.data(), poly(lane), lane);
    }
    FOR_LANES phase_sample_extract(L[lane], lwe_out, lane);
    return 0;
}

// forward transform of one digit polynomial pair through the wave data path (both halves get the
// same polynomial); returns the Fourier values in oracle (DIF position) order for half 0
extern "C" int wave_emu_forward(const int64_t *digits, double *out_re, double *out_im)
{
    std::vector<double> tw(2 * FN), w(2 * FN);
    tfhe_hip_make_tables(FN, tw.data(), w.data());
    std::vector<char> gtab = build_tables(tw.data(), w.data());
    std::vector<char> lds(LDS_BYTES, 0);
    std::memcpy(lds.data() + TAB_TW, gtab.data(), TAB_BYTES);
    std::vector<Lane> L(64);
    auto poly = [&](int lane) { return WAVE0 + (lane >> 5) * POLY_BYTES; };
    std::vector<uint64_t> zero_lut(2 * N, 0);
    const WaveConsts K = make_consts(w.data());
    FOR_LANES phase_init(L[lane], zero_lut.data(), 0, lane);
    FOR_LANES for (int r = 0; r < 32; r++) {
        const int l = lane & 31, q = l + 32 * r;
        c64 d;
        d.re = (double)digits[q];
        d.im = (double)digits[q + FN];
        L[lane].R[r] = cmul(d, ld_c64(lds.data(), TAB_TW + (r * 32 + l) * 16));
    }
    FOR_LANES fwd_pass_a(L[lane], lds.data(), lane);
    FOR_LANES fwd_transpose_write(L[lane], lds.data(), poly(lane), lane);
    FOR_LANES fwd_pass_b(L[lane], lds.data(), poly(lane), lane, K);
    for (int lp = 0; lp < 32; lp++)
        for (int j = 0; j < 32; j++) {
            out_re[fourier_pos(lp, j)] = L[lp].R[j].re;
            out_im[fourier_pos(lp, j)] = L[lp].R[j].im;
        }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// variant 2 (pair_pbs.hpp): two waves per ciphertext.  Lanes 0..63 = wave 0 (mask polynomial),
// 64..127 = wave 1 (body polynomial); barriers are the boundaries between the loops below.
// ------------------------------------------------------------------------------------------------
namespace P = tfhe_hip::pair;

static void emu_swap32(P::Lane *W) // one wave: X.upper <-> Y.lower for every pair
{
    for (int l = 0; l < 32; l++)
        for (int m = 0; m < 8; m++) std::swap(W[l + 32].R[2 * m], W[l].R[2 * m + 1]);
}
static void emu_swap16(P::Lane *W) // X.row1 <-> Y.row0, X.row3 <-> Y.row2
{
    for (int half = 0; half < 2; half++)
        for (int l = 0; l < 16; l++)
            for (int m = 0; m < 8; m++) std::swap(W[32 * half + 16 + l].R[2 * m], W[32 * half + l].R[2 * m + 1]);
}

extern "C" int pair_emu_blind_rotate(const double *bsk_f, const uint64_t *lwe_small, const uint64_t *lut,
                                     uint64_t *lwe_out, int n)
{
    std::vector<double> tw(2 * FN), w(2 * FN);
    tfhe_hip_make_tables(FN, tw.data(), w.data());
    std::vector<char> gtab = P::build_tables(tw.data(), w.data());
    std::vector<char> lds(P::LDS_BYTES, 0);
    std::memcpy(lds.data() + P::TAB0, gtab.data(), P::TAB_BYTES);
    const WaveConsts K = make_consts(w.data());
    std::vector<c64> bsk_p((size_t)n * 4 * FN);
    const c64 *src = reinterpret_cast<const c64 *>(bsk_f);
    for (size_t i = 0; i < (size_t)n; i++)
        for (int h = 0; h < 2; h++)
            for (int j = 0; j < 16; j++)
                for (int row = 0; row < 2; row++)
                    for (int lane = 0; lane < 64; lane++)
                    {
                        c64 g = src[(i * 4 + row * 2 + h) * FN + 16 * lane + j];
                        g.re *= P::BSK_PRESCALE;
                        g.im *= P::BSK_PRESCALE;
                        bsk_p[P::bsk_pair_index(i, h, j, row, lane)] = g;
                    }
    std::vector<P::Lane> L(128);
    auto off = [&](int wv) { return P::WAVE0 + wv * P::WAVE_BYTES; }; // waves 0,1 of the workgroup
#define FOR_WL for (int wv = 0; wv < 2; wv++) for (int lane = 0; lane < 64; lane++)
#define LN L[wv * 64 + lane]
    FOR_WL P::phase_init(LN, lut, wv, modswitch(lwe_small[n], LOG2N), lane, gtab.data());
    for (int i = 0; i < n; i++) {
        const uint32_t a_hat = modswitch(lwe_small[i], LOG2N);
        const c64 *col[2] = {bsk_p.data() + P::bsk_pair_index(i, 0, 0, 0, 0), bsk_p.data() + P::bsk_pair_index(i, 1, 0, 0, 0)};
        FOR_WL P::ggsw_prefetch<0, P::PGRING>(LN, col[wv], lane);
        FOR_WL P::phase_mirror_write(LN, lds.data(), off(wv), lane);
        FOR_WL P::phase_rotate_decompose_twist<23>(LN, lds.data(), off(wv), lane, a_hat);
        FOR_WL P::fwd_stage12(LN, lds.data(), lane);
        for (int wv = 0; wv < 2; wv++) emu_swap32(&L[wv * 64]);
        FOR_WL P::fwd_stage3_layer_a(LN);
        for (int wv = 0; wv < 2; wv++) emu_swap16(&L[wv * 64]);
        FOR_WL P::fwd_stage3_layer_b(LN, lds.data(), lane);
        FOR_WL P::fwd_transpose_write(LN, lds.data(), off(wv), lane);
        FOR_WL P::fwd_pass_b(LN, lds.data(), off(wv), lane, K);
        FOR_WL P::exchange_write(LN, lds.data(), off(wv), lane);
        // barrier 1
        for (int lane = 0; lane < 64; lane++) P::phase_pointwise<0>(L[lane], lds.data(), off(1), col[0], lane);
        for (int lane = 0; lane < 64; lane++) P::phase_pointwise<1>(L[64 + lane], lds.data(), off(0), col[1], lane);
        // barrier 2
        FOR_WL P::inv_pass_b(LN, K);
        FOR_WL P::inv_transpose_write(LN, lds.data(), off(wv), lane);
        FOR_WL P::inv_stage3_read_layer_a(LN, lds.data(), off(wv), lane);
        for (int wv = 0; wv < 2; wv++) emu_swap16(&L[wv * 64]);
        FOR_WL P::inv_stage3_layer_b(LN, lane);
        for (int wv = 0; wv < 2; wv++) emu_swap32(&L[wv * 64]);
        FOR_WL P::inv_stage21_accumulate(LN, lds.data(), lane);
    }
    FOR_WL P::phase_sample_extract(LN, lwe_out, wv, lane);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// variant 3 (block_pbs.hpp): one 512-thread workgroup per ciphertext.  Every phase is run for all
// 512 threads before the next one starts (a superset of the kernel's barriers / in-order DS rules).
// ------------------------------------------------------------------------------------------------
#include "../../tfhe_rs_string_amd/csrc/block_pbs.hpp"
namespace B = tfhe_hip::blk;

static void emu_transpose54(B::Lane *W) // one wavefront: register index <-> lane bits (5,4)
{
    for (int l = 0; l < 32; l++)
        for (int m = 0; m < 2; m++) std::swap(W[l + 32].R[m], W[l].R[m + 2]);
    for (int half = 0; half < 2; half++)
        for (int l = 0; l < 16; l++)
            for (int m = 0; m < 4; m += 2) std::swap(W[32 * half + 16 + l].R[m], W[32 * half + l].R[m + 1]);
}

extern "C" int block_emu_blind_rotate(const double *bsk_f, const uint64_t *lwe_small, const uint64_t *lut,
                                      uint64_t *lwe_out, int n)
{
    std::vector<double> tw(2 * FN), w(2 * FN);
    tfhe_hip_make_tables(FN, tw.data(), w.data());
    const c64 *twist = reinterpret_cast<const c64 *>(tw.data());
    const c64 *W = reinterpret_cast<const c64 *>(w.data());
    std::vector<char> lds(B::LDS_BYTES, 0);
    std::vector<c64> bsk_b((size_t)n * 4 * FN);
    const c64 *src = reinterpret_cast<const c64 *>(bsk_f);
    for (size_t i = 0; i < (size_t)n; i++)
        for (int p = 0; p < 2; p++)
            for (int q = 0; q < 4; q++)
                for (int e = 0; e < 8; e++)
                    for (int lane = 0; lane < 64; lane++)
                    {
                        c64 g = src[(i * 4 + (e >> 2) * 2 + p) * FN + 256 * q + 4 * lane + (e & 3)];
                        g.re *= B::BSK_PRESCALE;
                        g.im *= B::BSK_PRESCALE;
                        bsk_b[B::bsk_block_index(i, p, q, e, lane)] = g;
                    }
    std::vector<B::Lane> L(B::THREADS);
#define FOR_T for (int tid = 0; tid < B::THREADS; tid++)
    FOR_T B::phase_init(L[tid], lds.data(), lut, twist, W, modswitch(lwe_small[n], LOG2N), B::make_ids(tid));
    for (int i = 0; i < n; i++) {
        const uint32_t a_hat = modswitch(lwe_small[i], LOG2N);
        FOR_T B::ggsw_prefetch(L[tid], bsk_b.data(), (size_t)i, B::make_ids(tid));
        FOR_T B::phase_rotate_stage1<23>(L[tid], lds.data(), a_hat, B::make_ids(tid));
        FOR_T B::phase_fwd2(L[tid], lds.data(), B::make_ids(tid));
        for (int wv = 0; wv < 8; wv++) emu_transpose54(&L[wv * 64]);
        FOR_T B::phase_fwd3(L[tid], lds.data(), B::make_ids(tid));
        FOR_T B::phase_fwd4(L[tid], lds.data(), B::make_ids(tid));
        FOR_T B::phase_pointwise(L[tid], lds.data(), B::make_ids(tid));
        FOR_T B::phase_inv4(L[tid], lds.data(), B::make_ids(tid));
        FOR_T B::phase_inv3(L[tid], lds.data(), B::make_ids(tid));
        for (int wv = 0; wv < 8; wv++) emu_transpose54(&L[wv * 64]);
        FOR_T B::phase_inv2(L[tid], lds.data(), B::make_ids(tid));
        FOR_T B::phase_inv1_accumulate(L[tid], lds.data(), B::make_ids(tid));
    }
    FOR_T B::phase_sample_extract(L[tid], lwe_out, B::make_ids(tid));
    return 0;
}

// variant 4 (blind_rotate_half_kernel): the same phases, thread (q, l) of a 256-thread workgroup doing both polynomials of
// its positions, in the PACKED LDS layout (the mirror inside the inverse image) with the fused external product
extern "C" int half_emu_blind_rotate(const double *bsk_f, const uint64_t *lwe_small, const uint64_t *lut,
                                     uint64_t *lwe_out, int n)
{
    std::vector<double> tw(2 * FN), w(2 * FN);
    tfhe_hip_make_tables(FN, tw.data(), w.data());
    const c64 *twist = reinterpret_cast<const c64 *>(tw.data());
    const c64 *W = reinterpret_cast<const c64 *>(w.data());
    std::vector<char> lds(B::LDS_BYTES_PACKED, 0);
    std::vector<c64> bsk_b((size_t)n * 4 * FN);
    const c64 *src = reinterpret_cast<const c64 *>(bsk_f);
    for (size_t i = 0; i < (size_t)n; i++)
        for (int p = 0; p < 2; p++)
            for (int q = 0; q < 4; q++)
                for (int e = 0; e < 8; e++)
                    for (int lane = 0; lane < 64; lane++)
                    {
                        c64 g = src[(i * 4 + (e >> 2) * 2 + p) * FN + 256 * q + 4 * lane + (e & 3)];
                        g.re *= B::BSK_PRESCALE;
                        g.im *= B::BSK_PRESCALE;
                        bsk_b[B::bsk_block_index(i, p, q, e, lane)] = g;
                    }
    std::vector<B::Lane> L(B::THREADS); // L[tid] = virtual thread tid of the 512-thread kernel; real thread = tid & 255
    FOR_T B::phase_init<true>(L[tid], lds.data(), lut, twist, W, modswitch(lwe_small[n], LOG2N), B::make_ids(tid));
    for (int i = 0; i < n; i++) {
        const uint32_t a_hat = modswitch(lwe_small[i], LOG2N);
        FOR_T B::ggsw_prefetch(L[tid], bsk_b.data(), (size_t)i, B::make_ids(tid));
        FOR_T B::phase_rotate_stage1<23, true>(L[tid], lds.data(), a_hat, B::make_ids(tid));
        FOR_T B::phase_fwd2(L[tid], lds.data(), B::make_ids(tid));
        for (int wv = 0; wv < 8; wv++) emu_transpose54(&L[wv * 64]);
        FOR_T B::phase_fwd3(L[tid], lds.data(), B::make_ids(tid));
        FOR_T B::phase_fwd4(L[tid], lds.data(), B::make_ids(tid));
        for (int t2 = 0; t2 < 256; t2++) B::phase_pointwise_both(L[t2], L[t2 + 256], lds.data(), B::make_ids(t2));
        FOR_T B::phase_inv4(L[tid], lds.data(), B::make_ids(tid));
        FOR_T B::phase_inv3(L[tid], lds.data(), B::make_ids(tid));
        for (int wv = 0; wv < 8; wv++) emu_transpose54(&L[wv * 64]);
        FOR_T B::phase_inv2(L[tid], lds.data(), B::make_ids(tid));
        // PACKED: a thread's stage-1' reads and mirror writes hit the same slots -- run thread by thread like the kernel
        FOR_T B::phase_inv1_accumulate<true>(L[tid], lds.data(), B::make_ids(tid));
    }
    FOR_T B::phase_sample_extract(L[tid], lwe_out, B::make_ids(tid));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// multi-bit blind rotation (block_mb_pbs.hpp), grouping factor g = 2 or 3
// bsk_f: oracle order [n/g groups][2^g][row*2+col][FN] interleaved (re, im)
// ------------------------------------------------------------------------------------------------
#include "../../tfhe_rs_string_amd/csrc/block_mb_pbs.hpp"
namespace MB = tfhe_hip::blkmb;

template <int BASE_LOG, int G>
static int block_mb_emu(const double *bsk_f, const uint64_t *lwe_small, const uint64_t *lut, uint64_t *lwe_out, int n)
{
    std::vector<double> tw(2 * FN), w(2 * FN);
    tfhe_hip_make_tables(FN, tw.data(), w.data());
    const c64 *twist = reinterpret_cast<const c64 *>(tw.data());
    const c64 *W = reinterpret_cast<const c64 *>(w.data());
    std::vector<char> lds(B::LDS_BYTES, 0);
    const size_t groups = (size_t)n / G, per = (size_t)1 << G;
    std::vector<c64> bsk_mb(groups * per * 4 * FN);
    const c64 *src = reinterpret_cast<const c64 *>(bsk_f);
    for (size_t grp = 0; grp < groups; grp++)
        for (size_t idx = 0; idx < per; idx++)
            for (int p = 0; p < 2; p++)
                for (int q = 0; q < 4; q++)
                    for (int e = 0; e < 8; e++)
                        for (int lane = 0; lane < 64; lane++)
                            bsk_mb[MB::bsk_mb_index<G>(grp, (int)idx, p, q, e, lane)] =
                                src[((grp * per + idx) * 4 + (e >> 2) * 2 + p) * FN + 256 * q + 4 * lane + (e & 3)];
    std::vector<MB::Lane> L(B::THREADS);
    FOR_T MB::phase_init(L[tid], lut, twist, W, modswitch(lwe_small[n], LOG2N), B::make_ids(tid));
    for (size_t grp = 0; grp < groups; grp++) {
        // the kernel spreads these slots over two CMUX-free steps; their order is what matters
        FOR_T MB::mb_slot<G, 0>(L[tid], bsk_mb.data(), grp, lwe_small, twist, W, B::make_ids(tid));
        FOR_T MB::mb_slot<G, 1>(L[tid], bsk_mb.data(), grp, lwe_small, twist, W, B::make_ids(tid));
        FOR_T MB::mb_slot<G, 2>(L[tid], bsk_mb.data(), grp, lwe_small, twist, W, B::make_ids(tid));
        FOR_T MB::mb_slot<G, 3>(L[tid], bsk_mb.data(), grp, lwe_small, twist, W, B::make_ids(tid));
        FOR_T MB::mb_slot<G, 4>(L[tid], bsk_mb.data(), grp, lwe_small, twist, W, B::make_ids(tid));
        FOR_T MB::mb_slot<G, 5>(L[tid], bsk_mb.data(), grp, lwe_small, twist, W, B::make_ids(tid));
        FOR_T MB::mb_slot<G, 6>(L[tid], bsk_mb.data(), grp, lwe_small, twist, W, B::make_ids(tid));
        FOR_T MB::mb_slot<G, 7>(L[tid], bsk_mb.data(), grp, lwe_small, twist, W, B::make_ids(tid));
        FOR_T MB::phase_decompose_stage1<BASE_LOG>(L[tid], lds.data(), B::make_ids(tid));
        FOR_T B::phase_fwd2(L[tid], lds.data(), B::make_ids(tid));
        for (int wv = 0; wv < 8; wv++) {
            MB::Lane *Wv = &L[wv * 64];
            for (int l = 0; l < 32; l++)
                for (int m = 0; m < 2; m++) std::swap(Wv[l + 32].R[m], Wv[l].R[m + 2]);
            for (int half = 0; half < 2; half++)
                for (int l = 0; l < 16; l++)
                    for (int m = 0; m < 4; m += 2) std::swap(Wv[32 * half + 16 + l].R[m], Wv[32 * half + l].R[m + 1]);
        }
        FOR_T B::phase_fwd3(L[tid], lds.data(), B::make_ids(tid));
        FOR_T B::phase_fwd4(L[tid], lds.data(), B::make_ids(tid));
        FOR_T MB::phase_pointwise(L[tid], lds.data(), B::make_ids(tid));
        FOR_T B::phase_inv4(L[tid], lds.data(), B::make_ids(tid));
        FOR_T B::phase_inv3(L[tid], lds.data(), B::make_ids(tid));
        for (int wv = 0; wv < 8; wv++) {
            MB::Lane *Wv = &L[wv * 64];
            for (int l = 0; l < 32; l++)
                for (int m = 0; m < 2; m++) std::swap(Wv[l + 32].R[m], Wv[l].R[m + 2]);
            for (int half = 0; half < 2; half++)
                for (int l = 0; l < 16; l++)
                    for (int m = 0; m < 4; m += 2) std::swap(Wv[32 * half + 16 + l].R[m], Wv[32 * half + l].R[m + 1]);
        }
        FOR_T B::phase_inv2(L[tid], lds.data(), B::make_ids(tid));
        FOR_T MB::phase_inv1_replace(L[tid], lds.data(), B::make_ids(tid));
    }
    FOR_T B::phase_sample_extract_plain(L[tid], lwe_out, B::make_ids(tid), 0);
    return 0;
}

extern "C" int block_mb_emu_blind_rotate(const double *bsk_f, const uint64_t *lwe_small, const uint64_t *lut,
                                         uint64_t *lwe_out, int n, int g)
{
    if (g == 2) return block_mb_emu<22, 2>(bsk_f, lwe_small, lut, lwe_out, n);
    if (g == 3) return block_mb_emu<21, 3>(bsk_f, lwe_small, lut, lwe_out, n);
    return 1;
}

// ------------------------------------------------------------------------------------------------
// variant 5 (quad_pbs.hpp): four wavefronts per ciphertext, 256 threads.  Every phase is run for all threads before the
// next one starts; the permlane swaps are done by hand.
// ------------------------------------------------------------------------------------------------
#include "../../tfhe_rs_string_amd/csrc/quad_pbs.hpp"
namespace Q = tfhe_hip::quad;

static void emu_q_swap32(Q::Lane *W) // one wavefront: (R[2k], R[2k+1]): X.upper <-> Y.lower
{
    for (int l = 0; l < 32; l++)
        for (int k = 0; k < 4; k++) std::swap(W[l + 32].R[2 * k], W[l].R[2 * k + 1]);
}
static void emu_q_swap16(Q::Lane *W) // (R[x], R[x+2]), x = 0, 1, 4, 5: X.row1 <-> Y.row0, X.row3 <-> Y.row2
{
    for (int half = 0; half < 2; half++)
        for (int l = 0; l < 16; l++)
            for (int k = 0; k < 4; k++) {
                const int x = (k & 1) + 4 * (k >> 1);
                std::swap(W[32 * half + 16 + l].R[x], W[32 * half + l].R[x + 2]);
            }
}

extern "C" int quad_emu_blind_rotate(const double *bsk_f, const uint64_t *lwe_small, const uint64_t *lut,
                                     uint64_t *lwe_out, int n)
{
    std::vector<double> tw(2 * FN), w(2 * FN);
    tfhe_hip_make_tables(FN, tw.data(), w.data());
    const c64 *twist = reinterpret_cast<const c64 *>(tw.data());
    const c64 *W = reinterpret_cast<const c64 *>(w.data());
    std::vector<char> lds(Q::LDS_BYTES, 0);
    std::vector<c64> bsk_q((size_t)n * 4 * FN);
    const c64 *src = reinterpret_cast<const c64 *>(bsk_f);
    for (size_t i = 0; i < (size_t)n; i++)
        for (int p = 0; p < 2; p++)
            for (int wv = 0; wv < 2; wv++)
                for (int reg = 0; reg < 8; reg++)
                    for (int row = 0; row < 2; row++)
                        for (int lane = 0; lane < 64; lane++) {
                            c64 g = src[(i * 4 + row * 2 + p) * FN + Q::fourier_pos(wv, lane, reg)];
                            g.re *= Q::BSK_PRESCALE;
                            g.im *= Q::BSK_PRESCALE;
                            bsk_q[Q::bsk_quad_index(i, p, wv, reg, row, lane)] = g;
                        }
    std::vector<Q::Lane> L(Q::THREADS);
#define FOR_Q for (int tid = 0; tid < Q::THREADS; tid++)
#define QSWAP32 for (int wv = 0; wv < 4; wv++) emu_q_swap32(&L[wv * 64])
#define QSWAP16 for (int wv = 0; wv < 4; wv++) emu_q_swap16(&L[wv * 64])
    FOR_Q Q::phase_init(L[tid], lds.data(), lut, twist, W, modswitch(lwe_small[n], LOG2N), Q::make_ids(tid), tid);
    for (int i = 0; i < n; i++) {
        const uint32_t a_hat = modswitch(lwe_small[i], LOG2N);
        auto col = [&](int tid) {
            const Q::Ids d = Q::make_ids(tid);
            return bsk_q.data() + Q::bsk_quad_index((size_t)i, d.p, d.w, 0, 0, 0);
        };
        FOR_Q Q::ggsw_prefetch<0, 8>(L[tid], col(tid), tid & 63);
        FOR_Q Q::phase_mirror_write(L[tid], lds.data(), Q::make_ids(tid));
        FOR_Q Q::phase_rotate_decompose_twist<23>(L[tid], lds.data(), Q::make_ids(tid), a_hat);
        FOR_Q Q::fwd_stage1(L[tid]);
        QSWAP32;
        FOR_Q Q::fwd_stage2(L[tid], Q::load_w2(lds.data(), Q::make_ids(tid)));
        QSWAP16;
        FOR_Q Q::fwd_stage3_layer_a(L[tid]);
        FOR_Q Q::transpose_write_a(L[tid], lds.data(), Q::make_ids(tid));
        FOR_Q Q::transpose_read_b(L[tid], lds.data(), Q::make_ids(tid));
        FOR_Q Q::fwd_pass_b1(L[tid], Q::load_twb(lds.data(), Q::make_ids(tid)), Q::make_ids(tid));
        QSWAP32;
        QSWAP16;
        FOR_Q Q::fwd_stage5(L[tid]);
        FOR_Q Q::exchange_write(L[tid], lds.data(), Q::make_ids(tid));
        FOR_Q {
            if (tid < 128) Q::phase_pointwise<0>(L[tid], lds.data(), Q::make_ids(tid));
            else Q::phase_pointwise<1>(L[tid], lds.data(), Q::make_ids(tid));
        }
        FOR_Q Q::inv_stage5(L[tid]);
        QSWAP16;
        QSWAP32;
        FOR_Q Q::inv_pass_b1(L[tid], Q::load_twb(lds.data(), Q::make_ids(tid)));
        FOR_Q Q::transpose_write_b(L[tid], lds.data(), Q::make_ids(tid));
        FOR_Q Q::transpose_read_a(L[tid], lds.data(), Q::make_ids(tid));
        FOR_Q Q::inv_stage3_layer_b(L[tid], Q::make_ids(tid));
        QSWAP16;
        FOR_Q Q::inv_stage2(L[tid], Q::load_w2(lds.data(), Q::make_ids(tid)));
        QSWAP32;
        FOR_Q Q::inv_stage1_accumulate(L[tid]);
    }
    FOR_Q Q::phase_sample_extract(L[tid], lwe_out, Q::make_ids(tid));
    return 0;
}

// the LDS slots of the transposition: every slot written exactly once per transform, and the lane groups of both
// directions conflict-free (returns the number of violations)
extern "C" int quad_emu_check_transposition(void)
{
    int bad = 0;
    std::vector<int> hit(1024, 0);
    for (int tid = 0; tid < 128; tid++)
        for (int r = 0; r < 8; r++) {
            const int a = Q::tr_base_a(tid >> 6, tid & 63) ^ Q::tr_xor_a(r);
            if (a & 15 || a < 0 || a >= 16384) { bad++; continue; }
            hit[a >> 4]++;
        }
    for (int v : hit) bad += v != 1;
    std::fill(hit.begin(), hit.end(), 0);
    for (int tid = 0; tid < 128; tid++)
        for (int j = 0; j < 8; j++) {
            const int a = (Q::tr_base_b(tid >> 6, tid & 63) ^ Q::tr_xor_b(j)) + j * 2048;
            if (a & 15 || a < 0 || a >= 16384) { bad++; continue; }
            hit[a >> 4]++;
        }
    for (int v : hit) bad += v != 1;
    // ds_write_b128: groups of 8 contiguous lanes must cover 8 distinct 16-B columns (mod 128 B)
    for (int wv = 0; wv < 2; wv++)
        for (int r = 0; r < 8; r++)
            for (int g = 0; g < 8; g++) {
                int cols = 0;
                for (int l = 8 * g; l < 8 * g + 8; l++) cols |= 1 << (((Q::tr_base_a(wv, l) ^ Q::tr_xor_a(r)) >> 4) & 7);
                bad += cols != 0xFF;
            }
    // ds_read_b128: groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32): 16 distinct columns (mod 256 B)
    static const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                   {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    for (int wv = 0; wv < 2; wv++)
        for (int j = 0; j < 8; j++)
            for (int h = 0; h < 2; h++)
                for (int g = 0; g < 2; g++) {
                    int cols_b = 0, cols_a = 0;
                    for (int e = 0; e < 16; e++) {
                        const int l = grp[g][e] + 32 * h;
                        cols_b |= 1 << ((((Q::tr_base_b(wv, l) ^ Q::tr_xor_b(j)) + j * 2048) >> 4) & 15);
                        cols_a |= 1 << (((Q::tr_base_a(wv, l) ^ Q::tr_xor_a(j)) >> 4) & 15);
                    }
                    bad += cols_b != 0xFFFF;
                    bad += (cols_a != 0xFFFF) << 16; // pass-A side reads (inverse transform): reported in the high half
                }
    // ds_write_b128 of the pass-B side (inverse transform)
    for (int wv = 0; wv < 2; wv++)
        for (int j = 0; j < 8; j++)
            for (int g = 0; g < 8; g++) {
                int cols = 0;
                for (int l = 8 * g; l < 8 * g + 8; l++) {
                    cols |= 1 << ((((Q::tr_base_b(wv, l) ^ Q::tr_xor_b(j)) + j * 2048) >> 4) & 7);
                }
                bad += cols != 0xFF;
            }
    return bad;
}
