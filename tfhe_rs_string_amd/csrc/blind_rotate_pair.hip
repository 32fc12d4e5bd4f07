// blind_rotate_pair.hip -- variant 2 of the blind rotation: two wavefronts per ciphertext, one per
// GLWE polynomial (lane-level source in pair_pbs.hpp).  512-thread workgroups = 4 ciphertexts,
// one workgroup per CU (156 KiB of LDS), two waves per SIMD.
#include "kernels.hpp"
#include "pair_pbs.hpp"
#include "lane_xchg.hpp"

namespace tfhe_hip {

using namespace pair;

namespace {

using xchg::swap16;
using xchg::swap32;
using xchg::wave_sync;

// DIAGNOSTIC BUILDS ONLY (wrong results; tools/build_stamps.sh): what the lane swaps / the pair hand-offs cost
#if defined(TFHE_PAIR_DIAG_NOSWAP)
#define PAIR_SWAPS 0
#else
#define PAIR_SWAPS 1
#endif
#if defined(TFHE_PAIR_DIAG_NOWAIT)
#define PAIR_WAIT(lds, flag, want, alive) (alive)
#else
#define PAIR_WAIT(lds, flag, want, alive) pair_wait(lds, flag, want, alive)
#endif

__device__ __forceinline__ void swap32_pairs(Lane &s)
{
    if (!PAIR_SWAPS) return;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        swap32(s.R[2 * m].re, s.R[2 * m + 1].re);
        swap32(s.R[2 * m].im, s.R[2 * m + 1].im);
    }
}
__device__ __forceinline__ void swap16_pairs(Lane &s)
{
    if (!PAIR_SWAPS) return;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        swap16(s.R[2 * m].re, s.R[2 * m + 1].re);
        swap16(s.R[2 * m].im, s.R[2 * m + 1].im);
    }
}

} // namespace

template <int BASE_LOG>
__global__ __launch_bounds__(512, 2) void blind_rotate_pair_kernel(const uint64_t *__restrict__ lwe_small,
                                                                    const uint64_t *__restrict__ luts,
                                                                    const uint32_t *__restrict__ lut_idx,
                                                                    const c64 *__restrict__ bsk_p,
                                                                    const char *__restrict__ gtab,
                                                                    uint64_t *__restrict__ lwe_out, int n, int B,
                                                                    const WaveConsts K, const BrCtl ctl)
{
    __shared__ __attribute__((aligned(256))) char lds[LDS_BYTES];
    const int tid = threadIdx.x;
    for (int off = tid * 16; off < TAB_BYTES; off += 512 * 16)
        *reinterpret_cast<c64 *>(lds + TAB0 + off) = *reinterpret_cast<const c64 *>(gtab + off);
    if (tid < WAVES_PER_WG) *reinterpret_cast<uint32_t *>(lds + FLAGS + tid * 16) = 0;
    __syncthreads(); // the only workgroup barrier

    const int wave_id = tid >> 6, lane = tid & 63, h = wave_id & 1;
    const int ct_raw = blockIdx.x * 4 + (wave_id >> 1);
    const bool live = ct_raw < B; // waves without a ciphertext still take part in the barriers
    const int ct = live ? ct_raw : B - 1;
    const int poly_off = WAVE0 + wave_id * WAVE_BYTES, partner_off = WAVE0 + (wave_id ^ 1) * WAVE_BYTES;
    const int my_flag = FLAGS + wave_id * 16, partner_flag = FLAGS + (wave_id ^ 1) * 16;
    const uint64_t *lwe = lwe_small + (size_t)ct * (n + 1);
    const uint64_t *lut = luts + (size_t)checked_lut_idx(lut_idx, ct, ctl.n_luts, ctl.err) * (2 * N);
    bool alive = true;                                          // false once a hand-off has timed out (wave-uniform)
    const bool mute = ctl.skip_signal_step >= 0 && ct_raw == 0 && h == 1; // diagnostics: see BrCtl

    Lane s;
    phase_init(s, lut, h, modswitch(lwe[n], LOG2N), lane, gtab);
#ifdef TFHE_PAIR_STAMPS
    // DIAGNOSTIC BUILD ONLY (never shipped, results are overwritten): per-phase cycle sums of one wave
    unsigned long long t_acc[12] = {0}, t_prev = __builtin_amdgcn_s_memtime();
#define STAMP(k)                                                                                     \
    do {                                                                                             \
        __builtin_amdgcn_s_waitcnt(0);                                                               \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();                               \
        t_acc[k] += t_now - t_prev;                                                                  \
        t_prev = t_now;                                                                              \
    } while (0)
#else
#define STAMP(k) ((void)0)
#endif

    for (int i = 0; i < n; i++) {
        const uint32_t a_hat = modswitch(lwe[i], LOG2N);
        const c64 *col = bsk_p + bsk_pair_index((size_t)i, h, 0, 0, 0);
        ggsw_prefetch<0, PGRING>(s, col, lane);
        phase_mirror_write(s, lds, poly_off, lane);
        wave_sync();
        STAMP(0);
        phase_rotate_decompose_twist<BASE_LOG>(s, lds, poly_off, lane, a_hat);
        STAMP(1);
        fwd_stage12(s, lds, lane);
        STAMP(2);
        swap32_pairs(s);
        fwd_stage3_layer_a(s);
        swap16_pairs(s);
        fwd_stage3_layer_b(s, lds, lane);
        wave_sync();
        STAMP(3);
        fwd_transpose_write(s, lds, poly_off, lane);
        wave_sync();
        fwd_pass_b(s, lds, poly_off, lane, K);
        wave_sync();
        STAMP(4);
        exchange_write(s, lds, poly_off, lane);
        wave_sync();
        if (!(mute && i >= ctl.skip_signal_step)) pair_signal(lds, my_flag, 2u * i + 1); // my Fourier polynomial is published
        alive = PAIR_WAIT(lds, partner_flag, 2u * i + 1, alive);                          // ... and so is the partner's
        wave_sync();
        STAMP(5);
        if (h == 0) phase_pointwise<0>(s, lds, partner_off, col, lane);
        else phase_pointwise<1>(s, lds, partner_off, col, lane);
        wave_sync();
        STAMP(6);
        if (!(mute && i >= ctl.skip_signal_step)) pair_signal(lds, my_flag, 2u * i + 2); // I have read the partner's buffer
        inv_pass_b(s, K);
        STAMP(7);
        alive = PAIR_WAIT(lds, partner_flag, 2u * i + 2, alive); // the partner has read mine: it may be overwritten
        wave_sync();
        STAMP(8);
        inv_transpose_write(s, lds, poly_off, lane);
        wave_sync();
        inv_stage3_read_layer_a(s, lds, poly_off, lane);
        swap16_pairs(s);
        inv_stage3_layer_b(s, lane);
        swap32_pairs(s);
        STAMP(9);
        inv_stage21_accumulate(s, lds, lane);
        wave_sync();
        STAMP(10);
    }
    if (!alive && lane == 0) report_error(ctl.err, ERR_PAIR_TIMEOUT); // the output of this ciphertext is NOT valid
#ifdef TFHE_PAIR_STAMPS
    // DIAGNOSTIC BUILD ONLY: workgroup 0 hands out its clock sums instead of its (then meaningless) results
    if (blockIdx.x == 0) {
        if (wave_id < 2 && lane == 0)
            for (int k = 0; k < 12; k++) lwe_out[64 + wave_id * 16 + k] = t_acc[k];
        return;
    }
#endif
    if (live) phase_sample_extract(s, lwe_out + (size_t)ct * (N + 1), h, lane);
}

// dst[i][h][j][row][lane] = BSK_PRESCALE * src[i][row][col = h][16 lane + j]   (level count 1)
__global__ void bsk_permute_pair_kernel(const c64 *__restrict__ src, c64 *__restrict__ dst, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int row = (int)((idx >> 6) & 1);
    const int j = (int)((idx >> 7) & 15);
    const int h = (int)((idx >> 11) & 1);
    const size_t i = idx >> 12;
    const c64 g = src[(i * 4 + row * 2 + h) * FN + 16 * lane + j];
    c64 o;
    o.re = g.re * BSK_PRESCALE; // exact (power of two): see pair_pbs.hpp
    o.im = g.im * BSK_PRESCALE;
    dst[idx] = o;
}

// the inverse: the oracle-order Fourier key back from the pair-order copy (both scalings are exact powers of two)
__global__ void bsk_unpermute_pair_kernel(const c64 *__restrict__ src, c64 *__restrict__ dst, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int row = (int)((idx >> 6) & 1);
    const int j = (int)((idx >> 7) & 15);
    const int h = (int)((idx >> 11) & 1);
    const size_t i = idx >> 12;
    const c64 g = src[idx];
    c64 o;
    o.re = g.re * (1.0 / BSK_PRESCALE);
    o.im = g.im * (1.0 / BSK_PRESCALE);
    dst[(i * 4 + row * 2 + h) * FN + 16 * lane + j] = o;
}

hipError_t launch_bsk_unpermute_pair(const DevParams &p, const double *d_bsk_p, double *d_bsk_f, hipStream_t s)
{
    if (p.pbs_level != 1 || p.k != 1 || p.N != N) return hipErrorInvalidValue;
    const size_t total = (size_t)p.n * 4 * FN;
    bsk_unpermute_pair_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>((const c64 *)d_bsk_p, (c64 *)d_bsk_f, total);
    return hipGetLastError();
}

hipError_t launch_bsk_permute_pair(const DevParams &p, const double *d_bsk_f, double *d_bsk_p, hipStream_t s)
{
    if (p.pbs_level != 1 || p.k != 1 || p.N != N) return hipErrorInvalidValue;
    const size_t total = (size_t)p.n * 4 * FN;
    bsk_permute_pair_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>((const c64 *)d_bsk_f, (c64 *)d_bsk_p, total);
    return hipGetLastError();
}

hipError_t launch_blind_rotate_pair(const DevParams &p, const uint64_t *d_lwe_small, const uint64_t *d_luts,
                                    const uint32_t *d_lut_idx, const double *d_bsk_p, const char *d_gtab,
                                    const double *wconsts, uint64_t *d_lwe_out, size_t B, const BrCtl &ctl, hipStream_t s)
{
    if (!blind_rotate_wave_supported(p) || B == 0) return hipErrorInvalidValue;
    WaveConsts K;
    for (int i = 0; i < 8; i++) K.w[i] = wconsts[i];
    const unsigned grid = (unsigned)((B + 3) / 4);
    blind_rotate_pair_kernel<23><<<grid, 512, 0, s>>>(d_lwe_small, d_luts, d_lut_idx, (const c64 *)d_bsk_p, d_gtab,
                                                      d_lwe_out, p.n, (int)B, K, ctl);
    return hipGetLastError();
}

} // namespace tfhe_hip
