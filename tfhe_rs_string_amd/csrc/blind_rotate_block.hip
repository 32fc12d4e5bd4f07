// blind_rotate_block.hip -- variant 3 of the blind rotation: one 512-thread workgroup per ciphertext
// (lane-level source in block_pbs.hpp).  The latency kernel for batches that do not fill the chip:
// a CMUX step of one ciphertext is spread over the 8 wavefronts (all 4 SIMDs) of a CU.
// 64 KiB of LDS and <= 128 registers would allow two workgroups per CU; the kernel is built for one
// (<= 256 registers, twiddles and the bootstrap-key prefetch in registers).
#include "kernels.hpp"
#include "block_pbs.hpp"
#include "lane_xchg.hpp"

namespace tfhe_hip {

using namespace blk;

using xchg::transpose54;
using xchg::wave_sync;

template <int BASE_LOG>
__global__ __launch_bounds__(THREADS) void blind_rotate_block_kernel(const uint64_t *__restrict__ lwe_small,
                                                                      const uint64_t *__restrict__ luts,
                                                                      const uint32_t *__restrict__ lut_idx,
                                                                      const c64 *__restrict__ bsk_b,
                                                                      const c64 *__restrict__ twist,
                                                                      const c64 *__restrict__ W,
                                                                      uint64_t *__restrict__ lwe_out, int n, const BrCtl ctl)
{
    __shared__ __attribute__((aligned(256))) char lds[LDS_BYTES];
    const Ids d = make_ids((int)threadIdx.x);
    const int ct = blockIdx.x;
    const uint64_t *lwe = lwe_small + (size_t)ct * (n + 1);
    const uint64_t *lut = luts + (size_t)checked_lut_idx(lut_idx, ct, ctl.n_luts, ctl.err) * (2 * N);

    Lane s;
    phase_init(s, lds, lut, twist, W, modswitch(lwe[n], LOG2N), d);
    ggsw_prefetch(s, bsk_b, 0, d);
    __syncthreads();

#ifdef TFHE_BLOCK_STAMPS
    // DIAGNOSTIC BUILD ONLY (never shipped, the output is overwritten): per-phase clock sums of wavefront 0
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
        phase_rotate_stage1<BASE_LOG>(s, lds, a_hat, d);
        STAMP(0);
        __syncthreads(); // blocks of 256 positions change wavefront
        STAMP(1);
        phase_fwd2(s, lds, d);
        transpose54(s);
        phase_fwd3(s, lds, d);
        wave_sync();
        STAMP(2);
        phase_fwd4(s, lds, d);
        STAMP(3);
        __syncthreads(); // both Fourier polynomials complete
        STAMP(4);
        phase_pointwise(s, lds, d);
        if (i + 1 < n) ggsw_prefetch(s, bsk_b, (size_t)i + 1, d); // a whole CMUX step ahead of its use
        wave_sync();
#ifdef TFHE_BLOCK_STAMPS
        { // without draining the prefetch
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();
            t_acc[5] += t_now - t_prev;
            t_prev = t_now;
        }
#endif
        phase_inv4(s, lds, d);
        wave_sync();
        phase_inv3(s, lds, d);
        transpose54(s);
        phase_inv2(s, lds, d);
#ifdef TFHE_BLOCK_STAMPS
        {
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();
            t_acc[6] += t_now - t_prev;
            t_prev = t_now;
        }
#endif
        __syncthreads();
#ifdef TFHE_BLOCK_STAMPS
        {
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();
            t_acc[7] += t_now - t_prev;
            t_prev = t_now;
        }
#endif
        phase_inv1_accumulate(s, lds, d);
#ifdef TFHE_BLOCK_STAMPS
        {
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();
            t_acc[8] += t_now - t_prev;
            t_prev = t_now;
        }
#endif
        __syncthreads(); // mirror of the new accumulator complete
#ifdef TFHE_BLOCK_STAMPS
        {
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();
            t_acc[9] += t_now - t_prev;
            t_prev = t_now;
        }
#endif
    }
    phase_sample_extract(s, lwe_out + (size_t)ct * (N + 1), d);
#ifdef TFHE_BLOCK_STAMPS
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 0; k < 12; k++) lwe_out[(size_t)ct * (N + 1) + 64 + k] = t_acc[k];
#endif
}

// dst[i][p][q][e = row * 4 + t][lane] = BSK_PRESCALE * src[i][row][col = p][256 q + 4 lane + t]   (level count 1)
__global__ void bsk_permute_block_kernel(const c64 *__restrict__ src, c64 *__restrict__ dst, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int e = (int)((idx >> 6) & 7);
    const int q = (int)((idx >> 9) & 3);
    const int p = (int)((idx >> 11) & 1);
    const size_t i = idx >> 12;
    const c64 g = src[(i * 4 + (e >> 2) * 2 + p) * FN + 256 * q + 4 * lane + (e & 3)];
    c64 o;
    o.re = g.re * BSK_PRESCALE; // exact (power of two): block_pbs.hpp
    o.im = g.im * BSK_PRESCALE;
    dst[idx] = o;
}

hipError_t launch_bsk_permute_block(const DevParams &p, const double *d_bsk_f, double *d_bsk_b, hipStream_t s)
{
    if (!blind_rotate_wave_supported(p)) return hipErrorInvalidValue;
    const size_t total = (size_t)p.n * 4 * FN;
    bsk_permute_block_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>((const c64 *)d_bsk_f, (c64 *)d_bsk_b, total);
    return hipGetLastError();
}

hipError_t launch_blind_rotate_block(const DevParams &p, const uint64_t *d_lwe_small, const uint64_t *d_luts,
                                     const uint32_t *d_lut_idx, const double *d_bsk_b, const double *d_twist,
                                     const double *d_w, uint64_t *d_lwe_out, size_t B, const BrCtl &ctl, hipStream_t s)
{
    if (!blind_rotate_wave_supported(p) || B == 0) return hipErrorInvalidValue;
    blind_rotate_block_kernel<23><<<(unsigned)B, THREADS, 0, s>>>(d_lwe_small, d_luts, d_lut_idx, (const c64 *)d_bsk_b,
                                                                   (const c64 *)d_twist, (const c64 *)d_w, d_lwe_out,
                                                                   p.n, ctl);
    return hipGetLastError();
}

} // namespace tfhe_hip
