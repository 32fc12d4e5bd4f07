// blind_rotate_quad.hip -- variant 5 of the blind rotation: four wavefronts per ciphertext (lane-level source in
// quad_pbs.hpp).  256-thread workgroups, 68 KiB of LDS, two workgroups per CU: a level of up to 512 bootstraps runs
// in one round with every ciphertext's CMUX step on all four SIMDs of its CU.
#include "kernels.hpp"
#include "quad_pbs.hpp"
#include "lane_xchg.hpp"

namespace tfhe_hip {

using namespace quad;

namespace {

using xchg::swap16;
using xchg::swap32;

// DIAGNOSTIC BUILDS ONLY (wrong results; tools/build_stamps.sh): what the lane swaps / the workgroup barriers cost
#if defined(TFHE_QUAD_DIAG_NOSWAP)
#define QUAD_SWAPS 0
#else
#define QUAD_SWAPS 1
#endif
#if defined(TFHE_QUAD_DIAG_NOBARRIER)
#define QUAD_BARRIER() xchg::wave_sync()
#else
#define QUAD_BARRIER() __syncthreads()
#endif

__device__ __forceinline__ void swap32_regs(Lane &s) // (R[2k], R[2k+1])
{
    if (!QUAD_SWAPS) return;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        swap32(s.R[2 * k].re, s.R[2 * k + 1].re);
        swap32(s.R[2 * k].im, s.R[2 * k + 1].im);
    }
}
__device__ __forceinline__ void swap16_regs(Lane &s) // (R[x], R[x+2]), x = 0, 1, 4, 5
{
    if (!QUAD_SWAPS) return;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = (k & 1) + 4 * (k >> 1);
        swap16(s.R[x].re, s.R[x + 2].re);
        swap16(s.R[x].im, s.R[x + 2].im);
    }
}

} // namespace

template <int BASE_LOG>
__global__ __launch_bounds__(THREADS, 2) void blind_rotate_quad_kernel(const uint64_t *__restrict__ lwe_small,
                                                                        const uint64_t *__restrict__ luts,
                                                                        const uint32_t *__restrict__ lut_idx,
                                                                        const c64 *__restrict__ bsk_q,
                                                                        const c64 *__restrict__ twist,
                                                                        const c64 *__restrict__ W,
                                                                        uint64_t *__restrict__ lwe_out, int n, const BrCtl ctl)
{
    __shared__ __attribute__((aligned(256))) char lds[LDS_BYTES];
    const int tid = (int)threadIdx.x;
    Ids d = make_ids(tid);
    d.p = __builtin_amdgcn_readfirstlane(d.p); // wave-uniform
    d.w = __builtin_amdgcn_readfirstlane(d.w);
    const int ct = blockIdx.x;
    const uint64_t *lwe = lwe_small + (size_t)ct * (n + 1);
    const uint64_t *lut = luts + (size_t)checked_lut_idx(lut_idx, ct, ctl.n_luts, ctl.err) * (2 * N);

    Lane s;
    phase_init(s, lds, lut, twist, W, modswitch(lwe[n], LOG2N), d, tid);
#ifdef TFHE_QUAD_STAMPS
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
        const c64 *col = bsk_q + bsk_quad_index((size_t)i, d.p, d.w, 0, 0, 0);
        ggsw_prefetch<0, QGHEAD>(s, col, d.l);
        phase_mirror_write(s, lds, d);
        STAMP(0);
        QUAD_BARRIER(); // the mirror of the accumulator is complete (and the stage-3 table, the first time)
        STAMP(1);
        phase_rotate_decompose_twist<BASE_LOG>(s, lds, d, a_hat);
        STAMP(2);
        const Tw3 w2 = load_w2(lds, d); // issued ahead of stage 1
        fwd_stage1(s);
        swap32_regs(s);
        fwd_stage2(s, w2);
        swap16_regs(s);
        fwd_stage3_layer_a(s);
        transpose_write_a(s, lds, d);
        const TwB tb = load_twb(lds, d); // (tables: not behind the barrier)
        STAMP(3);
        QUAD_BARRIER(); // blocks of 32 positions change thread
        STAMP(4);
        transpose_read_b(s, lds, d);
        fwd_pass_b1(s, tb, d);
        ggsw_prefetch<QGHEAD, 8 - QGHEAD>(s, col, d.l);
        swap32_regs(s);
        swap16_regs(s);
        fwd_stage5(s);
        exchange_write(s, lds, d);
        STAMP(5);
        QUAD_BARRIER(); // both Fourier polynomials complete
        STAMP(6);
        // (the two bodies differ only in which operand is the thread's own polynomial; the markers keep the compiler from
        // merging them into one body behind 32 register moves on one side)
        if (d.p == 0) {
            phase_pointwise<0>(s, lds, d);
            TFHE_BRANCH_MARK("external product, column 0");
        } else {
            phase_pointwise<1>(s, lds, d);
            TFHE_BRANCH_MARK("external product, column 1");
        }
        const TwB tbi = load_twb(lds, d); // under stage 5' and the swaps
        inv_stage5(s);
        swap16_regs(s);
        swap32_regs(s);
        inv_pass_b1(s, tbi);
        transpose_write_b(s, lds, d);
        const Tw3 w2i = load_w2(lds, d);
        STAMP(7);
        QUAD_BARRIER();
        STAMP(8);
        transpose_read_a(s, lds, d);
        inv_stage3_layer_b(s, d);
        swap16_regs(s);
        inv_stage2(s, w2i);
        swap32_regs(s);
        inv_stage1_accumulate(s);
        STAMP(9);
    }
#ifdef TFHE_QUAD_STAMPS
    if (blockIdx.x == 0) {
        if (tid == 0)
            for (int k = 0; k < 12; k++) lwe_out[64 + k] = t_acc[k];
        return;
    }
#endif
    phase_sample_extract(s, lwe_out + (size_t)ct * (N + 1), d);
}

// dst[i][p][w][reg][row][lane] = BSK_PRESCALE * src[i][row][col = p][fourier_pos(w, lane, reg)]   (level count 1)
__global__ void bsk_permute_quad_kernel(const c64 *__restrict__ src, c64 *__restrict__ dst, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int row = (int)((idx >> 6) & 1);
    const int reg = (int)((idx >> 7) & 7);
    const int w = (int)((idx >> 10) & 1);
    const int p = (int)((idx >> 11) & 1);
    const size_t i = idx >> 12;
    const c64 g = src[(i * 4 + row * 2 + p) * FN + fourier_pos(w, lane, reg)];
    c64 o;
    o.re = g.re * BSK_PRESCALE; // exact (power of two): pair_pbs.hpp
    o.im = g.im * BSK_PRESCALE;
    dst[idx] = o;
}

hipError_t launch_bsk_permute_quad(const DevParams &p, const double *d_bsk_f, double *d_bsk_q, hipStream_t s)
{
    if (!blind_rotate_wave_supported(p)) return hipErrorInvalidValue;
    const size_t total = (size_t)p.n * 4 * FN;
    bsk_permute_quad_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>((const c64 *)d_bsk_f, (c64 *)d_bsk_q, total);
    return hipGetLastError();
}

hipError_t launch_blind_rotate_quad(const DevParams &p, const uint64_t *d_lwe_small, const uint64_t *d_luts,
                                    const uint32_t *d_lut_idx, const double *d_bsk_q, const double *d_twist,
                                    const double *d_w, uint64_t *d_lwe_out, size_t B, const BrCtl &ctl, hipStream_t s)
{
    if (!blind_rotate_wave_supported(p) || B == 0) return hipErrorInvalidValue;
    blind_rotate_quad_kernel<23><<<(unsigned)B, THREADS, 0, s>>>(d_lwe_small, d_luts, d_lut_idx, (const c64 *)d_bsk_q,
                                                                 (const c64 *)d_twist, (const c64 *)d_w, d_lwe_out, p.n, ctl);
    return hipGetLastError();
}

} // namespace tfhe_hip
