// blind_rotate_wave.hip -- variant 1 of the blind rotation: one wavefront per ciphertext
// (lane-level source in wave_pbs.hpp), 4 waves = 4 ciphertexts per 256-thread workgroup,
// one workgroup per CU (it declares ~159 KiB of the CU's 160 KiB LDS).
#include "kernels.hpp"
#include "wave_pbs.hpp"
#include "lane_xchg.hpp"

namespace tfhe_hip {

using namespace wave;

using xchg::wave_sync;

__device__ __forceinline__ void phase_swap(Lane &s)
{
#pragma unroll
    for (int j = 0; j < 16; j++) {
        xchg::swap32(s.R[j].re, s.R[16 + j].re);
        xchg::swap32(s.R[j].im, s.R[16 + j].im);
    }
}

template <int BASE_LOG>
__global__ __launch_bounds__(256, 1) void blind_rotate_wave_kernel(const uint64_t *__restrict__ lwe_small,
                                                                    const uint64_t *__restrict__ luts,
                                                                    const uint32_t *__restrict__ lut_idx,
                                                                    const c64 *__restrict__ bsk_w,
                                                                    const char *__restrict__ gtab,
                                                                    uint64_t *__restrict__ lwe_out, int n, int B,
                                                                    const WaveConsts K, const BrCtl ctl)
{
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int tid = threadIdx.x;
    for (int off = tid * 16; off < TAB_BYTES; off += 256 * 16)
        *reinterpret_cast<c64 *>(lds + TAB_TW + off) = *reinterpret_cast<const c64 *>(gtab + off);
    __syncthreads(); // the only workgroup barrier: from here on every wave is on its own

    const int wave_id = tid >> 6, lane = tid & 63;
    const int ct = blockIdx.x * WAVES_PER_WG + wave_id;
    if (ct >= B) return;
    const int poly_off = WAVE0 + wave_id * WAVE_BYTES + (lane >> 5) * POLY_BYTES;
    const uint64_t *lwe = lwe_small + (size_t)ct * (n + 1);
    const uint64_t *lut = luts + (size_t)checked_lut_idx(lut_idx, ct, ctl.n_luts, ctl.err) * (2 * N);

    Lane s;
    phase_init(s, lut, modswitch(lwe[n], LOG2N), lane);

    for (int i = 0; i < n; i++) {
        const uint32_t a_hat = modswitch(lwe[i], LOG2N);
        const c64 *ggsw = bsk_w + (size_t)i * (4 * FN);
        ggsw_prefetch<0, GRING>(s, ggsw, lane); // 32 x 16 B per lane in flight under the forward transform
        phase_mirror_write(s, lds, poly_off, lane);
        wave_sync();
        phase_rotate_decompose_twist<BASE_LOG>(s, lds, poly_off, lane, a_hat);
        fwd_pass_a(s, lds, lane);
        wave_sync();
        fwd_transpose_write(s, lds, poly_off, lane);
        wave_sync();
        fwd_pass_b(s, lds, poly_off, lane, K);
        phase_swap(s);
        phase_pointwise(s, ggsw, lane);
        phase_swap(s);
        inv_pass_b(s, lds, lane, K);
        wave_sync();
        inv_transpose_write(s, lds, poly_off, lane);
        wave_sync();
        inv_pass_a(s, lds, poly_off, lane);
        wave_sync();
    }
    phase_sample_extract(s, lwe_out + (size_t)ct * (N + 1), lane);
}

// dst[i][e = j*4 + rc][lane] = src[i][rc][bsk_src_pos(lane, j)]   (level count 1)
__global__ void bsk_permute_wave_kernel(const c64 *__restrict__ src, c64 *__restrict__ dst, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int e = (int)((idx >> 6) & 63);
    const size_t i = idx >> 12;
    const int j = e >> 2, rc = e & 3;
    dst[idx] = src[(i * 4 + rc) * FN + bsk_src_pos(lane, j)];
}

hipError_t launch_bsk_permute_wave(const DevParams &p, const double *d_bsk_f, double *d_bsk_w, hipStream_t s)
{
    if (p.pbs_level != 1 || p.k != 1 || p.N != N) return hipErrorInvalidValue;
    const size_t total = (size_t)p.n * 4 * FN;
    bsk_permute_wave_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>((const c64 *)d_bsk_f, (c64 *)d_bsk_w, total);
    return hipGetLastError();
}

bool blind_rotate_wave_supported(const DevParams &p)
{
    return p.k == 1 && p.N == N && p.pbs_level == 1 && p.pbs_base_log == 23;
}

hipError_t launch_blind_rotate_wave(const DevParams &p, const uint64_t *d_lwe_small, const uint64_t *d_luts,
                                    const uint32_t *d_lut_idx, const double *d_bsk_w, const char *d_gtab,
                                    const double *wconsts, uint64_t *d_lwe_out, size_t B, const BrCtl &ctl, hipStream_t s)
{
    if (!blind_rotate_wave_supported(p)) return hipErrorInvalidValue;
    WaveConsts K;
    for (int i = 0; i < 8; i++) K.w[i] = wconsts[i];
    const unsigned grid = (unsigned)((B + WAVES_PER_WG - 1) / WAVES_PER_WG);
    blind_rotate_wave_kernel<23><<<grid, 256, 0, s>>>(d_lwe_small, d_luts, d_lut_idx, (const c64 *)d_bsk_w, d_gtab,
                                                      d_lwe_out, p.n, (int)B, K, ctl);
    return hipGetLastError();
}

} // namespace tfhe_hip
