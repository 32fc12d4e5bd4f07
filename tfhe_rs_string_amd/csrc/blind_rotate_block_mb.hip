// blind_rotate_block_mb.hip -- multi-bit blind rotation (grouping factor 2 or 3), one 512-thread workgroup
// per ciphertext (lane-level source in block_mb_pbs.hpp).  Replaces multi_bit_blind_rotate_assign
// (core_crypto/algorithms/lwe_multi_bit_programmable_bootstrapping.rs:295-545) + sample extraction.
#include "kernels.hpp"
#include "block_mb_pbs.hpp"
#include "lane_xchg.hpp"

namespace tfhe_hip {

using namespace blkmb;
using MbLane = blkmb::Lane;

template <int BASE_LOG, int G>
__global__ __launch_bounds__(THREADS) void blind_rotate_block_mb_kernel(const uint64_t *__restrict__ lwe_small,
                                                                         const uint64_t *__restrict__ luts,
                                                                         const uint32_t *__restrict__ lut_idx,
                                                                         const c64 *__restrict__ bsk_mb,
                                                                         const c64 *__restrict__ twist,
                                                                         const c64 *__restrict__ W,
                                                                         uint64_t *__restrict__ lwe_out, int n, const BrCtl ctl)
{
    __shared__ __attribute__((aligned(256))) char lds[4 * IMG_BYTES]; // forward + inverse images (no mirror)
    const Ids d = make_ids((int)threadIdx.x);
    const int ct = blockIdx.x;
    const uint64_t *lwe = lwe_small + (size_t)ct * (n + 1);
    const uint64_t *lut = luts + (size_t)checked_lut_idx(lut_idx, ct, ctl.n_luts, ctl.err) * (2 * N);
    const int groups = n / G;

    MbLane s;
    phase_init(s, lut, twist, W, modswitch(lwe[n], LOG2N), d);
    // combined GGSW of group 0
    mb_slot<G, 0>(s, bsk_mb, 0, lwe, twist, W, d);
    mb_slot<G, 1>(s, bsk_mb, 0, lwe, twist, W, d);
    mb_slot<G, 2>(s, bsk_mb, 0, lwe, twist, W, d);
    mb_slot<G, 3>(s, bsk_mb, 0, lwe, twist, W, d);
    mb_slot<G, 4>(s, bsk_mb, 0, lwe, twist, W, d);
    mb_slot<G, 5>(s, bsk_mb, 0, lwe, twist, W, d);
    mb_slot<G, 6>(s, bsk_mb, 0, lwe, twist, W, d);
    mb_slot<G, 7>(s, bsk_mb, 0, lwe, twist, W, d);
    for (int grp = 0; grp < groups; grp++) {
        // slots 5-7 of THIS group's combined GGSW ride behind the forward transform (grouping factor 3 only) ...
        phase_decompose_stage1<BASE_LOG>(s, lds, d);
        if (grp > 0) mb_slot<G, 5>(s, bsk_mb, (size_t)grp, lwe, twist, W, d);
        __syncthreads();
        phase_fwd2(s, lds, d);
        xchg::transpose54(s);
        phase_fwd3(s, lds, d);
        if (grp > 0) mb_slot<G, 6>(s, bsk_mb, (size_t)grp, lwe, twist, W, d);
        xchg::wave_sync();
        phase_fwd4(s, lds, d);
        if (grp > 0) mb_slot<G, 7>(s, bsk_mb, (size_t)grp, lwe, twist, W, d);
        __syncthreads();
        phase_pointwise(s, lds, d);
        // ... slots 0-4 of the NEXT group's behind the inverse transform
        const bool more = grp + 1 < groups;
        const size_t nxt = (size_t)grp + 1;
        if (more) mb_slot<G, 0>(s, bsk_mb, nxt, lwe, twist, W, d);
        xchg::wave_sync();
        phase_inv4(s, lds, d);
        if (more) mb_slot<G, 1>(s, bsk_mb, nxt, lwe, twist, W, d);
        xchg::wave_sync();
        phase_inv3(s, lds, d);
        xchg::transpose54(s);
        if (more) mb_slot<G, 2>(s, bsk_mb, nxt, lwe, twist, W, d);
        phase_inv2(s, lds, d);
        if (more) mb_slot<G, 3>(s, bsk_mb, nxt, lwe, twist, W, d);
        __syncthreads();
        phase_inv1_replace(s, lds, d);
        if (more) mb_slot<G, 4>(s, bsk_mb, nxt, lwe, twist, W, d);
    }
    phase_sample_extract_plain(s, lwe_out + (size_t)ct * (N + 1), d, 0);
}

// dst[grp][idx][p][q][e][lane] = src[grp][idx][row = e / 4][col = p][256 q + 4 lane + (e & 3)]   (level count 1)
__global__ void bsk_permute_block_mb_kernel(const c64 *__restrict__ src, c64 *__restrict__ dst, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int e = (int)((idx >> 6) & 7);
    const int q = (int)((idx >> 9) & 3);
    const int p = (int)((idx >> 11) & 1);
    const size_t ggsw = idx >> 12; // group * 2^G + subset index
    dst[idx] = src[(ggsw * 4 + (e >> 2) * 2 + p) * FN + 256 * q + 4 * lane + (e & 3)];
}

bool blind_rotate_multi_bit_supported(const DevParams &p, int grouping_factor)
{
    if (p.k != 1 || p.N != N || p.pbs_level != 1) return false;
    if (grouping_factor == 2) return p.pbs_base_log == 22 && p.n % 2 == 0;
    if (grouping_factor == 3) return p.pbs_base_log == 21 && p.n % 3 == 0;
    return false;
}

hipError_t launch_bsk_permute_block_mb(const DevParams &p, int grouping_factor, const double *d_bsk_f, double *d_bsk_mb,
                                       hipStream_t s)
{
    if (!blind_rotate_multi_bit_supported(p, grouping_factor)) return hipErrorInvalidValue;
    const size_t total = (size_t)(p.n / grouping_factor) * ((size_t)1 << grouping_factor) * 4 * FN;
    bsk_permute_block_mb_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>((const c64 *)d_bsk_f, (c64 *)d_bsk_mb, total);
    return hipGetLastError();
}

hipError_t launch_blind_rotate_block_mb(const DevParams &p, int grouping_factor, const uint64_t *d_lwe_small,
                                        const uint64_t *d_luts, const uint32_t *d_lut_idx, const double *d_bsk_mb,
                                        const double *d_twist, const double *d_w, uint64_t *d_lwe_out, size_t B,
                                        const BrCtl &ctl, hipStream_t s)
{
    if (!blind_rotate_multi_bit_supported(p, grouping_factor) || B == 0) return hipErrorInvalidValue;
    if (grouping_factor == 2)
        blind_rotate_block_mb_kernel<22, 2><<<(unsigned)B, THREADS, 0, s>>>(d_lwe_small, d_luts, d_lut_idx, (const c64 *)d_bsk_mb,
                                                                            (const c64 *)d_twist, (const c64 *)d_w, d_lwe_out, p.n, ctl);
    else
        blind_rotate_block_mb_kernel<21, 3><<<(unsigned)B, THREADS, 0, s>>>(d_lwe_small, d_luts, d_lut_idx, (const c64 *)d_bsk_mb,
                                                                            (const c64 *)d_twist, (const c64 *)d_w, d_lwe_out, p.n, ctl);
    return hipGetLastError();
}

} // namespace tfhe_hip
