// keyswitch_mfma.hip -- the LWE keyswitch as an exact int8 matrix product on the matrix cores (gfx950).
//
// Replaces core_crypto/algorithms/lwe_keyswitch.rs:96-170 (keyswitch_lwe_ciphertext) like kernels.hip's dot4 kernel,
// for whole batches:
//
//   out[ct][o] = (0, .., 0, b)[o] - sum_{row} d[ct][row] * ksk[row][o]            row = coefficient * levels + level
//
// is a (ciphertexts x rows) * (rows x columns) product over Z/2^64 with SMALL signed left factors (|d| <= 2^(base_log-1)
// <= 64).  The key is split in its eight byte planes, each byte taken as signed (u - 128), so that
//
//   sum_row d * ksk = sum_{plane b} 2^(8b) * ( sum_row d * (u_b - 128)  +  128 * sum_row d )
//                   = sum_b 2^(8b) * D_b[ct][o]                         +  128 * S[ct] * 0x0101010101010101
//
// where D_b is an int8 x int8 -> int32 matrix product (v_mfma_i32_32x32x32_i8; |D_b| <= 64 * 128 * rows < 2^31, exact) and
// S[ct] the digit sum of the ciphertext.  Everything is integer arithmetic: the result is bit for bit the reference's.
//
// Data layout (both operands are produced by kernels of this file, so they are stored in MFMA fragment order and a
// wavefront's load of a fragment is one contiguous KiB):
//   digits  At[mtile = ct / 32][kstep = row / 32][lane 64][16 B]  lane (r = lane & 31, h = lane >> 5) holds the digits of
//           ciphertext 32 mtile + r for rows 32 kstep + 16 h + j, j < 16                      (per launch, ks_decompose)
//   key     Bt[ntile = o / 32][kstep][plane 8][lane 64][16 B]      lane (r, h) holds bytes (u - 128) of plane `plane` of
//           column 32 ntile + r for the same rows                                              (once, at key upload)
// A wavefront owns 32 ciphertexts x 32 columns x 8 planes (eight 32x32 accumulators, 128 registers): the planes of a
// column meet in ONE lane, so the recombination sum_b 2^(8b) D_b is lane-local.  The four wavefronts of a workgroup
// share the key fragments (same columns, four ciphertext tiles); workgroups that share them run on the same XCD (one
// L2) back to back.  The rows are split over `splits` workgroups whose partial sums the finalize kernel subtracts.
#include "kernels.hpp"

#include "device_math.hpp"

namespace tfhe_hip {

namespace {

using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;

constexpr int MT = 32;  // ciphertexts per MFMA tile
constexpr int NT = 32;  // columns per MFMA tile
constexpr int KSTEP = 32; // rows per MFMA
constexpr int WAVES = 4;  // ciphertext tiles per workgroup

// ---- key -> fragment order (one-off) ------------------------------------------------------------------------------
__global__ void ksk_to_mfma_tiles_kernel(const uint64_t *__restrict__ ksk, uint4 *__restrict__ Bt, int out_size, int ksteps,
                                         size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // ((ntile * ksteps + kstep) * 8 + plane) * 64 + lane
    if (idx >= total) return;
    const int lane = (int)(idx & 63), plane = (int)((idx >> 6) & 7);
    const size_t t = idx >> 9;
    const int kstep = (int)(t % ksteps), ntile = (int)(t / ksteps);
    const int r = lane & 31, h = lane >> 5, o = ntile * NT + r;
    uint32_t w[4] = {0, 0, 0, 0};
    if (o < out_size) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const size_t row = (size_t)kstep * KSTEP + 16 * h + j;
            const uint32_t u = (uint32_t)(ksk[row * out_size + o] >> (8 * plane)) & 0xFFu;
            w[j >> 2] |= (u ^ 0x80u) << (8 * (j & 3)); // u - 128 as a signed byte
        }
    }
    Bt[idx] = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- digits of a batch -> fragment order, digit sums -----------------------------------------------------------------
// grid (in_dim / 32, mtiles): a workgroup decomposes 32 coefficients of 32 ciphertexts = LV k-steps of one tile
template <int LV, int BL>
__global__ __launch_bounds__(256) void ks_decompose_kernel(const uint64_t *__restrict__ lwe_in, uint4 *__restrict__ At,
                                                            int *__restrict__ sums, int B, int in_dim, int ksteps)
{
    __shared__ __attribute__((aligned(16))) uint8_t dig[MT][KSTEP * LV];
    __shared__ int ssum[MT];
    const int tid = threadIdx.x, cb = blockIdx.x, mtile = blockIdx.y;
    constexpr uint64_t mask = ((uint64_t)1 << BL) - 1;
    if (tid < MT) ssum[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int ctl = (tid >> 5) + 8 * i, cf = tid & 31;
        const int ct = mtile * MT + ctl;
        int8_t d[LV];
        int s = 0;
#pragma unroll
        for (int l = 0; l < LV; l++) d[l] = 0; // rows beyond the batch contribute nothing
        if (ct < B) {
            const uint64_t x = lwe_in[(size_t)ct * (in_dim + 1) + cb * 32 + cf];
            uint64_t state = closest_representable(x, BL, LV) >> (64 - BL * LV);
#pragma unroll
            for (int l = 0; l < LV; l++) {
                d[l] = (int8_t)decompose_one_level(BL, state, mask);
                s += d[l];
            }
        }
#pragma unroll
        for (int l = 0; l < LV; l++) dig[ctl][cf * LV + l] = (uint8_t)d[l];
        if (s) atomicAdd(&ssum[ctl], s);
    }
    __syncthreads();
    for (int item = tid; item < 64 * LV; item += 256) {
        const int st = item >> 6, lane = item & 63, r = lane & 31, h = lane >> 5;
        At[((size_t)mtile * ksteps + cb * LV + st) * 64 + lane] = *reinterpret_cast<const uint4 *>(&dig[r][KSTEP * st + 16 * h]);
    }
    if (tid < MT && mtile * MT + tid < B && ssum[tid]) atomicAdd(&sums[mtile * MT + tid], ssum[tid]);
}

// ---- the product -----------------------------------------------------------------------------------------------------
// Workgroup index -> (key slice q = (column tile, row split), ciphertext group g): the groups of one slice sit on
// consecutive slots of ONE XCD (workgroup w runs on XCD w % 8), so a slice of the key crosses the fabric once per XCD.
__global__ __launch_bounds__(64 * WAVES, 2) void keyswitch_mfma_kernel(const uint4 *__restrict__ At, const uint4 *__restrict__ Bt,
                                                                       uint64_t *__restrict__ partial, int B, int out_size,
                                                                       int ksteps, int ks_per_split, int mtiles, int mgroups,
                                                                       int ntiles, int q_count)
{
    const int w = blockIdx.x, xcd = w & 7, slot = w >> 3;
    const int q = (slot / mgroups) * 8 + xcd, g = slot % mgroups;
    if (q >= q_count) return;
    const int ntile = q % ntiles, split = q / ntiles;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int mtile = g * WAVES + wave;
    if (mtile >= mtiles) return; // no workgroup barriers below
    const int ks0 = split * ks_per_split, ks1 = min(ksteps, ks0 + ks_per_split);

    i32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; p++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[p][e] = 0;
    const i32x4 *a_ptr = reinterpret_cast<const i32x4 *>(At) + ((size_t)mtile * ksteps + ks0) * 64 + lane;
    const i32x4 *b_ptr = reinterpret_cast<const i32x4 *>(Bt) + ((size_t)ntile * ksteps + ks0) * 8 * 64 + lane;
    // two k-steps in flight beyond the one being multiplied
    i32x4 a[3], b[3][8];
    const i32x4 zero = {0, 0, 0, 0};
    auto fetch = [&](int slot_, int ks) {
        if (ks < ks1) {
            a[slot_] = a_ptr[(size_t)(ks - ks0) * 64];
#pragma unroll
            for (int p = 0; p < 8; p++) b[slot_][p] = b_ptr[((size_t)(ks - ks0) * 8 + p) * 64];
        } else {
            a[slot_] = zero;
#pragma unroll
            for (int p = 0; p < 8; p++) b[slot_][p] = zero;
        }
    };
    fetch(0, ks0);
    fetch(1, ks0 + 1);
    for (int ks = ks0; ks < ks1; ks += 3) {
#pragma unroll
        for (int u = 0; u < 3; u++) {
            fetch((u + 2) % 3, ks + u + 2);
#pragma unroll
            for (int p = 0; p < 8; p++) acc[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[u], b[u][p], acc[p], 0, 0, 0);
        }
    }
    // C layout (32x32): column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    const int o = ntile * NT + (lane & 31);
    if (o >= out_size) return;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int ct = mtile * MT + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (ct >= B) continue;
        uint64_t v = 0;
#pragma unroll
        for (int p = 0; p < 8; p++) v += (uint64_t)(int64_t)acc[p][e] << (8 * p);
        partial[((size_t)split * B + ct) * out_size + o] = v;
    }
}

__global__ void keyswitch_mfma_finalize_kernel(const uint64_t *__restrict__ lwe_in, const int *__restrict__ sums,
                                               const uint64_t *__restrict__ partial, uint64_t *__restrict__ lwe_out, int B,
                                               int in_dim, int out_size, int splits)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * out_size) return;
    const int ct = (int)(idx / out_size), o = (int)(idx % out_size);
    uint64_t v = (o == out_size - 1) ? lwe_in[(size_t)ct * (in_dim + 1) + in_dim] : 0;
    for (int s = 0; s < splits; s++) v -= partial[((size_t)s * B + ct) * out_size + o];
    v -= (uint64_t)(int64_t)(128 * sums[ct]) * 0x0101010101010101ull;
    lwe_out[idx] = v;
}

} // namespace

bool keyswitch_mfma_supported(const DevParams &p)
{
    const int in_dim = p.k * p.N;
    const bool shape = (p.ks_level == 5 && p.ks_base_log == 3) || (p.ks_level == 4 && p.ks_base_log == 4) ||
                       (p.ks_level == 3 && p.ks_base_log == 5) || (p.ks_level == 2 && p.ks_base_log == 7);
    return shape && in_dim % 32 == 0;
}

size_t keyswitch_mfma_key_bytes(const DevParams &p)
{
    const size_t ntiles = (p.n + 1 + NT - 1) / NT, ksteps = (size_t)p.k * p.N * p.ks_level / KSTEP;
    return ntiles * ksteps * 8 * 64 * 16;
}
size_t keyswitch_mfma_digit_bytes(const DevParams &p, size_t B)
{
    const size_t mtiles = (B + MT - 1) / MT, ksteps = (size_t)p.k * p.N * p.ks_level / KSTEP;
    return mtiles * ksteps * 64 * 16;
}

// row splits: about three workgroups per CU, at least 8 k-steps each
int keyswitch_mfma_splits_for(const DevParams &p, size_t B, int cu_count)
{
    const size_t mtiles = (B + MT - 1) / MT, mgroups = (mtiles + WAVES - 1) / WAVES, ntiles = (p.n + 1 + NT - 1) / NT;
    const int ksteps = p.k * p.N * p.ks_level / KSTEP;
    const size_t want = (size_t)cu_count * 3;
    size_t splits = (want + mgroups * ntiles - 1) / (mgroups * ntiles);
    if (splits < 1) splits = 1;
    if (splits > (size_t)KS_MAX_SPLITS) splits = KS_MAX_SPLITS;
    if ((int)splits > ksteps / 8) splits = ksteps / 8;
    if (splits < 1) splits = 1;
    return (int)splits;
}

hipError_t launch_ksk_to_mfma_tiles(const DevParams &p, const uint64_t *d_ksk, void *d_tiles, hipStream_t s)
{
    if (!keyswitch_mfma_supported(p)) return hipErrorInvalidValue;
    const int out_size = p.n + 1, ksteps = p.k * p.N * p.ks_level / KSTEP;
    const size_t total = keyswitch_mfma_key_bytes(p) / 16;
    ksk_to_mfma_tiles_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(d_ksk, (uint4 *)d_tiles, out_size, ksteps, total);
    return hipGetLastError();
}

// d_digits: keyswitch_mfma_digit_bytes(p, B); d_sums: ceil(B / 32) * 32 ints; d_partial: splits x B x (n + 1) words
hipError_t launch_keyswitch_mfma(const DevParams &p, const uint64_t *d_lwe_in, const void *d_tiles, void *d_digits, int *d_sums,
                                 uint64_t *d_partial, uint64_t *d_lwe_out, size_t B, int splits, hipStream_t s)
{
    if (!keyswitch_mfma_supported(p) || B == 0 || splits < 1) return hipErrorInvalidValue;
    const int in_dim = p.k * p.N, out_size = p.n + 1, ksteps = in_dim * p.ks_level / KSTEP;
    const int mtiles = (int)((B + MT - 1) / MT), mgroups = (mtiles + WAVES - 1) / WAVES, ntiles = (out_size + NT - 1) / NT;
    hipError_t e = hipMemsetAsync(d_sums, 0, (size_t)mtiles * MT * sizeof(int), s);
    if (e != hipSuccess) return e;
    const dim3 dgrid((unsigned)(in_dim / 32), (unsigned)mtiles);
    if (p.ks_level == 5 && p.ks_base_log == 3) // PARAM_MESSAGE_2_CARRY_2_KS_PBS
        ks_decompose_kernel<5, 3><<<dgrid, 256, 0, s>>>(d_lwe_in, (uint4 *)d_digits, d_sums, (int)B, in_dim, ksteps);
    else if (p.ks_level == 4 && p.ks_base_log == 4) // PARAM_MESSAGE_2_CARRY_2_PBS_KS
        ks_decompose_kernel<4, 4><<<dgrid, 256, 0, s>>>(d_lwe_in, (uint4 *)d_digits, d_sums, (int)B, in_dim, ksteps);
    else if (p.ks_level == 3 && p.ks_base_log == 5) // PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS
        ks_decompose_kernel<3, 5><<<dgrid, 256, 0, s>>>(d_lwe_in, (uint4 *)d_digits, d_sums, (int)B, in_dim, ksteps);
    else // PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS
        ks_decompose_kernel<2, 7><<<dgrid, 256, 0, s>>>(d_lwe_in, (uint4 *)d_digits, d_sums, (int)B, in_dim, ksteps);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    const int ks_per_split = (ksteps + splits - 1) / splits;
    const int q_count = ntiles * splits;
    const unsigned grid = (unsigned)((q_count + 7) / 8 * 8 * mgroups);
    keyswitch_mfma_kernel<<<grid, 64 * WAVES, 0, s>>>((const uint4 *)d_digits, (const uint4 *)d_tiles, d_partial, (int)B, out_size,
                                                      ksteps, ks_per_split, mtiles, mgroups, ntiles, q_count);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    const size_t total = B * (size_t)out_size;
    keyswitch_mfma_finalize_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(d_lwe_in, d_sums, d_partial, d_lwe_out, (int)B,
                                                                                  in_dim, out_size, splits);
    return hipGetLastError();
}

} // namespace tfhe_hip
