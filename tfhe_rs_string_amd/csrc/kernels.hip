// kernels.hip -- HIP kernels of the KS-PBS engine for gfx950 (MI355X).  wave = 64 lanes.
//
// Kernels (reference function each one replaces, paths relative to /root/reference/tfhe/src):
//   ksk_colsum_kernel, keyswitch_kernel, keyswitch_finalize_kernel
//        core_crypto/algorithms/lwe_keyswitch.rs:96-170 (keyswitch_lwe_ciphertext)
//   bsk_to_fourier_kernel
//        core_crypto/algorithms/lwe_bootstrap_key_conversion.rs:99-152 -> fft64/math/fft/mod.rs:378-386
//   blind_rotate_v0_kernel
//        fft64/crypto/bootstrap.rs:242-364 (blind_rotate_assign + bootstrap), ggsw.rs:477-598
//        (add_external_product_assign), glwe_sample_extraction.rs:91-147
#include "kernels.hpp"

#include <algorithm>

#include "device_math.hpp"

namespace tfhe_hip {

// =============================================================================================
// Keyswitch
// =============================================================================================
//
// out = (0,...,0,b) - sum_{j<kN} sum_{l} d_{j,l} * ksk[j][l][:]   (u64 wrapping)
// with d in [-B/2, B/2].  The signed digit is biased to d' = d + B/2 >= 0 so the inner loop is an
// unsigned multiply-add; the bias is removed with the precomputed column sum
//   colsum[o] = B/2 * sum_{j,l} ksk[j][l][o]   =>   out = init + colsum - sum d' * ksk
// (exact: everything is arithmetic in the ring Z/2^64).
//
// Work split: workgroup = (tile of KS_BT ciphertexts) x (256 output columns) x (one j-split).
// Thread = one output column, KS_BT accumulators in registers; every KSK element loaded
// (coalesced, 8 B/lane) is reused for the KS_BT ciphertexts of the tile; the digits of the tile
// live in LDS, packed (base_log+1) bits per level, and are read as wave-wide broadcasts.

constexpr int KS_JC = 256; // input coefficients per LDS digit chunk

// The multiply-add d' * ksk (d' < 2^8, ksk 64 bits) is done byte-plane by byte-plane with the packed
// 8-bit dot product v_dot4_u32_u8 (full-rate VALU; the 64-bit v_mad_u64_u32 is quarter rate):
// the key is re-laid out once as planes[g][o][b] = bytes b of the four consecutive rows 4g..4g+3
// (row = j * levels + level) of column o packed in one dword, so one dot4 accumulates four rows of
// one byte plane.  A plane sum stays below 10240 * 2^(BL+1) * 255 < 2^32; the eight plane sums are
// recombined with their weights 2^(8b) at the end (exact in Z/2^64).
template <int LV, int BL>
__global__ __launch_bounds__(KS_COLS) void keyswitch_kernel(const uint64_t *__restrict__ lwe_in,
                                                            const uint32_t *__restrict__ planes,
                                                            uint64_t *__restrict__ partial, int B, int in_dim,
                                                            int out_size, int j_per_split)
{
    static_assert((KS_JC * LV) % 4 == 0, "a digit chunk must hold whole row groups");
    constexpr int ROWS = KS_JC * LV; // rows per chunk
    __shared__ __attribute__((aligned(16))) uint8_t dig[KS_BT][ROWS];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, split = blockIdx.z;
    const int o = blockIdx.y * KS_COLS + tid;
    const bool active = o < out_size;
    const int j0 = split * j_per_split;
    const int j1 = min(in_dim, j0 + j_per_split);
    constexpr uint64_t mask = ((uint64_t)1 << BL) - 1;
    constexpr int32_t half = 1 << (BL - 1);

    uint32_t p[KS_BT][8];
#pragma unroll
    for (int c = 0; c < KS_BT; c++)
#pragma unroll
        for (int b = 0; b < 8; b++) p[c][b] = 0;

    for (int jc = j0; jc < j1; jc += KS_JC) {
        __syncthreads();
        for (int e = tid; e < KS_BT * KS_JC; e += KS_COLS) {
            const int c = e / KS_JC, jj = e % KS_JC;
            const int ct = tile * KS_BT + c, j = jc + jj;
            uint8_t d8[LV];
#pragma unroll
            for (int l = 0; l < LV; l++) d8[l] = 0; // rows beyond the batch / split contribute nothing
            if (ct < B && j < j1) {
                const uint64_t x = lwe_in[(size_t)ct * (in_dim + 1) + j];
                uint64_t state = closest_representable(x, BL, LV) >> (64 - BL * LV);
#pragma unroll
                for (int l = 0; l < LV; l++) d8[l] = (uint8_t)((int32_t)decompose_one_level(BL, state, mask) + half);
            }
#pragma unroll
            for (int l = 0; l < LV; l++) dig[c][jj * LV + l] = d8[l];
        }
        __syncthreads();
        if (active) {
            const int jn = min(KS_JC, j1 - jc);
            const int groups = jn * LV / 4; // j ranges are multiples of 4
            const uint4 *kp = reinterpret_cast<const uint4 *>(planes + ((size_t)(jc * LV / 4) * out_size + o) * 8);
            for (int g = 0; g < groups; g++) {
                const uint4 k0 = kp[0], k1 = kp[1]; // byte planes 0..3, 4..7 of rows 4g..4g+3
                kp += (size_t)out_size * 2;
#pragma unroll
                for (int c = 0; c < KS_BT; c++) {
                    const uint32_t d4 = *reinterpret_cast<const uint32_t *>(&dig[c][4 * g]);
                    p[c][0] = __builtin_amdgcn_udot4(d4, k0.x, p[c][0], false);
                    p[c][1] = __builtin_amdgcn_udot4(d4, k0.y, p[c][1], false);
                    p[c][2] = __builtin_amdgcn_udot4(d4, k0.z, p[c][2], false);
                    p[c][3] = __builtin_amdgcn_udot4(d4, k0.w, p[c][3], false);
                    p[c][4] = __builtin_amdgcn_udot4(d4, k1.x, p[c][4], false);
                    p[c][5] = __builtin_amdgcn_udot4(d4, k1.y, p[c][5], false);
                    p[c][6] = __builtin_amdgcn_udot4(d4, k1.z, p[c][6], false);
                    p[c][7] = __builtin_amdgcn_udot4(d4, k1.w, p[c][7], false);
                }
            }
        }
    }
    if (active) {
#pragma unroll
        for (int c = 0; c < KS_BT; c++) {
            const int ct = tile * KS_BT + c;
            if (ct >= B) continue;
            uint64_t v = 0;
#pragma unroll
            for (int b = 0; b < 8; b++) v += (uint64_t)p[c][b] << (8 * b);
            partial[((size_t)split * B + ct) * out_size + o] = v;
        }
    }
}

// planes[g][o][b] (dword) = bytes b of ksk rows 4g .. 4g+3, column o   (one-off, at key upload)
__global__ void ksk_to_planes_kernel(const uint64_t *__restrict__ ksk, uint32_t *__restrict__ planes, int groups,
                                     int out_size)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)groups * out_size) return;
    const size_t g = idx / out_size;
    const int o = (int)(idx % out_size);
    uint64_t k[4];
#pragma unroll
    for (int r = 0; r < 4; r++) k[r] = ksk[(g * 4 + r) * out_size + o];
#pragma unroll
    for (int b = 0; b < 8; b++) {
        uint32_t w = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) w |= (uint32_t)((k[r] >> (8 * b)) & 0xFF) << (8 * r);
        planes[idx * 8 + b] = w;
    }
}

// grid (column tiles, row slices): each thread sums one column over its slice of rows and adds it with one atomic
// (u64 wrapping addition is associative and commutative: the result does not depend on the order)
__global__ void ksk_colsum_kernel(const uint64_t *__restrict__ ksk, unsigned long long *__restrict__ colsum, int rows,
                                  int out_size, uint64_t half)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= out_size) return;
    const int per = (rows + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    uint64_t s = 0;
    for (int r = r0; r < r1; r++) s += ksk[(size_t)r * out_size + o];
    atomicAdd(&colsum[o], (unsigned long long)(s * half));
}

__global__ void keyswitch_finalize_kernel(const uint64_t *__restrict__ lwe_in, const uint64_t *__restrict__ colsum,
                                          const uint64_t *__restrict__ partial, uint64_t *__restrict__ lwe_out, int B,
                                          int in_dim, int out_size, int splits)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * out_size) return;
    const int ct = (int)(idx / out_size), o = (int)(idx % out_size);
    uint64_t v = colsum[o];
    if (o == out_size - 1) v += lwe_in[(size_t)ct * (in_dim + 1) + in_dim];
    for (int s = 0; s < splits; s++) v -= partial[((size_t)s * B + ct) * out_size + o];
    lwe_out[idx] = v;
}

int ks_splits_for(const DevParams &p, size_t B, int cu_count)
{
    const int in_dim = p.k * p.N;
    const size_t tiles = (B + KS_BT - 1) / KS_BT;
    const size_t colblks = (p.n + 1 + KS_COLS - 1) / KS_COLS;
    const size_t want = (size_t)cu_count * 3; // >= 3 workgroups per CU
    size_t splits = (want + tiles * colblks - 1) / (tiles * colblks);
    if (splits < 1) splits = 1;
    if (splits > KS_MAX_SPLITS) splits = KS_MAX_SPLITS;
    // j_per_split must be a multiple of KS_JC
    const int max_splits = in_dim / KS_JC;
    if ((int)splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    return (int)splits;
}

hipError_t launch_ksk_colsum(const DevParams &p, const uint64_t *d_ksk, uint64_t *d_colsum, hipStream_t s)
{
    const int out_size = p.n + 1, rows = p.k * p.N * p.ks_level;
    hipError_t e = hipMemsetAsync(d_colsum, 0, (size_t)out_size * 8, s);
    if (e != hipSuccess) return e;
    const dim3 grid((out_size + 255) / 256, (unsigned)std::min(rows, 256));
    ksk_colsum_kernel<<<grid, 256, 0, s>>>(d_ksk, (unsigned long long *)d_colsum, rows, out_size, (uint64_t)1 << (p.ks_base_log - 1));
    return hipGetLastError();
}

hipError_t launch_ksk_to_planes(const DevParams &p, const uint64_t *d_ksk, uint32_t *d_planes, hipStream_t s)
{
    const int out_size = p.n + 1, rows = p.k * p.N * p.ks_level;
    if (rows % 4) return hipErrorInvalidValue;
    const size_t total = (size_t)(rows / 4) * out_size;
    ksk_to_planes_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(d_ksk, d_planes, rows / 4, out_size);
    return hipGetLastError();
}

hipError_t launch_keyswitch(const DevParams &p, const uint64_t *d_lwe_in, const uint32_t *d_ksk, uint64_t *d_partial,
                            size_t B, int splits, hipStream_t s)
{
    const int in_dim = p.k * p.N, out_size = p.n + 1;
    const int per = ((in_dim + splits - 1) / splits + KS_JC - 1) / KS_JC * KS_JC;
    dim3 grid((unsigned)((B + KS_BT - 1) / KS_BT), (unsigned)((out_size + KS_COLS - 1) / KS_COLS), (unsigned)splits);
    if (p.ks_level == 5 && p.ks_base_log == 3) // PARAM_MESSAGE_2_CARRY_2_KS_PBS
        keyswitch_kernel<5, 3><<<grid, KS_COLS, 0, s>>>(d_lwe_in, d_ksk, d_partial, (int)B, in_dim, out_size, per);
    else if (p.ks_level == 4 && p.ks_base_log == 4) // PARAM_MESSAGE_2_CARRY_2_PBS_KS
        keyswitch_kernel<4, 4><<<grid, KS_COLS, 0, s>>>(d_lwe_in, d_ksk, d_partial, (int)B, in_dim, out_size, per);
    else if (p.ks_level == 3 && p.ks_base_log == 5) // PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS
        keyswitch_kernel<3, 5><<<grid, KS_COLS, 0, s>>>(d_lwe_in, d_ksk, d_partial, (int)B, in_dim, out_size, per);
    else if (p.ks_level == 2 && p.ks_base_log == 7) // PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS
        keyswitch_kernel<2, 7><<<grid, KS_COLS, 0, s>>>(d_lwe_in, d_ksk, d_partial, (int)B, in_dim, out_size, per);
    else if (p.ks_level == 2 && p.ks_base_log == 5) // PARAM_MESSAGE_1_CARRY_0_KS_PBS
        keyswitch_kernel<2, 5><<<grid, KS_COLS, 0, s>>>(d_lwe_in, d_ksk, d_partial, (int)B, in_dim, out_size, per);
    else if (p.ks_level == 3 && p.ks_base_log == 4) // PARAM_MESSAGE_{1_1, 1_2, 2_1, 3_0}_KS_PBS
        keyswitch_kernel<3, 4><<<grid, KS_COLS, 0, s>>>(d_lwe_in, d_ksk, d_partial, (int)B, in_dim, out_size, per);
    else if (p.ks_level == 4 && p.ks_base_log == 3) // PARAM_MESSAGE_2_CARRY_0_KS_PBS
        keyswitch_kernel<4, 3><<<grid, KS_COLS, 0, s>>>(d_lwe_in, d_ksk, d_partial, (int)B, in_dim, out_size, per);
    else if (p.ks_level == 6 && p.ks_base_log == 3) // PARAM_MESSAGE_2_CARRY_3_KS_PBS
        keyswitch_kernel<6, 3><<<grid, KS_COLS, 0, s>>>(d_lwe_in, d_ksk, d_partial, (int)B, in_dim, out_size, per);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_keyswitch_finalize(const DevParams &p, const uint64_t *d_lwe_in, const uint64_t *d_colsum,
                                     const uint64_t *d_partial, uint64_t *d_lwe_out, size_t B, int splits,
                                     hipStream_t s)
{
    const int out_size = p.n + 1;
    const size_t total = B * (size_t)out_size;
    keyswitch_finalize_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(d_lwe_in, d_colsum, d_partial,
                                                                              d_lwe_out, (int)B, p.k * p.N, out_size,
                                                                              splits);
    return hipGetLastError();
}

// =============================================================================================
// LDS FFT (variant 0 and the key conversion): n = N/2 complex points in LDS, ANY power of two >= 4, 256 threads, the
// butterflies of a stage dealt out over the threads, a barrier per stage.  Schedule of DESIGN.md section 4: radix-4
// DIF / DIT stages, behind ONE radix-2 stage when n = 2 * 4^k (the oracle's fft_fwd / fft_inv)
// =============================================================================================

// forward: caller syncs before (data written) -- a barrier follows every stage, so the result is visible to all
// threads on return.  Output in DIF position order.  W: n twiddles exp(-2 pi i k / n).
__device__ __forceinline__ void lds_fft_fwd(c64 *buf, const c64 *__restrict__ W, int n, int tid)
{
    int lg = 0;
    while ((1 << lg) < n) lg++;
    int L = n;
    if (lg & 1) { // radix-2 DIF stage: x[j] = a + b ; x[j + n/2] = (a - b) W[j]
        const int h = n / 2;
        for (int j = tid; j < h; j += 256) {
            const c64 a = buf[j], b = buf[j + h];
            c64 sm, df;
            sm.re = a.re + b.re; sm.im = a.im + b.im;
            df.re = a.re - b.re; df.im = a.im - b.im;
            buf[j] = sm;
            buf[j + h] = j ? cmul(df, W[j]) : df;
        }
        __syncthreads();
        L = h;
    }
    for (; L >= 4; L >>= 2) {
        const int Q = L / 4, step = n / L;
        for (int t = tid; t < n / 4; t += 256) {
            const int jj = t & (Q - 1);
            const int i0 = (t / Q) * L + jj;
            c64 a = buf[i0], b = buf[i0 + Q], c = buf[i0 + 2 * Q], d = buf[i0 + 3 * Q];
            bfly4_fwd(a, b, c, d);
            if (jj != 0) {
                const int e = jj * step;
                b = cmul(b, W[e]);
                c = cmul(c, W[2 * e]);
                d = cmul(d, W[3 * e]);
            }
            buf[i0] = a; buf[i0 + Q] = b; buf[i0 + 2 * Q] = c; buf[i0 + 3 * Q] = d;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void lds_fft_inv(c64 *buf, const c64 *__restrict__ W, int n, int tid)
{
    int lg = 0;
    while ((1 << lg) < n) lg++;
    const int n4 = (lg & 1) ? n / 2 : n;
    for (int L = 4; L <= n4; L <<= 2) {
        const int Q = L / 4, step = n / L;
        for (int t = tid; t < n / 4; t += 256) {
            const int jj = t & (Q - 1);
            const int i0 = (t / Q) * L + jj;
            c64 a = buf[i0], b = buf[i0 + Q], c = buf[i0 + 2 * Q], d = buf[i0 + 3 * Q];
            if (jj != 0) {
                const int e = jj * step;
                b = cmul_conj(b, W[e]);
                c = cmul_conj(c, W[2 * e]);
                d = cmul_conj(d, W[3 * e]);
            }
            bfly4_inv(a, b, c, d);
            buf[i0] = a; buf[i0 + Q] = b; buf[i0 + 2 * Q] = c; buf[i0 + 3 * Q] = d;
        }
        __syncthreads();
    }
    if (lg & 1) { // the mirror of the radix-2 stage
        const int h = n / 2;
        for (int j = tid; j < h; j += 256) {
            const c64 a = buf[j];
            c64 b = buf[j + h];
            if (j) b = cmul_conj(b, W[j]);
            c64 sm, df;
            sm.re = a.re + b.re; sm.im = a.im + b.im;
            df.re = a.re - b.re; df.im = a.im - b.im;
            buf[j] = sm;
            buf[j + h] = df;
        }
        __syncthreads();
    }
}

// =============================================================================================
// BSK -> Fourier (one-off): forward_as_torus (fft64/math/fft/mod.rs:378-386) on every polynomial of the key
// =============================================================================================
__global__ __launch_bounds__(256) void bsk_to_fourier_kernel(const uint64_t *__restrict__ bsk_std,
                                                             c64 *__restrict__ bsk_f, const c64 *__restrict__ twist,
                                                             const c64 *__restrict__ W, int n)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    c64 *buf = reinterpret_cast<c64 *>(smem);
    const int tid = threadIdx.x;
    const uint64_t *poly = bsk_std + (size_t)blockIdx.x * (2 * n);
    const double norm = 5.421010862427522170037264004349708557128906250e-20; // 2^-64
    for (int q = tid; q < n; q += 256) {
        c64 v;
        v.re = i64_to_f64((int64_t)poly[q]) * norm;
        v.im = i64_to_f64((int64_t)poly[q + n]) * norm;
        buf[q] = cmul(v, twist[q]);
    }
    __syncthreads();
    lds_fft_fwd(buf, W, n, tid);
    c64 *out = bsk_f + (size_t)blockIdx.x * n;
    for (int q = tid; q < n; q += 256) out[q] = buf[q];
}

hipError_t launch_bsk_to_fourier(const DevParams &p, const uint64_t *d_bsk_std, double *d_bsk_f, size_t n_polys,
                                 const double *d_twist, const double *d_w, hipStream_t s)
{
    if (p.N < 8 || (p.N & (p.N - 1)) || p.N > 8192) return hipErrorInvalidValue;
    const int n = p.N / 2;
    bsk_to_fourier_kernel<<<(unsigned)n_polys, 256, (size_t)n * sizeof(c64), s>>>(d_bsk_std, (c64 *)d_bsk_f, (const c64 *)d_twist,
                                                                                (const c64 *)d_w, n);
    return hipGetLastError();
}

// =============================================================================================
// Blind rotation, variant 0 (the generic kernel: LDS-resident accumulator and transform buffers)
// =============================================================================================
//
// One 256-thread workgroup per ciphertext; ANY GLWE dimension k = 1 ... 5, power-of-two polynomial size 256 <= N <= 4096 and
// level count whose images fit the LDS ((k+1) N <= 10240: what the wave kernels do not cover -- the reference string layer's
// own test set PARAM_MESSAGE_2_CARRY_2_TEST with N = 256, tfhe/examples/fhe_strings/ciphertext.rs:75-90, and the classic
// 1 ... 3-bit and 5-bit sets of shortint/parameters/mod.rs:599-760, k = 5 / 3 / 2 with N = 256 / 512 / 1024 and k = 1 with
// N = 4096).  Thread t owns the folded points q = t + 256 m (m < M), i.e. coefficients q and q + N/2 of all GS = k + 1 GLWE
// polynomials.  Dynamic LDS: acc [GS][N] u64, then fb [GS][N/2] c64 = 16 GS N bytes.
template <int M, int GS>
__global__ __launch_bounds__(256) void blind_rotate_v0_kernel(const uint64_t *__restrict__ lwe_small,
                                                              const uint64_t *__restrict__ luts,
                                                              const uint32_t *__restrict__ lut_idx,
                                                              const c64 *__restrict__ bsk_f,
                                                              const c64 *__restrict__ twist,
                                                              const c64 *__restrict__ W,
                                                              uint64_t *__restrict__ lwe_out, int n, int N, int base_log,
                                                              int level, const BrCtl ctl)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int FN = N / 2;
    int LOG2N = 0;
    while ((1 << LOG2N) < N) LOG2N++;
    uint64_t *acc[GS];
    c64 *fb[GS];
#pragma unroll
    for (int p = 0; p < GS; p++) {
        acc[p] = reinterpret_cast<uint64_t *>(smem) + (size_t)p * N;
        fb[p] = reinterpret_cast<c64 *>(smem + (size_t)GS * N * 8) + (size_t)p * FN;
    }
    const int tid = threadIdx.x;
    const int ct = blockIdx.x;
    const uint64_t *lwe = lwe_small + (size_t)ct * (n + 1);
    const uint64_t *lut = luts + (size_t)checked_lut_idx(lut_idx, ct, ctl.n_luts, ctl.err) * ((size_t)GS * N);

    // acc = lut * X^{-b_hat}   (polynomial_wrapping_monic_monomial_div, polynomial_algorithms.rs:315-354)
    {
        const uint32_t b_hat = modswitch(lwe[n], LOG2N);
        const int rem = b_hat & (N - 1);
        const int cyc = b_hat >> LOG2N;
        for (int c = tid; c < N; c += 256) {
            int src = c + rem;
            int neg = cyc & 1;
            if (src >= N) { src -= N; neg ^= 1; }
#pragma unroll
            for (int p = 0; p < GS; p++) {
                const uint64_t v = lut[(size_t)p * N + src];
                acc[p][c] = neg ? (uint64_t)0 - v : v;
            }
        }
    }
    __syncthreads();

    const uint64_t dmask = ((uint64_t)1 << base_log) - 1;
    const int dshift = 64 - base_log * level;
    const double inv_n = 1.0 / (double)FN;

    for (int i = 0; i < n; i++) {
        const uint64_t a = lwe[i];
        if (a == 0) continue; // bootstrap.rs:281 (uniform across the workgroup)
        const uint32_t a_hat = modswitch(a, LOG2N);
        const int rem = a_hat & (N - 1);
        const int cyc = a_hat >> LOG2N;

        // ct1 = acc * X^{a_hat} - acc (polynomial_algorithms.rs:425-490), then decomposition state
        uint64_t st[GS][M][2];
#pragma unroll
        for (int p = 0; p < GS; p++)
#pragma unroll
            for (int m = 0; m < M; m++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int q = tid + 256 * m;
                    if (q >= FN) { st[p][m][h] = 0; continue; }
                    const int c = q + FN * h;
                    int src = c - rem;
                    int neg = cyc & 1;
                    if (src < 0) { src += N; neg ^= 1; }
                    const uint64_t r = acc[p][src];
                    const uint64_t d = (neg ? (uint64_t)0 - r : r) - acc[p][c];
                    st[p][m][h] = closest_representable(d, base_log, level) >> dshift;
                }

        c64 of[GS][M]; // Fourier accumulators: [output poly][m]
        bool first = true;
        const c64 *ggsw = bsk_f + (size_t)i * level * (GS * GS) * FN;
        for (int lv = level; lv >= 1; lv--) { // ggsw.rs:524: highest level first
#pragma unroll
            for (int p = 0; p < GS; p++)
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const int q = tid + 256 * m;
                    if (q >= FN) continue;
                    c64 v;
                    v.re = (double)(int32_t)decompose_one_level(base_log, st[p][m][0], dmask);
                    v.im = (double)(int32_t)decompose_one_level(base_log, st[p][m][1], dmask);
                    fb[p][q] = cmul(v, twist[q]); // convert_forward_integer, x86.rs:562-563
                }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < GS; p++) lds_fft_fwd(fb[p], W, FN, tid); // (each call ends in a barrier)
            const c64 *lvl = ggsw + (size_t)(lv - 1) * (GS * GS) * FN;
#pragma unroll
            for (int row = 0; row < GS; row++) {
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const int q = tid + 256 * m;
                    if (q >= FN) continue;
                    const c64 f = fb[row][q];
#pragma unroll
                    for (int col = 0; col < GS; col++) {
                        const c64 g = lvl[(size_t)(row * GS + col) * FN + q];
                        of[col][m] = first ? cmul(g, f) : cmul_add(g, f, of[col][m]);
                    }
                }
                first = false;
            }
            // the next level's fold overwrites fb[p][q] at this thread's own q only: no barrier needed
        }

#pragma unroll
        for (int col = 0; col < GS; col++)
#pragma unroll
            for (int m = 0; m < M; m++)
                if (tid + 256 * m < FN) fb[col][tid + 256 * m] = of[col][m];
        __syncthreads();
#pragma unroll
        for (int col = 0; col < GS; col++) lds_fft_inv(fb[col], W, FN, tid);
        // convert_add_backward_torus (x86.rs:823-874): w' = w/n ; x*conj(w') ; from_torus ; wrapping add
#pragma unroll
        for (int col = 0; col < GS; col++)
#pragma unroll
            for (int m = 0; m < M; m++) {
                const int q = tid + 256 * m;
                if (q >= FN) continue;
                c64 w = twist[q];
                w.re *= inv_n;
                w.im *= inv_n;
                const c64 r = cmul_conj(fb[col][q], w);
                acc[col][q] += torus_from_f64(r.re);
                acc[col][q + FN] += torus_from_f64(r.im);
            }
        __syncthreads();
    }

    // extract_lwe_sample_from_glwe_ciphertext at X^0 (glwe_sample_extraction.rs:125-146): per mask polynomial a_p the
    // coefficients (a_p[0], -a_p[N-1], ..., -a_p[1]); the body's constant coefficient
    uint64_t *out = lwe_out + (size_t)ct * ((size_t)(GS - 1) * N + 1);
#pragma unroll
    for (int p = 0; p < GS - 1; p++)
        for (int j = tid; j < N; j += 256) out[(size_t)p * N + j] = (j == 0) ? acc[p][0] : (uint64_t)0 - acc[p][N - j];
    if (tid == 0) out[(size_t)(GS - 1) * N] = acc[GS - 1][0];
}

bool blind_rotate_v0_supported(const DevParams &p)
{
    return p.k >= 1 && p.k <= 5 && p.N >= 256 && p.N <= 4096 && (p.N & (p.N - 1)) == 0 && (size_t)(p.k + 1) * p.N <= 10240;
}

template <int M, int GS>
static hipError_t launch_v0(const DevParams &p, const uint64_t *d_lwe_small, const uint64_t *d_luts, const uint32_t *d_lut_idx,
                            const double *d_bsk_f, const double *d_twist, const double *d_w, uint64_t *d_lwe_out, size_t B,
                            const BrCtl &ctl, hipStream_t s)
{
    const size_t lds = (size_t)16 * GS * p.N;
    if (lds > 65536) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&blind_rotate_v0_kernel<M, GS>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    blind_rotate_v0_kernel<M, GS><<<(unsigned)B, 256, lds, s>>>(d_lwe_small, d_luts, d_lut_idx, (const c64 *)d_bsk_f,
                                                               (const c64 *)d_twist, (const c64 *)d_w, d_lwe_out, p.n, p.N,
                                                               p.pbs_base_log, p.pbs_level, ctl);
    return hipGetLastError();
}

template <int GS>
static hipError_t launch_v0_gs(const DevParams &p, const uint64_t *d_lwe_small, const uint64_t *d_luts, const uint32_t *d_lut_idx,
                               const double *d_bsk_f, const double *d_twist, const double *d_w, uint64_t *d_lwe_out, size_t B,
                               const BrCtl &ctl, hipStream_t s)
{
    const int pts = p.N / 2;
    if (pts <= 256) return launch_v0<1, GS>(p, d_lwe_small, d_luts, d_lut_idx, d_bsk_f, d_twist, d_w, d_lwe_out, B, ctl, s);
    if (pts == 512) return launch_v0<2, GS>(p, d_lwe_small, d_luts, d_lut_idx, d_bsk_f, d_twist, d_w, d_lwe_out, B, ctl, s);
    if (pts == 1024) return launch_v0<4, GS>(p, d_lwe_small, d_luts, d_lut_idx, d_bsk_f, d_twist, d_w, d_lwe_out, B, ctl, s);
    if (GS == 2) return launch_v0<8, 2>(p, d_lwe_small, d_luts, d_lut_idx, d_bsk_f, d_twist, d_w, d_lwe_out, B, ctl, s);
    return hipErrorInvalidValue; // N = 4096 only fits with k = 1
}

hipError_t launch_blind_rotate_v0(const DevParams &p, const uint64_t *d_lwe_small, const uint64_t *d_luts,
                                  const uint32_t *d_lut_idx, const double *d_bsk_f, const double *d_twist,
                                  const double *d_w, uint64_t *d_lwe_out, size_t B, const BrCtl &ctl, hipStream_t s)
{
    if (!blind_rotate_v0_supported(p)) return hipErrorInvalidValue;
    switch (p.k) {
    case 1: return launch_v0_gs<2>(p, d_lwe_small, d_luts, d_lut_idx, d_bsk_f, d_twist, d_w, d_lwe_out, B, ctl, s);
    case 2: return launch_v0_gs<3>(p, d_lwe_small, d_luts, d_lut_idx, d_bsk_f, d_twist, d_w, d_lwe_out, B, ctl, s);
    case 3: return launch_v0_gs<4>(p, d_lwe_small, d_luts, d_lut_idx, d_bsk_f, d_twist, d_w, d_lwe_out, B, ctl, s);
    case 4: return launch_v0_gs<5>(p, d_lwe_small, d_luts, d_lut_idx, d_bsk_f, d_twist, d_w, d_lwe_out, B, ctl, s);
    default: return launch_v0_gs<6>(p, d_lwe_small, d_luts, d_lut_idx, d_bsk_f, d_twist, d_w, d_lwe_out, B, ctl, s);
    }
}

// =============================================================================================
// Leveled operations on the device (host_ops.hpp keeps ciphertexts resident between the levels of an op-graph)
// =============================================================================================
// dst[r][w] = sum_{t in row r} coef[t] * src[t][w]  (+ body[r] on the last word): the lazy linear combinations of
// the host layer (lwe_ciphertext_add / cleartext_mul / plaintext_add chains, core_crypto/algorithms/lwe_linear_algebra.rs)
// evaluated once, straight into the input rows of the next batched launch.  HBM-bound gather: every source word is read
// once, coalesced (a workgroup walks one row).
__global__ __launch_bounds__(256) void lincomb_rows_kernel(uint64_t *__restrict__ dst, int row_words, const uint32_t *__restrict__ row_ptr,
                                                           const uint64_t *__restrict__ coef, const uint64_t *const *__restrict__ src,
                                                           const uint64_t *__restrict__ body)
{
    const size_t r = blockIdx.x;
    const uint32_t t0 = row_ptr[r], t1 = row_ptr[r + 1];
    uint64_t *out = dst + r * (size_t)row_words;
    for (int w = threadIdx.x; w < row_words; w += blockDim.x) {
        uint64_t acc = (w == row_words - 1) ? body[r] : 0;
        for (uint32_t t = t0; t < t1; t++) acc += coef[t] * src[t][w];
        out[w] = acc;
    }
}

hipError_t launch_lincomb_rows(uint64_t *d_dst, size_t rows, int row_words, const uint32_t *d_row_ptr, const uint64_t *d_coef,
                               const uint64_t *const *d_src, const uint64_t *d_body, hipStream_t s)
{
    if (rows == 0) return hipSuccess;
    if (rows > 0x7fffffffull || row_words <= 0) return hipErrorInvalidValue;
    lincomb_rows_kernel<<<(unsigned)rows, 256, 0, s>>>(d_dst, row_words, d_row_ptr, d_coef, d_src, d_body);
    return hipGetLastError();
}

} // namespace tfhe_hip
