// mfma_f64_probe.hip -- two facts about v_mfma_f64_16x16x4_f64 on gfx950 that a "16-point DFTs on the matrix cores"
// variant of the blind rotation would rest on:
//   (1) numerics: is D = A*B + C bit for bit a k-ordered chain of IEEE fma's (which order)?
//   (2) does a wavefront's f64 MFMA stream overlap with its own / its neighbour's f64 VALU stream, or do they serialise?
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_f64_probe tools/probes/mfma_f64_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

using f64x4 = __attribute__((ext_vector_type(4))) double;

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                         \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

// one MFMA: A 16x4 (lane l: A[l & 15][l >> 4]), B 4x16 (lane l: B[l >> 4][l & 15]), C/D: col = l & 15, row = (l >> 4) + 4 reg
__global__ void one_mfma(const double *A, const double *B, const double *C, double *D)
{
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)], b = B[(l >> 4) * 16 + (l & 15)];
    f64x4 c;
    for (int r = 0; r < 4; r++) c[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
    const f64x4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = d[r];
}

// throughput loops: MODE 1 = VALU fma chains only, 2 = MFMA only, 3 = both interleaved in one wavefront
template <int MODE> __global__ __launch_bounds__(256) void pipes(double *out, int iters, double x)
{
    double v[8];
    for (int i = 0; i < 8; i++) v[i] = x + i + threadIdx.x;
    f64x4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = (f64x4){x, x, x, x};
    const double a = x * 0.5, b = x * 0.25;
    for (int it = 0; it < iters; it++) {
        if (MODE & 2) {
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        if (MODE & 1) {
#pragma unroll
            for (int rep = 0; rep < 8; rep++)
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = fma(v[i], a, b); // 64 VALU f64 per iteration
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += v[i];
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> static float time_mode(double *d_out, int iters, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    pipes<MODE><<<blocks, 256>>>(d_out, 16, 1.0);
    hipEventRecord(e0);
    pipes<MODE><<<blocks, 256>>>(d_out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<double> A(64), B(64), C(256), D(256);
    double *dA, *dB, *dC, *dD;
    CK(hipMalloc(&dA, 64 * 8));
    CK(hipMalloc(&dB, 64 * 8));
    CK(hipMalloc(&dC, 256 * 8));
    CK(hipMalloc(&dD, 256 * 8));
    int fwd = 0, bwd = 0, pair = 0, other = 0;
    for (int trial = 0; trial < 200; trial++) {
        for (auto &v : A) v = U(rng) * std::ldexp(1.0, (int)(rng() % 40) - 20);
        for (auto &v : B) v = U(rng) * std::ldexp(1.0, (int)(rng() % 40) - 20);
        for (auto &v : C) v = U(rng) * std::ldexp(1.0, (int)(rng() % 40) - 20);
        CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dC, C.data(), 256 * 8, hipMemcpyHostToDevice));
        one_mfma<<<1, 64>>>(dA, dB, dC, dD);
        CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                const double c = C[i * 16 + j];
                double f = c, g = c;
                for (int k = 0; k < 4; k++) f = std::fma(A[i * 4 + k], B[k * 16 + j], f);      // k = 0 first
                for (int k = 3; k >= 0; k--) g = std::fma(A[i * 4 + k], B[k * 16 + j], g);     // k = 3 first
                // pairwise: (a0 b0 + a1 b1) + (a2 b2 + a3 b3) + c, each inner sum an fma
                const double p = std::fma(A[i * 4 + 1], B[16 + j], A[i * 4] * B[j]) + std::fma(A[i * 4 + 3], B[48 + j], A[i * 4 + 2] * B[32 + j]) + c;
                const double d = D[i * 16 + j];
                if (d == f) fwd++;
                else if (d == g) bwd++;
                else if (d == p) pair++;
                else other++;
            }
    }
    std::printf("numerics over %d outputs: == fma chain k=0..3: %d, == chain k=3..0: %d, == pairwise: %d, none of these: %d\n",
                200 * 256, fwd, bwd, pair, other);
    // overlap
    double *d_out;
    const int blocks = 256 * 2; // two workgroups of 4 wavefronts per CU: 2 wavefronts per SIMD
    CK(hipMalloc(&d_out, (size_t)blocks * 256 * 8));
    const int iters = 20000;
    const float tv = time_mode<1>(d_out, iters, blocks), tm = time_mode<2>(d_out, iters, blocks), tb = time_mode<3>(d_out, iters, blocks);
    std::printf("per iteration and wavefront: 64 VALU f64 fma, 4 MFMA f64 16x16x4; %d blocks x 4 waves, %d iterations\n", blocks, iters);
    std::printf("VALU only %.3f ms, MFMA only %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n", tv, tm, tb, tv + tm, tv > tm ? tv : tm);
    const double waves_per_simd = blocks * 4.0 / (256 * 4);
    std::printf("cycles per VALU fma ~ %.2f, per MFMA ~ %.1f (assuming 2.0 GHz, %.1f waves per SIMD)\n",
                tv * 1e-3 * 2.0e9 / (iters * 64.0 * waves_per_simd), tm * 1e-3 * 2.0e9 / (iters * 4.0 * waves_per_simd), waves_per_simd);
    return 0;
}
