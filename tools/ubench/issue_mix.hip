// Microbenchmark: does a second wave per SIMD hide the non-f64 issue slots of an f64+int mix?
// Each wave runs the same dependent-free mix: F64 fma's + INT 32-bit ops per loop iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int F64, int INT>
__global__ __launch_bounds__(64) void mix(double *out, unsigned *iout, int iters, double a, unsigned b)
{
    double x[8];
    unsigned y[8];
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x + i; y[i] = threadIdx.x * 3 + i; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < F64; k++) x[k & 7] = __builtin_fma(x[k & 7], a, 1.0);
#pragma unroll
        for (int k = 0; k < INT; k++) y[k & 7] = (y[k & 7] ^ b) + (unsigned)k;
    }
    double s = 0; unsigned t = 0;
    for (int i = 0; i < 8; i++) { s += x[i]; t += y[i]; }
    out[blockIdx.x * 64 + threadIdx.x] = s;
    iout[blockIdx.x * 64 + threadIdx.x] = t;
}

template <int F64, int INT> void run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * 4 * waves_per_simd, iters = 20000;
    double *o; unsigned *io;
    hipMalloc(&o, blocks * 64 * 8); hipMalloc(&io, blocks * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix<F64, INT><<<blocks, 64>>>(o, io, 100, 1.0000001, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mix<F64, INT><<<blocks, 64>>>(o, io, iters, 1.0000001, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_iter_ns = ms * 1e6 / iters;
    printf("%-18s waves/SIMD=%d  %.3f ms  %.1f ns/iter  -> %.2f ns per wave-iter-equivalent\n", name, waves_per_simd, ms, per_iter_ns,
           per_iter_ns / waves_per_simd);
    hipFree(o); hipFree(io);
}

int main()
{
    for (int w = 1; w <= 2; w++) {
        run<32, 0>("f64x32", w);
        run<0, 32>("int x32", w);
        run<32, 32>("f64x32+intx32", w);
        run<26, 30>("f64x26+intx30", w);
    }
    return 0;
}
