// Issue cost of the VALU instructions the blind-rotation kernels use (wave64, gfx950): N independent
// chains per wave, 1 or 2 waves per SIMD; prints ns per wave-instruction and cycles at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(x) x x x x x x x x
template <int OP> __global__ __launch_bounds__(64) void k(unsigned long long *out, int iters, double a, unsigned b)
{
    double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    unsigned long long u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    unsigned x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { asm volatile("v_fma_f64 %0, %0, %8, %0\n v_fma_f64 %1, %1, %8, %1\n v_fma_f64 %2, %2, %8, %2\n v_fma_f64 %3, %3, %8, %3\n v_fma_f64 %4, %4, %8, %4\n v_fma_f64 %5, %5, %8, %5\n v_fma_f64 %6, %6, %8, %6\n v_fma_f64 %7, %7, %8, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a)); }
        if (OP == 1) { asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a)); }
        if (OP == 2) { asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a)); }
        if (OP == 3) { asm volatile("v_cvt_f64_i32 %0, %8\n v_cvt_f64_i32 %1, %9\n v_cvt_f64_i32 %2, %10\n v_cvt_f64_i32 %3, %11\n v_cvt_f64_i32 %4, %12\n v_cvt_f64_i32 %5, %13\n v_cvt_f64_i32 %6, %14\n v_cvt_f64_i32 %7, %15" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7)); }
        if (OP == 4) { asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(u0 | 1)); }
        if (OP == 5) { asm volatile("v_lshlrev_b64 %0, %8, %0\n v_lshlrev_b64 %1, %8, %1\n v_lshlrev_b64 %2, %8, %2\n v_lshlrev_b64 %3, %8, %3\n v_lshlrev_b64 %4, %8, %4\n v_lshlrev_b64 %5, %8, %5\n v_lshlrev_b64 %6, %8, %6\n v_lshlrev_b64 %7, %8, %7" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b & 1)); }
        if (OP == 6) { asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b)); }
        if (OP == 7) { asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b)); }
        if (OP == 8) { asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (OP == 9) { asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b) : "vcc"); }
        if (OP == 10) { asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b)); }
        if (OP == 11) { asm volatile("v_sub_co_u32 %0, vcc, %0, %8\n v_subb_co_u32 %1, vcc, %1, %8, vcc\n v_sub_co_u32 %2, vcc, %2, %8\n v_subb_co_u32 %3, vcc, %3, %8, vcc\n v_sub_co_u32 %4, vcc, %4, %8\n v_subb_co_u32 %5, vcc, %5, %8, vcc\n v_sub_co_u32 %6, vcc, %6, %8\n v_subb_co_u32 %7, vcc, %7, %8, vcc" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b) : "vcc"); }
    }
    out[blockIdx.x * 64 + threadIdx.x] = (unsigned long long)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7 + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int OP> void run(const char *name)
{
    const int iters = 20000;
    unsigned long long *o;
    hipMalloc(&o, 256 * 4 * 2 * 64 * 8);
    for (int w = 1; w <= 2; w++) {
        const int blocks = 256 * 4 * w;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, 64>>>(o, 100, 1.0000001, 3u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<OP><<<blocks, 64>>>(o, iters, 1.0000001, 3u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-22s waves/SIMD=%d  %.3f ns per wave-instruction (SIMD-time)\n", name, w, ms * 1e6 / iters / 8 / w);
    }
    hipFree(o);
}

int main()
{
    run<0>("v_fma_f64"); run<1>("v_add_f64"); run<2>("v_mul_f64"); run<3>("v_cvt_f64_i32"); run<4>("v_lshl_add_u64");
    run<5>("v_lshlrev_b64"); run<6>("v_add_u32"); run<7>("v_xor_b32"); run<8>("v_permlane32_swap"); run<9>("v_cndmask_b32");
    run<10>("v_mov_b32"); run<11>("v_sub_co/subb_co");
    return 0;
}
