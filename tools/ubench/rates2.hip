// v_cndmask issue cost in its different forms (follow-up of rates.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP> __global__ __launch_bounds__(64) void k(unsigned long long *out, int iters, unsigned b)
{
    unsigned x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned long long m = 0x5555555555555555ull ^ b;
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { asm volatile("s_mov_b64 vcc, %9\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "s"(m) : "vcc"); }
        if (OP == 1) { asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "s"(m)); }
        if (OP == 2) { asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_lt_u32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_u32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b) : "vcc"); }
        if (OP == 3) { asm volatile("v_bfi_b32 %0, %8, %0, %1\n v_bfi_b32 %1, %8, %1, %2\n v_bfi_b32 %2, %8, %2, %3\n v_bfi_b32 %3, %8, %3, %4\n v_bfi_b32 %4, %8, %4, %5\n v_bfi_b32 %5, %8, %5, %6\n v_bfi_b32 %6, %8, %6, %7\n v_bfi_b32 %7, %8, %7, %0" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b)); }
        if (OP == 4) { asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (OP == 5) { asm volatile("v_bfe_u32 %0, %0, 3, 7\n v_bfe_u32 %1, %1, 3, 7\n v_bfe_u32 %2, %2, 3, 7\n v_bfe_u32 %3, %3, 3, 7\n v_bfe_u32 %4, %4, 3, 7\n v_bfe_u32 %5, %5, 3, 7\n v_bfe_u32 %6, %6, 3, 7\n v_bfe_u32 %7, %7, 3, 7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int OP> void run(const char *name)
{
    const int iters = 20000;
    unsigned long long *o;
    hipMalloc(&o, 256 * 4 * 2 * 64 * 8);
    for (int w = 1; w <= 2; w++) {
        const int blocks = 256 * 4 * w;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, 64>>>(o, 100, 3u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<OP><<<blocks, 64>>>(o, iters, 3u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s waves/SIMD=%d  %.3f ns per wave-instruction (SIMD-time)\n", name, w, ms * 1e6 / iters / 8 / w);
    }
    hipFree(o);
}
int main()
{
    run<0>("v_cndmask_b32 (vcc, s_mov once)"); run<1>("v_cndmask_b32_e64 (sgpr pair)"); run<2>("v_cmp + v_cndmask pairs");
    run<3>("v_bfi_b32"); run<4>("v_permlane16_swap"); run<5>("v_bfe_u32");
    return 0;
}
