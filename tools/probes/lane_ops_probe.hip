// lane_ops_probe.hip -- issue cost of the cross-lane primitives the blind-rotation kernels choose between (gfx950):
// v_permlane32_swap / v_permlane16_swap, v_mov_b32_dpp (row_ror:8, quad_perm), against plain v_fma_f64 / v_mov_b32.
// Two wavefronts per SIMD (the kernels' occupancy), long dependent-free streams.
// Build: hipcc --offload-arch=gfx950 -O2 -o lane_ops_probe tools/probes/lane_ops_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP8(x) x x x x x x x x

template <int MODE> __global__ __launch_bounds__(256) void stream(unsigned *out, int iters)
{
    unsigned a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 9, a5 = a0 + 11, a6 = a0 + 13, a7 = a0 + 15;
    double f0 = a0, f1 = a1, f2 = a2, f3 = a3;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { // 32 plain moves / adds
            REP8(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 1) { // 32 v_permlane32_swap
            REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 2) { // 32 v_permlane16_swap
            REP8(asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 3) { // 32 dpp moves, row_ror:8
            REP8(asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %2, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_ror:8 row_mask:0xf bank_mask:0xf"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 4) { // 32 dpp moves, quad_perm
            REP8(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else { // 32 f64 fma on 4 chains
            REP8(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %2, %2, %3, %0\n v_fma_f64 %3, %3, %0, %1" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)(f0 + f1 + f2 + f3);
}

template <int MODE> static float run(unsigned *d_out, int iters, int blocks)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    stream<MODE><<<blocks, 256>>>(d_out, 8);
    (void)hipEventRecord(e0);
    stream<MODE><<<blocks, 256>>>(d_out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    unsigned *d_out;
    const int blocks = 512, iters = 20000; // 2 wavefronts per SIMD
    if (hipMalloc(&d_out, (size_t)blocks * 256 * 4) != hipSuccess) return 1;
    const char *names[6] = {"v_add_u32", "v_permlane32_swap_b32", "v_permlane16_swap_b32", "v_mov_b32_dpp row_ror:8", "v_mov_b32_dpp quad_perm", "v_fma_f64"};
    float ms[6] = {run<0>(d_out, iters, blocks), run<1>(d_out, iters, blocks), run<2>(d_out, iters, blocks),
                   run<3>(d_out, iters, blocks), run<4>(d_out, iters, blocks), run<5>(d_out, iters, blocks)};
    for (int m = 0; m < 6; m++)
        std::printf("%-28s %.3f ms for %d x 32 per wavefront, 2 wavefronts per SIMD: %.2f x the cost of v_add_u32\n", names[m], ms[m], iters, ms[m] / ms[0]);
    return 0;
}
