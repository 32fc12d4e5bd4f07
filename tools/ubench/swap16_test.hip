#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out)
{
    unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
}
int main()
{
    unsigned *d, h[128];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("A' (vdst) : "); for (int i = 0; i < 64; i += 8) printf("%u ", h[i]); printf("\n");
    printf("B' (src0) : "); for (int i = 0; i < 64; i += 8) printf("%u ", h[64 + i]); printf("\n");
    return 0;
}
