// Which XCD does block b of a 1-D grid land on?  (HW_REG_XCC_ID per block; speed-only knowledge.)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* o) {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    if (threadIdx.x == 0) o[blockIdx.x] = x;
}
int main() {
    const int n = 1024;
    int* d;
    hipMalloc(&d, n * sizeof(int));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
        int h[n];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int hist[16] = {0}, match = 0;
        for (int i = 0; i < n; ++i) { hist[h[i] & 15]++; match += ((h[i] & 7) == ((h[0] + i) & 7)); }
        printf("rep %d first 24:", rep);
        for (int i = 0; i < 24; ++i) printf(" %d", h[i]);
        printf("\n  histogram:");
        for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
        printf("\n  blocks with xcc == (xcc[0] + b) %% 8: %d of %d\n", match, n);
    }
    return 0;
}
