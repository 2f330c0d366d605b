// What the DPP shift modes deliver on this chip: lane i of the result should hold lane i - 1's value.
// build: hipcc -O2 --offload-arch=gfx950 dpp_shift.hip -o dpp_shift
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int v = threadIdx.x;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, false);        // wave_shr:1
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xF, 0xF, false);   // row_shr:1
}
int main() {
    int* d;
    hipMalloc(&d, 128 * sizeof(int));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[128];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave_shr:1:");
    for (int i = 0; i < 64; ++i) printf(" %d", h[i]);
    printf("\nrow_shr:1: ");
    for (int i = 0; i < 64; ++i) printf(" %d", h[64 + i]);
    printf("\n");
    return 0;
}
