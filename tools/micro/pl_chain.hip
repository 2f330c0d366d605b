// Reproducer: hipcc (ROCm 7.2) miscompiles chained __builtin_amdgcn_permlane16_swap / permlane32_swap on gfx950 --
// the second result of the builtin is lost (one register group is stored four times).  wpt_haar.hip uses inline asm instead.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2v __attribute__((ext_vector_type(2)));
template <int DIST>
__device__ __forceinline__ void lane_swap(float& a, float& b) {
    const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    const u2v r = DIST == 16 ? __builtin_amdgcn_permlane16_swap(ua, ub, false, false)
                             : __builtin_amdgcn_permlane32_swap(ua, ub, false, false);
    a = __builtin_bit_cast(float, r.x);
    b = __builtin_bit_cast(float, r.y);
}
__device__ __forceinline__ void row_transpose(float (&v)[16]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { lane_swap<16>(v[0 + c], v[4 + c]); lane_swap<16>(v[8 + c], v[12 + c]); }
#pragma unroll
    for (int c = 0; c < 4; ++c) { lane_swap<32>(v[0 + c], v[8 + c]); lane_swap<32>(v[4 + c], v[12 + c]); }
}
__global__ void k(const float* in, float* out) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[threadIdx.x * 16 + i];
    row_transpose(v);
    for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = v[i];
}
int main(){ float *d,*o; hipMalloc(&d,64*16*4); hipMalloc(&o,64*16*4); float h[1024]; for(int i=0;i<1024;i++) h[i]=i; hipMemcpy(d,h,4096,hipMemcpyHostToDevice);
 k<<<1,64>>>(d,o); hipMemcpy(h,o,4096,hipMemcpyDeviceToHost);
 int bad=0; for(int lane=0;lane<64;lane++) for(int j=0;j<4;j++) for(int c=0;c<4;c++){ int l=lane&15,kk=lane>>4; float want=(16*j+l)*16+4*kk+c; if(h[lane*16+4*j+c]!=want){ if(bad<8) printf("lane %d j %d c %d got %g want %g\n",lane,j,c,h[lane*16+4*j+c],want); bad++; } }
 printf("bad %d\n",bad); return 0; }
