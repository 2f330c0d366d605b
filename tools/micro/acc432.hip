// Can one wave hold the 432 accumulator registers of twelve 16 x 16 blocks x 36 Winograd positions (the "96 co x 32 ci per
// CU, one 512-register wave per SIMD" form of the backward-weight kernel)?  hipcc 7.2, gfx950:
//   hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only acc432.hip -o - | grep -E 'NumVgprs|NumAgprs|ScratchSize|Occupancy'
//   -> 236 VGPRs + 256 AGPRs, no scratch, occupancy 1; but every v_mfma takes its accumulator in the AGPR file and the
//   176 registers that do not fit there travel through v_accvgpr_write / v_accvgpr_read (grep -c v_accvgpr: 1 044, of
//   which 180 sit in the loop beside its 108 matrix instructions).  With 256 architectural VGPRs at most, 176 of them
//   accumulators, 80 registers remain for patch, operands, addresses and transform temporaries.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int rounds) {
    extern __shared__ float lds[];
    f32x4 acc[108];
#pragma unroll
    for (int p = 0; p < 108; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    for (int r = 0; r < rounds; ++r) {
        float d[36];
#pragma unroll
        for (int i = 0; i < 36; ++i) d[i] = a[(r * 36 + i) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 36; ++i) lds[i * 64 + lane] = d[i] * 2.f;
        __syncthreads();
#pragma unroll
        for (int pg = 0; pg < 9; ++pg) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(lds + pg * 256 + lane * 4);
#pragma unroll
            for (int blk = 0; blk < 3; ++blk) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(lds + 4096 + (blk * 9 + pg) * 256 + lane * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[blk * 36 + 4 * pg + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc[blk * 36 + 4 * pg + q], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < 108; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) out[(p * 4 + j) * 64 + lane] = acc[p][j];
}
