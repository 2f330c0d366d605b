// hipcc 7.2 / gfx950: a DPP move whose source is the HIGH element of a float2 that a v_pk_fma_f32 just produced
// is emitted with the pair's LOW register as its source (seen in csrc/wpt4.hip's lattice level: the neighbour lane
// received A instead of B).  Pinning the element in a register of its own (empty asm with a "+v" constraint) gives
// the right code.  Check:  hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only dpp_subreg.hip -o - | grep -B3 dpp
// expected: the v_mov_b32_dpp of `bad` reads the odd register of the pair written by the v_pk_fma_f32 before it.
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const f2* in, f2* out, f2 ab) {
    f2 P[2];
    for (int j = 0; j < 2; ++j) {
        const f2 eo = in[threadIdx.x * 2 + j];
        P[j] = __builtin_elementwise_fma(ab, f2{eo.y, eo.y}, f2{eo.x, eo.x});
    }
    const float bad = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, P[1].y), 0x138, 0xF, 0xF, false));
    float by = P[1].y;
    asm volatile("" : "+v"(by));
    const float good = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, by), 0x138, 0xF, 0xF, false));
    P[1] = __builtin_elementwise_fma(ab, f2{P[0].y, P[0].y}, f2{P[1].x, P[1].x});
    P[0] = __builtin_elementwise_fma(ab, f2{bad, good}, f2{P[0].x, P[0].x});
    out[threadIdx.x * 2] = P[0];
    out[threadIdx.x * 2 + 1] = P[1];
}
