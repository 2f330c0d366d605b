// What fp32 operands as three bf16 terms (x = x1 + x2 + x3, six bf16 products per fp32 product) buy and cost on
// gfx950 -- the costing behind DESIGN section 7, item 1, as a measurement:
//   (1) accuracy: a 16 x 16 x K product of random fp32 matrices through the six-term split against float64, beside the
//       plain f32 MFMA result;
//   (2) matrix time per K = 32 of one 16 x 16 tile and wave: 8 x v_mfma_f32_16x16x4_f32 against 6 x
//       v_mfma_f32_16x16x32_bf16 (one wave per SIMD, four independent accumulators);
//   (3) the same with the split of ONE operand done on the vector ALU inside the loop (8 fp32 values per lane -> 3 x 8
//       bf16), which is what a kernel that produces its operand on the fly (the Winograd transforms) would pay.
// build: hipcc -O3 --offload-arch=gfx950 bf16x3.hip -o bf16x3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& p1, bf16x8& p2, bf16x8& p3) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 a = (__bf16)x[i];
        const float r1 = x[i] - (float)a;
        const __bf16 b = (__bf16)r1;
        const float r2 = r1 - (float)b;
        p1[i] = a; p2[i] = b; p3[i] = (__bf16)r2;
    }
}

// A [16][K], B [K][16] row-major fp32, K % 32 == 0; one wave.  out[0]: f32 MFMA, out[1]: six-term bf16 split
__global__ void accuracy_kernel(const float* A, const float* B, float* out, int K) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    f32x4 c32 = {0, 0, 0, 0}, c16 = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
        for (int s = 0; s < 8; ++s)  // 16x16x4: A lane (row r, k = 4 s + g), B lane (col r, k = 4 s + g)
            c32 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k0 + 4 * s + g], B[(k0 + 4 * s + g) * 16 + r], c32, 0, 0, 0);
        float a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // 16x16x32: lane (row / col r, k = 8 g + i)
            a[i] = A[r * K + k0 + 8 * g + i];
            b[i] = B[(k0 + 8 * g + i) * 16 + r];
        }
        bf16x8 a1, a2, a3, b1, b2, b3;
        split3(a, a1, a2, a3);
        split3(b, b1, b2, b3);
        // smallest terms first
        c16 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, c16, 0, 0, 0);
        c16 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, c16, 0, 0, 0);
        c16 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, c16, 0, 0, 0);
        c16 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, c16, 0, 0, 0);
        c16 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, c16, 0, 0, 0);
        c16 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, c16, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // D: row 4 g + j, column r
        out[(4 * g + j) * 16 + r] = c32[j];
        out[256 + (4 * g + j) * 16 + r] = c16[j];
    }
}

// MODE 0: f32 (8 x 16x16x4 per step), 1: bf16 split operands ready (6 x 16x16x32), 2: one operand split in the loop
template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(const float* src, float* out, int steps, long long* cycles) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0, 0, 0, 0};
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = src[lane * 8 + i];
    bf16x8 p1, p2, p3, q1, q2, q3;
    split3(x, p1, p2, p3);
    split3(x, q1, q2, q3);
    const long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[k], x[(k + t) & 7], acc[t], 0, 0, 0);
            } else {
                if (MODE == 2) {
                    float y[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) y[i] = x[i] + acc[t][i & 3];  // an operand that depends on running data
                    split3(y, q1, q2, q3);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p1, q3, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p3, q1, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p2, q2, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p1, q2, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p2, q1, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p1, q1, acc[t], 0, 0, 0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) v += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = v;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    const int K = 512;
    std::vector<float> A(16 * K), B(K * 16);
    srand(3);
    for (auto& v : A) v = 2.f * rand() / RAND_MAX - 1.f;
    for (auto& v : B) v = 2.f * rand() / RAND_MAX - 1.f;
    float *dA, *dB, *dO;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dO, 512 * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(accuracy_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dO, K);
    std::vector<float> O(512);
    (void)hipMemcpy(O.data(), dO, 512 * 4, hipMemcpyDeviceToHost);
    double e32 = 0, e16 = 0, mx = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)A[i * K + k] * (double)B[k * 16 + j];
            mx = fmax(mx, fabs(ref));
            e32 = fmax(e32, fabs(O[i * 16 + j] - ref));
            e16 = fmax(e16, fabs(O[256 + i * 16 + j] - ref));
        }
    printf("accuracy, K = %d, entries in [-1, 1]: largest |C| %.3f; max error f32 MFMA %.3e, six-term bf16 split %.3e\n", K, mx, e32, e16);
    float *dS, *dR;
    long long* dC;
    (void)hipMalloc(&dS, 64 * 8 * 4); (void)hipMalloc(&dR, 1024 * 256 * 4); (void)hipMalloc(&dC, 8);
    (void)hipMemcpy(dS, A.data(), 64 * 8 * 4, hipMemcpyHostToDevice);
    const int steps = 2000;
    const char* names[3] = {"f32: 8 x 16x16x4 per K = 32", "bf16 x 3: 6 x 16x16x32 per K = 32, operands ready",
                            "bf16 x 3 with one operand split in the loop"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(256), dim3(256), 0, 0, dS, dR, steps, dC);
            if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(256), dim3(256), 0, 0, dS, dR, steps, dC);
            if (mode == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(256), dim3(256), 0, 0, dS, dR, steps, dC);
            (void)hipDeviceSynchronize();
        }
        long long c = 0;
        (void)hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
        printf("%-52s %8.1f shader-clock cycles per tile step of K = 32 (one wave per SIMD, 4 tiles in flight)\n", names[mode],
               (double)c / steps / 4.0);
    }
    return 0;
}
