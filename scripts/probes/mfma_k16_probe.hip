// cycles per MFMA, back to back on one wave per SIMD: v_mfma_f32_16x16x32_bf16 vs the K = 16 form (v_mfma_f32_16x16x16_bf16)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned* ticks, int n) {
    f32x4 acc[4] = {};
    bf16x8 a, b;
    s16x4 a4, b4;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
    for (int e = 0; e < 4; ++e) { a4[e] = (short)(threadIdx.x + e); b4[e] = (short)(e + 1); }
    const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a4), "v"(b4));
        }
    }
    const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[MODE] = t1 - t0;
}
int main() {
    float* out; unsigned* ticks; unsigned h[2];
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 8);
    const int n = 10000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, ticks, n);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, out, ticks, 20 * n);       // 4 workgroups per CU: every SIMD 4 waves deep
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(h, ticks, 8, hipMemcpyDeviceToHost);
    printf("loaded chip: %u s_memtime ticks in a %.3f ms kernel = %.3f ticks per ns\n", h[0], ms, h[0] / (ms * 1e6));
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, ticks, n);
    hipDeviceSynchronize();
    hipMemcpy(h, ticks, 8, hipMemcpyDeviceToHost);
    printf("16x16x32: %.2f ticks per MFMA   16x16x16: %.2f ticks per MFMA\n", h[0] / (4.0 * n), h[1] / (4.0 * n));
    return 0;
}
