// MFMA-only power probe: which bf16 MFMA shape gives the most FLOP/s under the socket power cap when
// nothing but the matrix pipe and the register file works?  Every wave keeps independent accumulators and issues MFMAs back to
// back on register operands (no LDS, no memory).  Usage: mfma_power <type> <seconds>   type: 16 = v_mfma_f32_16x16x32_bf16, 17 = the same on varying operands,
// 32 = v_mfma_f32_32x32x16_bf16, 15 = the K = 16 form v_mfma_f32_16x16x16_bf16 (is a 16-deep tail half the price of a 32-deep step?).  Prints TFLOP/s; scripts/gpu_mfma_power.sh samples rocm-smi beside it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k16(float* out, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}
// type 17: the same instruction on operands that differ from one MFMA to the next (16 register pairs of hashed bit patterns,
// cycled): operand buses and multipliers toggle as they do on real data
__global__ __launch_bounds__(512) void k16r(float* out, int iters) {
    bf16x8 a[16], b[16];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 16; ++i)
        for (int e = 0; e < 8; ++e) {
            h = h * 1664525u + 1013904223u; a[i][e] = (__bf16)(((int)(h >> 16) % 2001 - 1000) * 1e-3f);
            h = h * 1664525u + 1013904223u; b[i][e] = (__bf16)(((int)(h >> 16) % 2001 - 1000) * 1e-3f);
        }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[(i + 5) & 15], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
// type 15: v_mfma_f32_16x16x16_bf16 (4 bf16 per lane and operand): half the FLOPs of the 16x16x32 form per instruction
__global__ __launch_bounds__(512) void k15(float* out, int iters) {
    s16x4 a, b;
    for (int e = 0; e < 4; ++e) { a[e] = (short)(0x3c00 + threadIdx.x + e); b[e] = (short)(0x3b00 + threadIdx.x * 3 - e); }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(512) void k32(float* out, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
    if (s == 12345.678f) out[0] = s;
}

int main(int argc, char** argv) {
    const int type = argc > 1 ? atoi(argv[1]) : 16;
    const double secs = argc > 2 ? atof(argv[2]) : 3.0;
    float* out;
    hipMalloc(&out, 4);
    const int iters = 20000, blocks = 256;                  // one 8-wave workgroup per CU: two waves per SIMD
    const double flop_per_launch = (type == 15 ? 16.0 * 2 * 16 * 16 * 16 : type != 32 ? 16.0 * 2 * 16 * 16 * 32 : 8.0 * 2 * 32 * 32 * 16) * iters * 8 * blocks;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto t0 = std::chrono::steady_clock::now();
    double flops = 0, ms_total = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        hipEventRecord(e0);
        for (int r = 0; r < 4; ++r) {
            if (type == 15) hipLaunchKernelGGL(k15, dim3(blocks), dim3(512), 0, 0, out, iters);
            else if (type == 16) hipLaunchKernelGGL(k16, dim3(blocks), dim3(512), 0, 0, out, iters);
            else if (type == 17) hipLaunchKernelGGL(k16r, dim3(blocks), dim3(512), 0, 0, out, iters);
            else hipLaunchKernelGGL(k32, dim3(blocks), dim3(512), 0, 0, out, iters);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ms_total += ms; flops += 4 * flop_per_launch;
    }
    printf("mfma %s: %.1f TFLOP/s over %.1f s\n", type == 15 ? "16x16x16 (constant operands)" : type == 16 ? "16x16x32 (constant operands)" : type == 17 ? "16x16x32 (varying operands)" : "32x32x16 (constant operands)", flops / (ms_total * 1e-3) / 1e12, ms_total * 1e-3);
    return 0;
}
