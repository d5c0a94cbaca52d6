// LoKr adapter support kernels for gfx950 (BASELINE config 5: ``lora_algo: lokr`` -> peft LoKrConfig(r, alpha,
// module_dropout, target_modules) wrapped around the model at /root/reference/common/trainer.py:212-238).
//
// [RECALL peft/tuners/lokr/layer.py]  For a target Linear / 1x1 Conv with weight [out, in]:
//   (out_l, out_k) = factorization(out), (in_m, in_n) = factorization(in);  w1 [out_l, in_m] (zeros at init),
//   w2 = w2_a [out_k, r] @ w2_b [r, in_n] (both kaiming-uniform);  delta_w = kron(w1, w2) * (alpha / r);
//   forward: base_layer(x) + F.linear(x, delta_w)   -- all in the module dtype (bf16), every op rounding.
// The dense products (x delta_w^T, dy delta_w, d_delta = dy^T x) run on the GEMM kernels of this library; what is left
// is HBM-bound glue, one launch per adapter:
//   yat_lokr_delta   : w2 = bf16(w2_a w2_b);  delta[(i,k),(j,n)] = bf16( bf16(w1[i,j] * w2[k,n]) * scale )
//   yat_lokr_project : autograd of the above from d_delta: d_w1, d_w2 (fp32 sums, fixed order), d_w2_a, d_w2_b.
#include "common.hpp"
#include "../../include/yat_hip.h"

namespace {

struct LokrP {
    int out_l, out_k, in_m, in_n, r, ld;
    float scale;
    const bf16_t* w1; const bf16_t* w2a; const bf16_t* w2b;
};

__device__ __forceinline__ float w2_elem(const LokrP& p, int k, int n) {
    float s = 0.f;
    for (int q = 0; q < p.r; ++q) s += bf2f(p.w2a[k * p.r + q]) * bf2f(p.w2b[q * p.in_n + n]);
    return rbf(s);                                             // the bf16 matmul w2_a @ w2_b
}

// grid (ceil(in / 256), out): one thread per element, consecutive threads along the row
__global__ __launch_bounds__(256) void lokr_delta_kernel(LokrP p, bf16_t* delta) {
    const int col = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    const int in = p.in_m * p.in_n;
    if (col >= in) return;
    const int i = row / p.out_k, k = row - i * p.out_k, j = col / p.in_n, n = col - j * p.in_n;
    float v = rbf(bf2f(p.w1[i * p.in_m + j]) * w2_elem(p, k, n));          // torch.kron in bf16
    if (p.scale != 1.0f) v = rbf(v * p.scale);                             // make_kron: rebuild * scale (skipped at 1)
    delta[(int64_t)row * p.ld + col] = f2bf(v);
}

// d_w1[i,j] = sum_{k,n} dR[(i,k),(j,n)] * w2[k,n],  dR = bf16(d_delta * scale).  One workgroup per (i,j).
__global__ __launch_bounds__(256) void lokr_dw1_kernel(LokrP p, const bf16_t* dd, bf16_t* dw1) {
    __shared__ float red[4];
    const int j = blockIdx.x, i = blockIdx.y;
    float s = 0.f;
    const int per = p.out_k * p.in_n;
    for (int e = threadIdx.x; e < per; e += 256) {
        const int k = e / p.in_n, n = e - k * p.in_n;
        float g = bf2f(dd[((int64_t)i * p.out_k + k) * p.ld + j * p.in_n + n]);
        if (p.scale != 1.0f) g = rbf(g * p.scale);
        s += g * w2_elem(p, k, n);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) dw1[i * p.in_m + j] = f2bf(red[0] + red[1] + red[2] + red[3]);
}

// partial[i][k*in_n + n] = sum_j dR[(i,k),(j,n)] * w1[i,j]   (one workgroup per i; summed over i by the next kernel)
__global__ __launch_bounds__(256) void lokr_dw2_partial_kernel(LokrP p, const bf16_t* dd, float* partial) {
    const int i = blockIdx.x;
    const int per = p.out_k * p.in_n;
    for (int e = threadIdx.x; e < per; e += 256) {
        const int k = e / p.in_n, n = e - k * p.in_n;
        const bf16_t* row = dd + ((int64_t)i * p.out_k + k) * p.ld + n;
        float s = 0.f;
        for (int j = 0; j < p.in_m; ++j) {
            float g = bf2f(row[j * p.in_n]);
            if (p.scale != 1.0f) g = rbf(g * p.scale);
            s += g * bf2f(p.w1[i * p.in_m + j]);
        }
        partial[(int64_t)i * per + e] = s;
    }
}

// d_w2 = bf16(sum_i partial[i]);  then d_w2_a = bf16(d_w2 w2_b^T), d_w2_b = bf16(w2_a^T d_w2) -- one workgroup, the
// d_w2 tile staged in LDS (out_k * in_n <= 16 K elements for every SANA target).
__global__ __launch_bounds__(256) void lokr_dw2_final_kernel(LokrP p, const float* partial, bf16_t* dw2a, bf16_t* dw2b) {
    extern __shared__ float dw2[];
    const int per = p.out_k * p.in_n;
    for (int e = threadIdx.x; e < per; e += 256) {
        float s = 0.f;
        for (int i = 0; i < p.out_l; ++i) s += partial[(int64_t)i * per + e];
        dw2[e] = rbf(s);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < p.out_k * p.r; e += 256) {          // d_w2_a[k, q] = sum_n d_w2[k, n] * w2_b[q, n]
        const int k = e / p.r, q = e - k * p.r;
        float s = 0.f;
        for (int n = 0; n < p.in_n; ++n) s += dw2[k * p.in_n + n] * bf2f(p.w2b[q * p.in_n + n]);
        dw2a[e] = f2bf(s);
    }
    for (int e = threadIdx.x; e < p.r * p.in_n; e += 256) {           // d_w2_b[q, n] = sum_k w2_a[k, q] * d_w2[k, n]
        const int q = e / p.in_n, n = e - q * p.in_n;
        float s = 0.f;
        for (int k = 0; k < p.out_k; ++k) s += bf2f(p.w2a[k * p.r + q]) * dw2[k * p.in_n + n];
        dw2b[e] = f2bf(s);
    }
}

// Small-output weight gradient of the factored adapter path: out[q, n] (+)= sum_row A[row, q] * X[row, n] with a few
// (R <= 16) x (N <= 128) outputs and a million-row reduction (rows = B*N_tokens*in_m) -- far outside what a tiled GEMM is
// for.  HBM-bound streaming pass: a wave walks rows, a lane owns two adjacent columns of X and all R accumulators for them;
// the four waves of a workgroup meet in LDS, every workgroup leaves one fp32 partial, a second launch sums the partials in a
// fixed order (no atomics).
template <int R>
__global__ __launch_bounds__(256) void lokr_small_wgrad_kernel(int64_t rows, int N, const bf16_t* A, const bf16_t* X,
                                                               float* partial) {
    __shared__ float red[4][R][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    const int c = 2 * lane;
    const bool live = c < N;
    float a0[R], a1[R];
#pragma unroll
    for (int q = 0; q < R; ++q) { a0[q] = 0.f; a1[q] = 0.f; }
    for (int64_t r = r0 + wave; r < r1; r += 4) {
        float av[R];
#pragma unroll
        for (int v = 0; v < R / 8; ++v) unpack8(*reinterpret_cast<const u32x4*>(A + r * R + v * 8), av + v * 8);
        const uint32_t xx = live ? *reinterpret_cast<const uint32_t*>(X + r * N + c) : 0u;
        const float x0 = __uint_as_float(xx << 16), x1 = __uint_as_float(xx & 0xffff0000u);
#pragma unroll
        for (int q = 0; q < R; ++q) { a0[q] += av[q] * x0; a1[q] += av[q] * x1; }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) { red[wave][q][c] = a0[q]; red[wave][q][c + 1] = a1[q]; }
    __syncthreads();
    for (int e = threadIdx.x; e < R * N; e += 256) {
        const int q = e / N, n = e - q * N;
        partial[(int64_t)blockIdx.x * R * N + e] = red[0][q][n] + red[1][q][n] + red[2][q][n] + red[3][q][n];
    }
}
// out[q, n] = (accumulate ? out : 0) + sum_g partial[g][q*N + n], q < r_out.  One wave per 16 outputs: lane = (g-slice, e):
// 4 slices of the partial list x 16 consecutive outputs (64-B runs), fixed order, then two shuffles.
__global__ __launch_bounds__(256) void lokr_small_wgrad_final_kernel(int G, int RN, int n_out, const float* partial,
                                                                     bf16_t* out, int accumulate) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = (blockIdx.x * 4 + wave) * 16 + (lane & 15), slice = lane >> 4;
    float s = 0.f;
    if (e < n_out)
        for (int g = slice; g < G; g += 4) s += partial[(int64_t)g * RN + e];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (e >= n_out || slice != 0) return;
    if (accumulate) s = rbf(s) + bf2f(out[e]);
    out[e] = f2bf(s);
}

int fill(LokrP& p, int out_l, int out_k, int in_m, int in_n, int r, const void* w1, const void* w2a, const void* w2b,
         float scale, int ld) {
    if (out_l <= 0 || out_k <= 0 || in_m <= 0 || in_n <= 0 || r <= 0 || r > 64 || !w1 || !w2a || !w2b ||
        ld < in_m * in_n || out_l * out_k > 65535)
        return YAT_EINVAL;
    p.out_l = out_l; p.out_k = out_k; p.in_m = in_m; p.in_n = in_n; p.r = r; p.ld = ld; p.scale = scale;
    p.w1 = (const bf16_t*)w1; p.w2a = (const bf16_t*)w2a; p.w2b = (const bf16_t*)w2b;
    return YAT_OK;
}

}  // namespace

extern "C" {

int yat_lokr_delta(int out_l, int out_k, int in_m, int in_n, int r, const void* w1, const void* w2_a, const void* w2_b,
                   float scale, void* delta, int ld, yat_stream_t stream) {
    LokrP p;
    if (fill(p, out_l, out_k, in_m, in_n, r, w1, w2_a, w2_b, scale, ld) || !delta) return YAT_EINVAL;
    hipLaunchKernelGGL(lokr_delta_kernel, dim3((in_m * in_n + 255) / 256, out_l * out_k), dim3(256), 0, (hipStream_t)stream, p,
                       (bf16_t*)delta);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

uint64_t yat_lokr_project_workspace_bytes(int out_l, int out_k, int in_n) {
    return (uint64_t)out_l * out_k * in_n * sizeof(float);
}

int yat_lokr_project(int out_l, int out_k, int in_m, int in_n, int r, const void* w1, const void* w2_a, const void* w2_b,
                     float scale, const void* d_delta, int ld, void* d_w1, void* d_w2_a, void* d_w2_b, void* workspace,
                     yat_stream_t stream) {
    LokrP p;
    if (fill(p, out_l, out_k, in_m, in_n, r, w1, w2_a, w2_b, scale, ld) || !d_delta || !d_w1 || !d_w2_a || !d_w2_b ||
        !workspace || (uint64_t)out_k * in_n * sizeof(float) > 65536)
        return YAT_EINVAL;
    hipLaunchKernelGGL(lokr_dw1_kernel, dim3(in_m, out_l), dim3(256), 0, (hipStream_t)stream, p, (const bf16_t*)d_delta,
                       (bf16_t*)d_w1);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(lokr_dw2_partial_kernel, dim3(out_l), dim3(256), 0, (hipStream_t)stream, p, (const bf16_t*)d_delta,
                       (float*)workspace);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(lokr_dw2_final_kernel, dim3(1), dim3(256), out_k * in_n * sizeof(float), (hipStream_t)stream, p,
                       (const float*)workspace, (bf16_t*)d_w2_a, (bf16_t*)d_w2_b);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

uint64_t yat_lokr_small_wgrad_workspace_bytes(int R, int N) { return (uint64_t)1024 * R * N * sizeof(float); }

int yat_lokr_small_wgrad(int64_t rows, int R, int N, int r_out, const void* a, const void* x, void* out, int accumulate,
                         void* workspace, yat_stream_t stream) {
    if (rows <= 0 || (R != 8 && R != 16) || N <= 0 || N > 128 || (N & 1) || r_out <= 0 || r_out > R || !a || !x || !out ||
        !workspace)
        return YAT_EINVAL;
    int64_t g64 = (rows + 255) / 256;
    const int G = (int)(g64 > 512 ? 512 : g64);          // two workgroups per CU; every extra partial lengthens the final pass
    if (R == 8)
        hipLaunchKernelGGL((lokr_small_wgrad_kernel<8>), dim3(G), dim3(256), 0, (hipStream_t)stream, rows, N, (const bf16_t*)a,
                           (const bf16_t*)x, (float*)workspace);
    else
        hipLaunchKernelGGL((lokr_small_wgrad_kernel<16>), dim3(G), dim3(256), 0, (hipStream_t)stream, rows, N, (const bf16_t*)a,
                           (const bf16_t*)x, (float*)workspace);
    YAT_CHECK_LAUNCH();
    const int n_out = r_out * N;
    hipLaunchKernelGGL(lokr_small_wgrad_final_kernel, dim3((n_out + 63) / 64), dim3(256), 0, (hipStream_t)stream, G, R * N,
                       n_out, (const float*)workspace, (bf16_t*)out, accumulate);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

}  // extern "C"
