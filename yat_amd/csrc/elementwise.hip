// Elementwise / recipe kernels for gfx950: activations, timestep sinusoid, ragged pad+mask,
// flow-matching mix and the MSE loss with its gradient.  All HBM-bound: 16-B accesses per lane,
// grid-stride over <= 2048 workgroups.
#include "common.hpp"
#include "../../include/yat_hip.h"

namespace {

__device__ __forceinline__ float act_f(int act, float x) { return act == 1 ? silu_f(x) : gelu_tanh_f(x); }
__device__ __forceinline__ float dact_f(int act, float x) { return act == 1 ? dsilu_f(x) : dgelu_tanh_f(x); }

inline unsigned grid_for(int64_t nvec) {
    int64_t b = (nvec + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// MODE 0: y = act(x); 1: dx = dy*act'(x); 2: out = a + b
template <int MODE>
__global__ void ew_kernel(int64_t n, int act, const bf16_t* a, const bf16_t* b, bf16_t* out) {
    const int64_t nvec = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        float x[8], y[8], o[8];
        unpack8(*reinterpret_cast<const u32x4*>(a + i * 8), x);
        if (MODE != 0) unpack8(*reinterpret_cast<const u32x4*>(b + i * 8), y);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (MODE == 0) o[e] = act_f(act, x[e]);
            else if (MODE == 1) o[e] = y[e] * dact_f(act, x[e]);
            else o[e] = x[e] + y[e];
        }
        *reinterpret_cast<u32x4*>(out + i * 8) = pack8(o);
    }
    // scalar tail
    if (blockIdx.x == 0) {
        for (int64_t i = (nvec << 3) + threadIdx.x; i < n; i += blockDim.x) {
            const float x = bf2f(a[i]);
            float o;
            if (MODE == 0) o = act_f(act, x);
            else if (MODE == 1) o = bf2f(b[i]) * dact_f(act, x);
            else o = x + bf2f(b[i]);
            out[i] = f2bf(o);
        }
    }
}

__global__ void f32_to_bf16_kernel(int64_t n, const float* x, bf16_t* y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = f2bf(x[i]);
}

// get_timestep_embedding(t, dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin], fp32 math
__global__ void timestep_embed_kernel(int B, int dim, const float* t, bf16_t* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim >> 1;
    if (i >= B * half) return;
    const int b = i / half, j = i % half;
    const float freq = expf(-logf(10000.0f) * (float)j / (float)half);
    const float arg = t[b] * freq;
    out[(int64_t)b * dim + j] = f2bf(cosf(arg));
    out[(int64_t)b * dim + half + j] = f2bf(sinf(arg));
}

// batched 2-D transpose through a padded LDS tile: in [B, R, Cc] -> out [B, Cc, R]
// (NCHW latents <-> token-major rows for patch-embed / unpatchify, patched_sana_transformer.py:284,336-340)
__global__ void transpose_kernel(int R, int Cc, const bf16_t* in, bf16_t* out) {
    __shared__ bf16_t tile[32][33];
    const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8 threads
    const bf16_t* ib = in + (int64_t)b * R * Cc;
    bf16_t* ob = out + (int64_t)b * R * Cc;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < R && c0 + tx < Cc) tile[k][tx] = ib[(int64_t)(r0 + k) * Cc + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < Cc && r0 + tx < R) ob[(int64_t)(c0 + k) * R + r0 + tx] = tile[tx][k];
}

// one wave per destination row: copy the source row or write zeros; lane 0 writes mask / bias
__global__ void pad_mask_kernel(int B, int T, int C, const bf16_t* src, const int* offsets, bf16_t* dst, int64_t* mask,
                                float* key_bias, int* kv_len) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= B * T) return;
    const int b = row / T, t = row % T;
    const int L = offsets[b + 1] - offsets[b];
    const bool keep = t < L;
    const bf16_t* s = src + (int64_t)(offsets[b] + t) * C;
    bf16_t* d = dst + (int64_t)row * C;
    const int nchunk = C >> 3;
    for (int c = lane; c < nchunk; c += 64) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (keep) v = *reinterpret_cast<const u32x4*>(s + c * 8);
        *reinterpret_cast<u32x4*>(d + c * 8) = v;
    }
    if (lane == 0) {
        if (mask) mask[row] = keep ? 1 : 0;
        // (1 - mask) * -10000 evaluated in bf16 as the reference does: bf16(-10000) = -9984
        if (key_bias) key_bias[row] = keep ? 0.0f : rbf(-10000.0f);
        if (kv_len && t == 0) kv_len[b] = L < T ? L : T;
    }
}

// the same without padding rows: row r of dst is source row r (r < offsets[B]) or zeros (r < rows_padded)
__global__ void pack_mask_kernel(int B, int T, int C, int rows_padded, const bf16_t* src, const int* offsets, bf16_t* dst,
                                 int64_t* mask, float* key_bias, int* kv_len) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row < rows_padded) {
        const bool keep = row < offsets[B];
        const bf16_t* s = src + (int64_t)row * C;
        bf16_t* d = dst + (int64_t)row * C;
        const int nchunk = C >> 3;
        for (int c = lane; c < nchunk; c += 64) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (keep) v = *reinterpret_cast<const u32x4*>(s + c * 8);
            *reinterpret_cast<u32x4*>(d + c * 8) = v;
        }
    }
    if (row < B * T && lane == 0) {
        const int b = row / T, t = row % T;
        const int L = offsets[b + 1] - offsets[b];
        const bool keep = t < L;
        if (mask) mask[row] = keep ? 1 : 0;
        if (key_bias) key_bias[row] = keep ? 0.0f : rbf(-10000.0f);
        if (kv_len && t == 0) kv_len[b] = L < T ? L : T;
    }
}

// noisy = (1 - s) * x + s * n (each op rounded to bf16); target = n - x
__global__ void flow_mix_kernel(int B, int64_t per, const bf16_t* x, const bf16_t* nz, const bf16_t* sigma, bf16_t* noisy,
                                bf16_t* target) {
    const int64_t total = (int64_t)B * per;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const float s = bf2f(sigma[i / per]);
        const float xv = bf2f(x[i]), nv = bf2f(nz[i]);
        const float oms = rbf(1.0f - s);
        noisy[i] = f2bf(rbf(oms * xv) + rbf(s * nv));
        target[i] = f2bf(nv - xv);
    }
}

// pass 1: per-block partial sums of (p - t)^2 and gradient; pass 2 (1 block): final mean
__global__ void mse_kernel(int64_t n, const bf16_t* pred, const bf16_t* target, float gcoef, float* partial,
                           bf16_t* dpred) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = bf2f(pred[i]) - bf2f(target[i]);
        s += d * d;
        if (dpred) dpred[i] = f2bf(d * gcoef);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void mse_final_kernel(int nb, const float* partial, float inv_n, float* loss) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 64) s += partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) loss[0] = s * inv_n;
}

// Counter-based dropout (LoRA's nn.Dropout on the adapter input, peft lora/layer.py [RECALL]): the keep decision of element i
// is a hash of (seed, i) -- splitmix64 -- so the backward regenerates the mask instead of storing it.  torch's arithmetic:
// out = x * mask * (1 / (1 - p)) evaluated in fp32, rounded once.
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t i, uint32_t thresh24) {
    uint64_t z = i + seed * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 40) >= thresh24;                    // 24 uniform bits against p * 2^24
}
// MODE 0: y = dropout(x);  MODE 1: io = bf16(io + dropout_mask(g))  (the gradient through the same mask)
template <int MODE>
__global__ void dropout_kernel(int64_t n, uint64_t seed, uint32_t thresh24, float inv_keep, const bf16_t* x, bf16_t* io) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = dropout_keep(seed, (uint64_t)i, thresh24) ? rbf(bf2f(x[i]) * inv_keep) : 0.f;
        io[i] = f2bf(MODE == 0 ? v : v + bf2f(io[i]));
    }
}

}  // namespace

extern "C" {

int yat_dropout(int64_t n, float p, uint64_t seed, int backward_add, const void* x, void* io, yat_stream_t stream) {
    if (n <= 0 || !(p >= 0.f && p < 1.f) || !x || !io) return YAT_EINVAL;
    const uint32_t thresh = (uint32_t)(p * 16777216.0f);
    const float inv_keep = 1.0f / (1.0f - p);
    if (backward_add)
        hipLaunchKernelGGL((dropout_kernel<1>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, seed, thresh, inv_keep,
                           (const bf16_t*)x, (bf16_t*)io);
    else
        hipLaunchKernelGGL((dropout_kernel<0>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, seed, thresh, inv_keep,
                           (const bf16_t*)x, (bf16_t*)io);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

int yat_act_fwd(int64_t n, int act, const void* x, void* y, yat_stream_t stream) {
    if (n <= 0 || (act != 1 && act != 2) || !x || !y) return YAT_EINVAL;
    hipLaunchKernelGGL((ew_kernel<0>), dim3(grid_for(n >> 3)), dim3(256), 0, (hipStream_t)stream, n, act,
                       (const bf16_t*)x, (const bf16_t*)nullptr, (bf16_t*)y);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}
int yat_act_bwd(int64_t n, int act, const void* x, const void* dy, void* dx, yat_stream_t stream) {
    if (n <= 0 || (act != 1 && act != 2) || !x || !dy || !dx) return YAT_EINVAL;
    hipLaunchKernelGGL((ew_kernel<1>), dim3(grid_for(n >> 3)), dim3(256), 0, (hipStream_t)stream, n, act,
                       (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}
int yat_add_bf16(int64_t n, const void* a, const void* b, void* out, yat_stream_t stream) {
    if (n <= 0 || !a || !b || !out) return YAT_EINVAL;
    hipLaunchKernelGGL((ew_kernel<2>), dim3(grid_for(n >> 3)), dim3(256), 0, (hipStream_t)stream, n, 0,
                       (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}
int yat_f32_to_bf16(int64_t n, const float* x, void* y, yat_stream_t stream) {
    if (n <= 0 || !x || !y) return YAT_EINVAL;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, x, (bf16_t*)y);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}
int yat_memset_zero(void* ptr, uint64_t nbytes, yat_stream_t stream) {
    if (!ptr || !nbytes) return YAT_EINVAL;
    const hipError_t e = hipMemsetAsync(ptr, 0, nbytes, (hipStream_t)stream);
    return e == hipSuccess ? YAT_OK : (int)e;
}
int yat_transpose_bf16(int B, int R, int Cc, const void* in, void* out, yat_stream_t stream) {
    if (B <= 0 || R <= 0 || Cc <= 0 || B > 65535 || !in || !out || in == out) return YAT_EINVAL;
    hipLaunchKernelGGL(transpose_kernel, dim3((Cc + 31) / 32, (R + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, R, Cc,
                       (const bf16_t*)in, (bf16_t*)out);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}
int yat_timestep_embed_fwd(int B, int dim, const float* t, void* out, yat_stream_t stream) {
    if (B <= 0 || dim <= 0 || (dim & 1) || !t || !out) return YAT_EINVAL;
    hipLaunchKernelGGL(timestep_embed_kernel, dim3((B * (dim / 2) + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, dim,
                       t, (bf16_t*)out);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}
int yat_pad_mask(int B, int T, int C, const void* src, const int* offsets, void* dst, int64_t* mask, float* key_bias,
                 int* kv_len, yat_stream_t stream) {
    if (B <= 0 || T <= 0 || C <= 0 || (C & 7) || !src || !offsets || !dst) return YAT_EINVAL;
    hipLaunchKernelGGL(pad_mask_kernel, dim3((B * T + 3) / 4), dim3(256), 0, (hipStream_t)stream, B, T, C,
                       (const bf16_t*)src, offsets, (bf16_t*)dst, mask, key_bias, kv_len);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}
int yat_pack_mask(int B, int T, int C, int rows_padded, const void* src, const int* offsets, void* dst, int64_t* mask,
                  float* key_bias, int* kv_len, yat_stream_t stream) {
    if (B <= 0 || T <= 0 || C <= 0 || (C & 7) || rows_padded <= 0 || !src || !offsets || !dst) return YAT_EINVAL;
    const int rows = rows_padded > B * T ? rows_padded : B * T;
    hipLaunchKernelGGL(pack_mask_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, B, T, C, rows_padded,
                       (const bf16_t*)src, offsets, (bf16_t*)dst, mask, key_bias, kv_len);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}
int yat_flow_mix(int B, int64_t per_sample, const void* x, const void* noise, const void* sigma, void* noisy,
                 void* target, yat_stream_t stream) {
    if (B <= 0 || per_sample <= 0 || !x || !noise || !sigma || !noisy || !target) return YAT_EINVAL;
    hipLaunchKernelGGL(flow_mix_kernel, dim3(grid_for(B * per_sample)), dim3(256), 0, (hipStream_t)stream, B, per_sample,
                       (const bf16_t*)x, (const bf16_t*)noise, (const bf16_t*)sigma, (bf16_t*)noisy, (bf16_t*)target);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}
int yat_mse_fwd_bwd(int64_t n, const void* pred, const void* target, float gscale, float* loss, void* dpred,
                    float* workspace_256, yat_stream_t stream) {
    if (n <= 0 || !pred || !target || !loss || !workspace_256) return YAT_EINVAL;
    int64_t nb64 = (n + 255) / 256;
    const int nb = (int)(nb64 > 256 ? 256 : nb64);
    hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, n, (const bf16_t*)pred,
                       (const bf16_t*)target, 2.0f * gscale / (float)n, workspace_256, (bf16_t*)dpred);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, nb, workspace_256, 1.0f / (float)n,
                       loss);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

}  // extern "C"
