// PixArt-Sigma recipe / embedding kernels for gfx950 (BASELINE config 3; /root/reference/train_pixart_sigma.py:151-185 and
// /root/reference/utils/patch_pixart_sigma_transformer.py:124-198).  All HBM-bound glue around the GEMM / attention / norm
// kernels the SANA path already has: consecutive lanes walk the contiguous axis of the WRITTEN tensor.
//   yat_patch_rearrange : NCHW latents <-> token rows of p x p patches (PatchEmbed's Conv2d(k=p, s=p) as a GEMM over
//                         [B*N, C*p*p] rows :130; unpatchify "nhwpqc->nchpwq" :186-191 and its backward)
//   yat_add_pos_embed   : x = bf16(x + pos)  with the fp32 2-D sin-cos table [N, D] ( [RECALL diffusers PatchEmbed] )
//   yat_ddpm_add_noise  : noisy = bf16(bf16(a_b * x) + bf16(c_b * n))   (DDPMScheduler.add_noise, train_pixart_sigma.py:176)
//   yat_mse_bf16_chunk  : MSELoss()(pred.chunk(2, 1)[0], noise) evaluated in bf16 as :180-184 does, with dL/dpred
#include "common.hpp"
#include "../../include/yat_hip.h"

namespace {

inline unsigned grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// token column order: channel_major ? c*p*p + pi*p + pj  (Conv2d weight [D, C, p, p] flattened)
//                                   : (pi*p + pj)*C + c  (proj_out rows, "nhwpqc")
template <bool TO_TOKENS>
__global__ void patch_kernel(int B, int C, int H, int W, int p, int channel_major, const bf16_t* src, bf16_t* dst) {
    const int hh = H / p, ww = W / p, pp = p * p, cols = C * pp;
    const int64_t total = (int64_t)B * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int b, c, y, x, col;
        int64_t tok;
        if (TO_TOKENS) {                         // i indexes the token matrix
            col = (int)(i % cols);
            tok = i / cols;
            const int n = (int)(tok % (hh * ww));
            b = (int)(tok / (hh * ww));
            int pi, pj;
            if (channel_major) { c = col / pp; pi = (col % pp) / p; pj = col % p; }
            else { c = col % C; pi = (col / C) / p; pj = (col / C) % p; }
            y = (n / ww) * p + pi;
            x = (n % ww) * p + pj;
            dst[i] = src[(((int64_t)b * C + c) * H + y) * W + x];
        } else {                                 // i indexes the NCHW image
            x = (int)(i % W);
            y = (int)((i / W) % H);
            c = (int)((i / ((int64_t)W * H)) % C);
            b = (int)(i / ((int64_t)W * H * C));
            const int pi = y % p, pj = x % p;
            col = channel_major ? c * pp + pi * p + pj : (pi * p + pj) * C + c;
            tok = ((int64_t)b * hh + y / p) * ww + x / p;
            dst[i] = src[tok * cols + col];
        }
    }
}

__global__ void add_pos_kernel(int64_t rows, int N, int D, const bf16_t* x, const float* pos, bf16_t* out) {
    const int dv = D >> 3;
    const int64_t nvec = rows * dv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / dv;
        const int c = (int)(i - r * dv) * 8;
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + r * D + c), v);
        const float* pr = pos + (int64_t)(r % N) * D + c;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += pr[e];
        *reinterpret_cast<u32x4*>(out + r * D + c) = pack8(v);
    }
}

__global__ void ddpm_mix_kernel(int B, int64_t per, const bf16_t* x, const bf16_t* nz, const bf16_t* ca, const bf16_t* cb,
                                bf16_t* noisy) {
    const int64_t total = (int64_t)B * per;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        noisy[i] = f2bf(rbf(bf2f(ca[b]) * bf2f(x[i])) + rbf(bf2f(cb[b]) * bf2f(nz[i])));
    }
}

// pred [B, stride] (the first `used` elements of every sample are the noise prediction), target [B, used] contiguous.
// torch's bf16 MSELoss: diff, diff*diff and the mean each round to bf16 (fp32 accumulation inside the mean); backward
// norm * (a - b) * grad with every factor and product in bf16.
__global__ void mse_bf16_kernel(int B, int64_t used, int64_t stride, const bf16_t* pred, const bf16_t* target, float norm_b,
                                float g_b, float* partial, bf16_t* dpred) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t total = (int64_t)B * stride;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / stride, j = i - b * stride;
        if (j < used) {
            const float d = rbf(bf2f(pred[i]) - bf2f(target[b * used + j]));
            s += rbf(d * d);
            if (dpred) dpred[i] = f2bf(rbf(norm_b * d) * g_b);
        } else if (dpred) {
            dpred[i] = 0;                        // the learned-sigma half is dropped by .chunk(2, 1)[0]: no gradient
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void mse_bf16_final_kernel(int nb, const float* partial, float inv_n, float* loss) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 64) s += partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) loss[0] = rbf(s * inv_n);
}

}  // namespace

extern "C" {

int yat_patch_rearrange(int B, int C, int H, int W, int p, int channel_major, int to_tokens, const void* src, void* dst,
                        yat_stream_t stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || p <= 0 || (H % p) || (W % p) || !src || !dst || src == dst) return YAT_EINVAL;
    const int64_t total = (int64_t)B * C * H * W;
    if (to_tokens)
        hipLaunchKernelGGL((patch_kernel<true>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, B, C, H, W, p,
                           channel_major, (const bf16_t*)src, (bf16_t*)dst);
    else
        hipLaunchKernelGGL((patch_kernel<false>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, B, C, H, W, p,
                           channel_major, (const bf16_t*)src, (bf16_t*)dst);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

int yat_add_pos_embed(int64_t rows, int N, int D, const void* x, const float* pos, void* out, yat_stream_t stream) {
    if (rows <= 0 || N <= 0 || D <= 0 || (D & 7) || !x || !pos || !out) return YAT_EINVAL;
    hipLaunchKernelGGL(add_pos_kernel, dim3(grid_for(rows * (D >> 3))), dim3(256), 0, (hipStream_t)stream, rows, N, D,
                       (const bf16_t*)x, pos, (bf16_t*)out);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

int yat_ddpm_add_noise(int B, int64_t per_sample, const void* x, const void* noise, const void* sqrt_alpha_prod,
                       const void* sqrt_one_minus_alpha_prod, void* noisy, yat_stream_t stream) {
    if (B <= 0 || per_sample <= 0 || !x || !noise || !sqrt_alpha_prod || !sqrt_one_minus_alpha_prod || !noisy)
        return YAT_EINVAL;
    hipLaunchKernelGGL(ddpm_mix_kernel, dim3(grid_for(B * per_sample)), dim3(256), 0, (hipStream_t)stream, B, per_sample,
                       (const bf16_t*)x, (const bf16_t*)noise, (const bf16_t*)sqrt_alpha_prod,
                       (const bf16_t*)sqrt_one_minus_alpha_prod, (bf16_t*)noisy);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

int yat_mse_bf16_chunk(int B, int64_t used, int64_t stride, const void* pred, const void* target, float gscale, float* loss,
                       void* dpred, float* workspace_256, yat_stream_t stream) {
    if (B <= 0 || used <= 0 || stride < used || !pred || !target || !loss || !workspace_256) return YAT_EINVAL;
    const int64_t total = (int64_t)B * stride, n = (int64_t)B * used;
    int64_t nb64 = (total + 255) / 256;
    const int nb = (int)(nb64 > 256 ? 256 : nb64);
    // bf16(2 / n) and bf16(upstream gradient) on the host: both are scalars of the reference's bf16 graph
    auto round_bf16 = [](float f) {
        union { float f; uint32_t u; } v;
        v.f = f;
        v.u = (v.u + 0x7fffu + ((v.u >> 16) & 1u)) & 0xffff0000u;
        return v.f;
    };
    hipLaunchKernelGGL(mse_bf16_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, B, used, stride, (const bf16_t*)pred,
                       (const bf16_t*)target, round_bf16(2.0f / (float)n), round_bf16(gscale), workspace_256, (bf16_t*)dpred);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(mse_bf16_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, nb, workspace_256, 1.0f / (float)n,
                       loss);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

}  // extern "C"
