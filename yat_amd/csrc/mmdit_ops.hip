// MMDiT (SD3.5, BASELINE config 4) glue around the joint attention of JointTransformerBlock
// (the reference trains diffusers' SD3Transformer2DModel: /root/reference/train_sd35.py:4,188-191; block internals [RECALL],
// restated in oracle/sd3_ref.py):
//   * per-head RMSNorm(head_dim, eps, affine) on q and k of the image stream (attn.norm_q / norm_k) and of the text stream
//     (attn.norm_added_q / norm_added_k), fused with the row concatenation [image tokens | text tokens] that
//     JointAttnProcessor2_0 performs with torch.cat(dim=2) -- and their backward (split + RMSNorm backward + the four
//     weight gradients);
//   * the split of the joint attention output back into the two streams, and its inverse for the output gradient.
// All HBM-bound streaming passes: 16-byte accesses, one workgroup walks rows with a FIXED lane -> column assignment, so a
// lane's norm weights stay in registers and the weight-gradient partials accumulate in registers across rows; reductions
// have a fixed order (no atomics -> bit-reproducible).
#include "common.hpp"

namespace {

constexpr int THREADS = 256;

struct JointP {
    int B, N, T, H, dh, D;       // tokens per image: N image + T text; D = H*dh
    const bf16_t* img; int ld_img;   // [B*N, >= 3D]  q | k | v projections of the image stream
    const bf16_t* txt; int ld_txt;   // [B*T, >= 3D]  ... of the text stream (T may be 0)
    const bf16_t *wq_img, *wk_img, *wq_txt, *wk_txt;     // [dh] each
    bf16_t* joint; int ld_joint;     // [B*(N+T), >= 3D]
    float* rstd;                     // [B*(N+T), 2H]
    float eps;
};

// source row of joint row `r`: image rows first, then the text rows of the same image
__device__ __forceinline__ const bf16_t* src_row(const JointP& p, int64_t r, bool& is_txt) {
    const int L = p.N + p.T;
    const int b = (int)(r / L), pos = (int)(r - (int64_t)b * L);
    is_txt = pos >= p.N;
    return is_txt ? p.txt + ((int64_t)b * p.T + (pos - p.N)) * p.ld_txt : p.img + ((int64_t)b * p.N + pos) * p.ld_img;
}

__global__ __launch_bounds__(THREADS) void qknorm_concat_fwd_kernel(JointP p) {
    const int lph = p.dh >> 3;                    // lanes per head (16-byte chunks per head): 4, 8 or 16
    const int cq = p.D >> 3;                      // chunks per section
    const int nqk = 2 * cq;
    const int64_t rows = (int64_t)p.B * (p.N + p.T);
    const int tid = threadIdx.x;
    // this lane's q|k chunks: c = tid + 256*j; channel offset inside the head is the same for every j (256 % lph == 0)
    const int ch0 = (tid % lph) * 8;
    float wi[2][8], wt[2][8] = {};                // [q|k][8 channels] for the image / text stream
    unpack8(*reinterpret_cast<const u32x4*>(p.wq_img + ch0), wi[0]);
    unpack8(*reinterpret_cast<const u32x4*>(p.wk_img + ch0), wi[1]);
    if (p.T) {
        unpack8(*reinterpret_cast<const u32x4*>(p.wq_txt + ch0), wt[0]);
        unpack8(*reinterpret_cast<const u32x4*>(p.wk_txt + ch0), wt[1]);
    }
    const float inv_dh = 1.0f / (float)p.dh;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        bool is_txt;
        const bf16_t* src = src_row(p, r, is_txt);
        bf16_t* dst = p.joint + r * p.ld_joint;
        for (int c = tid; c < nqk; c += THREADS) {
            float x[8];
            unpack8(*reinterpret_cast<const u32x4*>(src + c * 8), x);
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
            for (int m = lph >> 1; m; m >>= 1) ss += __shfl_xor(ss, m, 64);
            const float rs = rsqrtf(ss * inv_dh + p.eps);
            const int sec = c >= cq;
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {       // (selects, not a pointer into the register arrays: that would be scratch)
                const float w = is_txt ? (sec ? wt[1][e] : wt[0][e]) : (sec ? wi[1][e] : wi[0][e]);
                y[e] = rbf(rbf(x[e] * rs) * w);                                 // RMSNorm: fp32 normalise -> bf16 -> * weight
            }
            *reinterpret_cast<u32x4*>(dst + c * 8) = pack8(y);
            if ((tid % lph) == 0) p.rstd[r * (2 * p.H) + c / lph] = rs;
        }
        for (int c = tid; c < cq; c += THREADS)                                  // v: plain copy
            *reinterpret_cast<u32x4*>(dst + 2 * p.D + c * 8) = *reinterpret_cast<const u32x4*>(src + 2 * p.D + c * 8);
    }
}

struct JointBwdP {
    JointP f;                        // forward operands (joint unused)
    const bf16_t* dj; int ld_dj;     // [B*(N+T), >= 3D]  gradients of the normalised q | k and of v (joint rows)
    bf16_t* dimg; int ld_dimg;       // [B*N, >= 3D]
    bf16_t* dtxt; int ld_dtxt;       // [B*T, >= 3D]
    float* partial;                  // [gridDim.x][2 streams][2 (q|k)][dh]
};

__global__ __launch_bounds__(THREADS) void qknorm_concat_bwd_kernel(JointBwdP q) {
    const JointP& p = q.f;
    const int lph = p.dh >> 3, cq = p.D >> 3, nqk = 2 * cq;
    const int L = p.N + p.T;
    const int tid = threadIdx.x;
    const int ch0 = (tid % lph) * 8;
    const float inv_dh = 1.0f / (float)p.dh;
    __shared__ float red[2][THREADS][8];
    // two phases with the same code: image rows (stream 0), then text rows (stream 1) -- a lane's weights and its
    // weight-gradient accumulators belong to one stream at a time
    for (int st = 0; st < (p.T ? 2 : 1); ++st) {
        float w[2][8], acc[2][8];
        unpack8(*reinterpret_cast<const u32x4*>((st ? p.wq_txt : p.wq_img) + ch0), w[0]);
        unpack8(*reinterpret_cast<const u32x4*>((st ? p.wk_txt : p.wk_img) + ch0), w[1]);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[0][e] = acc[1][e] = 0.f;
        const int per = st ? p.T : p.N;
        const int64_t srows = (int64_t)p.B * per;
        for (int64_t sr = blockIdx.x; sr < srows; sr += gridDim.x) {
            const int b = (int)(sr / per), pos = (int)(sr - (int64_t)b * per);
            const int64_t jr = (int64_t)b * L + (st ? p.N + pos : pos);
            const bf16_t* x = st ? p.txt + sr * p.ld_txt : p.img + sr * p.ld_img;
            const bf16_t* dy = q.dj + jr * q.ld_dj;
            bf16_t* dx = st ? q.dtxt + sr * q.ld_dtxt : q.dimg + sr * q.ld_dimg;
            for (int c = tid; c < nqk; c += THREADS) {
                const int sec = c >= cq;
                float xv[8], g[8];
                unpack8(*reinterpret_cast<const u32x4*>(x + c * 8), xv);
                unpack8(*reinterpret_cast<const u32x4*>(dy + c * 8), g);
                const float rs = p.rstd[jr * (2 * p.H) + c / lph];
                float dot = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = xv[e] * rs;
                    acc[sec][e] += g[e] * rbf(xh);                 // d_w: dy * (the bf16 x_hat the forward multiplied by w)
                    g[e] *= w[sec][e];                             // d_xhat
                    dot += g[e] * xh;
                    xv[e] = xh;
                }
                for (int m = lph >> 1; m; m >>= 1) dot += __shfl_xor(dot, m, 64);
                dot *= inv_dh;
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rs * (g[e] - xv[e] * dot);
                *reinterpret_cast<u32x4*>(dx + c * 8) = pack8(o);
            }
            for (int c = tid; c < cq; c += THREADS)                // d_v: plain copy
                *reinterpret_cast<u32x4*>(dx + 2 * p.D + c * 8) = *reinterpret_cast<const u32x4*>(dy + 2 * p.D + c * 8);
        }
        // workgroup reduction of the weight-gradient partials, fixed order: lanes with the same tid % lph share channels
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[0][tid][e] = acc[0][e]; red[1][tid][e] = acc[1][e]; }
        __syncthreads();
        if (tid < 2 * p.dh) {
            const int sec = tid / p.dh, ch = tid - sec * p.dh;
            float s = 0.f;
            for (int t = ch >> 3; t < THREADS; t += lph) s += red[sec][t][ch & 7];
            q.partial[(((int64_t)blockIdx.x * 2 + st) * 2 + sec) * p.dh + ch] = s;
        }
    }
}

// dw[stream][q|k][dh] (+)= sum over workgroups.  One workgroup per (stream, q|k): lane group g = tid / dh sums workgroups
// g, g + G, ... of its channel, the G group sums are added in a fixed order (bit-reproducible).
__global__ __launch_bounds__(THREADS) void qknorm_dw_kernel(int nblk, int dh, const float* partial, bf16_t* dwq_img,
                                                            bf16_t* dwk_img, bf16_t* dwq_txt, bf16_t* dwk_txt, int accumulate) {
    const int st = blockIdx.x >> 1, sec = blockIdx.x & 1;
    const int G = THREADS / dh, g = threadIdx.x / dh, ch = threadIdx.x - g * dh;
    __shared__ float red[THREADS];
    float s = 0.f;
    for (int b = g; b < nblk; b += G) s += partial[(((int64_t)b * 2 + st) * 2 + sec) * dh + ch];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < dh) {
        float t = 0.f;
        for (int k = 0; k < G; ++k) t += red[k * dh + threadIdx.x];
        bf16_t* out = st ? (sec ? dwk_txt : dwq_txt) : (sec ? dwk_img : dwq_img);
        if (accumulate) t += bf2f(out[threadIdx.x]);
        out[threadIdx.x] = f2bf(t);
    }
}

// rows of width C (16-byte chunks) between the joint layout and the two per-stream layouts
struct RowsP {
    int B, N, T, C;
    bf16_t* joint; int ld_joint;
    bf16_t* img; int ld_img;
    bf16_t* txt; int ld_txt;         // may be null: to_joint writes zeros into the text rows, from_joint skips them
    int to_joint;
};

__global__ __launch_bounds__(THREADS) void joint_rows_kernel(RowsP p) {
    const int L = p.N + p.T, cpr = p.C >> 3;
    const int64_t total = (int64_t)p.B * L * cpr;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * THREADS) {
        const int64_t r = i / cpr;
        const int c = (int)(i - r * cpr);
        const int b = (int)(r / L), pos = (int)(r - (int64_t)b * L);
        bf16_t* j = p.joint + r * p.ld_joint + c * 8;
        bf16_t* s = pos < p.N ? p.img + ((int64_t)b * p.N + pos) * p.ld_img + c * 8
                              : (p.txt ? p.txt + ((int64_t)b * p.T + (pos - p.N)) * p.ld_txt + c * 8 : nullptr);
        if (p.to_joint) *reinterpret_cast<u32x4*>(j) = s ? *reinterpret_cast<const u32x4*>(s) : u32x4{0u, 0u, 0u, 0u};
        else if (s) *reinterpret_cast<u32x4*>(s) = *reinterpret_cast<const u32x4*>(j);
    }
}

bool joint_args_ok(int B, int N, int T, int H, int dh, int ld_img, int ld_txt, int ld_joint) {
    if (B <= 0 || N <= 0 || T < 0 || H <= 0) return false;
    if (dh != 32 && dh != 64 && dh != 128) return false;
    const int D = H * dh;
    if (ld_img < 3 * D || (ld_img & 7) || ld_joint < 3 * D || (ld_joint & 7)) return false;
    if (T && (ld_txt < 3 * D || (ld_txt & 7))) return false;
    return true;
}

int grid_for_rows(int64_t rows) { return (int)(rows < 1024 ? rows : 1024); }

}  // namespace

extern "C" {

int yat_qknorm_concat_fwd(int B, int N, int T, int H, int dh, float eps, const void* qkv_img, int ld_img, const void* qkv_txt,
                          int ld_txt, const void* wq_img, const void* wk_img, const void* wq_txt, const void* wk_txt,
                          void* joint, int ld_joint, float* rstd, yat_stream_t stream) {
    if (!joint_args_ok(B, N, T, H, dh, ld_img, ld_txt, ld_joint)) return YAT_EINVAL;
    if (!qkv_img || !wq_img || !wk_img || !joint || !rstd || (T && (!qkv_txt || !wq_txt || !wk_txt))) return YAT_EINVAL;
    JointP p{B, N, T, H, dh, H * dh, (const bf16_t*)qkv_img, ld_img, (const bf16_t*)qkv_txt, ld_txt, (const bf16_t*)wq_img,
             (const bf16_t*)wk_img, (const bf16_t*)wq_txt, (const bf16_t*)wk_txt, (bf16_t*)joint, ld_joint, rstd, eps};
    hipLaunchKernelGGL(qknorm_concat_fwd_kernel, dim3(grid_for_rows((int64_t)B * (N + T))), dim3(THREADS), 0,
                       (hipStream_t)stream, p);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

uint64_t yat_qknorm_concat_bwd_workspace_bytes(int B, int N, int T, int dh) {
    return (uint64_t)grid_for_rows((int64_t)B * (N > T ? N : T)) * 2 * 2 * dh * sizeof(float);
}

int yat_qknorm_concat_bwd(int B, int N, int T, int H, int dh, const void* qkv_img, int ld_img, const void* qkv_txt, int ld_txt,
                          const void* wq_img, const void* wk_img, const void* wq_txt, const void* wk_txt, const float* rstd,
                          const void* d_joint, int ld_dj, void* dqkv_img, int ld_dimg, void* dqkv_txt, int ld_dtxt,
                          void* dwq_img, void* dwk_img, void* dwq_txt, void* dwk_txt, int accumulate_dw, void* workspace,
                          yat_stream_t stream) {
    if (!joint_args_ok(B, N, T, H, dh, ld_img, ld_txt, ld_dj)) return YAT_EINVAL;
    const int D = H * dh;
    if (!qkv_img || !wq_img || !wk_img || !rstd || !d_joint || !dqkv_img || !dwq_img || !dwk_img || !workspace)
        return YAT_EINVAL;
    if (ld_dimg < 3 * D || (ld_dimg & 7)) return YAT_EINVAL;
    if (T && (!qkv_txt || !wq_txt || !wk_txt || !dqkv_txt || !dwq_txt || !dwk_txt || ld_dtxt < 3 * D || (ld_dtxt & 7)))
        return YAT_EINVAL;
    JointBwdP q;
    q.f = JointP{B, N, T, H, dh, D, (const bf16_t*)qkv_img, ld_img, (const bf16_t*)qkv_txt, ld_txt, (const bf16_t*)wq_img,
                 (const bf16_t*)wk_img, (const bf16_t*)wq_txt, (const bf16_t*)wk_txt, nullptr, 0, (float*)rstd, 0.f};
    q.dj = (const bf16_t*)d_joint; q.ld_dj = ld_dj;
    q.dimg = (bf16_t*)dqkv_img; q.ld_dimg = ld_dimg;
    q.dtxt = (bf16_t*)dqkv_txt; q.ld_dtxt = ld_dtxt;
    q.partial = (float*)workspace;
    const int nblk = grid_for_rows((int64_t)B * (N > T ? N : T));
    hipLaunchKernelGGL(qknorm_concat_bwd_kernel, dim3(nblk), dim3(THREADS), 0, (hipStream_t)stream, q);
    YAT_CHECK_LAUNCH();
    const int nst = T ? 2 : 1;
    hipLaunchKernelGGL(qknorm_dw_kernel, dim3(nst * 2), dim3(THREADS), 0, (hipStream_t)stream, nblk, dh,
                       (const float*)workspace, (bf16_t*)dwq_img, (bf16_t*)dwk_img, (bf16_t*)dwq_txt, (bf16_t*)dwk_txt,
                       accumulate_dw);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

int yat_joint_rows(int B, int N, int T, int C, void* joint, int ld_joint, void* img, int ld_img, void* txt, int ld_txt,
                   int to_joint, yat_stream_t stream) {
    if (B <= 0 || N <= 0 || T < 0 || C <= 0 || (C & 7) || !joint || !img) return YAT_EINVAL;
    if (ld_joint < C || (ld_joint & 7) || ld_img < C || (ld_img & 7) || (txt && (ld_txt < C || (ld_txt & 7)))) return YAT_EINVAL;
    RowsP p{B, N, T, C, (bf16_t*)joint, ld_joint, (bf16_t*)img, ld_img, (bf16_t*)txt, ld_txt, to_joint};
    const int64_t total = (int64_t)B * (N + T) * (C >> 3);
    int64_t nb = (total + THREADS - 1) / THREADS;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(joint_rows_kernel, dim3((unsigned)nb), dim3(THREADS), 0, (hipStream_t)stream, p);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

}  // extern "C"
