// Fused optimizer step for gfx950 over ONE flat bf16 parameter buffer:
//   global gradient-norm clip (torch.nn.utils.clip_grad_norm_, common/trainer.py:347)
//   + AdamW (torch.optim.AdamW, common/trainer.py:246-248,348) + zero_grad (:356) + optional EMA (:350-351).
// HBM-bound: 14 B/param algorithmic traffic (read p,g,m,v; write p,m,v) + 2 B for the grad clear.
// The arithmetic follows torch's single-tensor op sequence on bf16 tensors: every torch op
// computes in fp32 and rounds its result to bf16, so the kernel rounds at the same points.
#include "common.hpp"
#include "../../include/yat_hip.h"
#include <math.h>

namespace {

constexpr int64_t NORM_CHUNK = 1 << 18;  // elements per gradnorm workgroup

// grid = (maxchunks, nseg): sum of squares of one chunk of one parameter tensor
__global__ __launch_bounds__(256) void gradnorm_partial_kernel(const bf16_t* g, const int64_t* seg_start, int maxchunks,
                                                               float* partial) {
    const int seg = blockIdx.y, ck = blockIdx.x;
    const int64_t s0 = seg_start[seg], s1 = seg_start[seg + 1];
    const int64_t c0 = s0 + (int64_t)ck * NORM_CHUNK;
    if (c0 >= s1) return;
    const int64_t c1 = (c0 + NORM_CHUNK < s1) ? c0 + NORM_CHUNK : s1;
    float s = 0.f;
    // segment starts are 16-B aligned; vector body + scalar tail
    const int64_t nvec = (c1 - c0) >> 3;
    for (int64_t i = threadIdx.x; i < nvec; i += 256) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(g + c0 + i * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e] * v[e];
    }
    for (int64_t i = c0 + (nvec << 3) + threadIdx.x; i < c1; i += 256) { const float v = bf2f(g[i]); s += v * v; }
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)seg * maxchunks + ck] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void gradnorm_final_kernel(int nseg, const int64_t* seg_start, int maxchunks,
                                                             const float* partial, float max_norm, float* norm_out,
                                                             float* clip_coef) {
    float acc = 0.f;
    for (int seg = threadIdx.x; seg < nseg; seg += 256) {
        const int64_t len = seg_start[seg + 1] - seg_start[seg];
        const int nck = (int)((len + NORM_CHUNK - 1) / NORM_CHUNK);
        float s = 0.f;
        for (int c = 0; c < nck; ++c) s += partial[(int64_t)seg * maxchunks + c];
        const float nb = rbf(sqrtf(s));   // per-tensor norm comes back as a bf16 tensor
        acc += nb * nb;
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float total = rbf(sqrtf(red[0] + red[1] + red[2] + red[3]));
        float coef = rbf(max_norm / rbf(total + 1e-6f));
        coef = fminf(coef, 1.0f);
        norm_out[0] = total;
        clip_coef[0] = coef;
    }
}

// ---- the same norm over PIECES (round 6): a piece is a tensor, or the part of a tensor inside one eighth of its
// data-parallel bucket; partial sums live compactly at chunk_base[piece] + chunk.  A tensor's sum of squares is the sum over its
// pieces, in order, of the sum over each piece's chunks, in order -- the same numbers whoever computed each piece, which is
// what lets N ranks each sum only the pieces they hold reduced gradients for (``owned``), add the arrays up (every slot has
// exactly one non-zero contributor) and arrive at the clip coefficient of the one-rank run BIT FOR BIT.
__global__ __launch_bounds__(256) void gradnorm_pieces_partial_kernel(const bf16_t* g, const int64_t* piece_start,
                                                                      const int* chunk_base, const unsigned char* owned,
                                                                      float* partial) {
    const int pc = blockIdx.y, ck = blockIdx.x;
    const int nck = chunk_base[pc + 1] - chunk_base[pc];
    if (ck >= nck) return;
    float* dst = partial + chunk_base[pc] + ck;
    if (owned && !owned[pc]) {
        if (threadIdx.x == 0) *dst = 0.f;
        return;
    }
    const int64_t s1 = piece_start[pc + 1];
    const int64_t c0 = piece_start[pc] + (int64_t)ck * NORM_CHUNK;
    const int64_t c1 = (c0 + NORM_CHUNK < s1) ? c0 + NORM_CHUNK : s1;
    float s = 0.f;
    const int64_t nvec = (c1 - c0) >> 3;
    for (int64_t i = threadIdx.x; i < nvec; i += 256) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(g + c0 + i * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e] * v[e];
    }
    for (int64_t i = c0 + (nvec << 3) + threadIdx.x; i < c1; i += 256) { const float v = bf2f(g[i]); s += v * v; }
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *dst = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void gradnorm_pieces_final_kernel(int ntensor, const int* tensor_first_piece,
                                                                    const int* chunk_base, const float* partial,
                                                                    float max_norm, float* norm_out, float* clip_coef) {
    float acc = 0.f;
    for (int t = threadIdx.x; t < ntensor; t += 256) {
        float s = 0.f;
        for (int pc = tensor_first_piece[t]; pc < tensor_first_piece[t + 1]; ++pc) {
            float sp = 0.f;
            for (int c = chunk_base[pc]; c < chunk_base[pc + 1]; ++c) sp += partial[c];
            s += sp;
        }
        const float nb = rbf(sqrtf(s));   // per-tensor norm comes back as a bf16 tensor
        acc += nb * nb;
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float total = rbf(sqrtf(red[0] + red[1] + red[2] + red[3]));
        float coef = rbf(max_norm / rbf(total + 1e-6f));
        coef = fminf(coef, 1.0f);
        norm_out[0] = total;
        clip_coef[0] = coef;
    }
}

struct AdamP {
    float wd_mul, w1, beta2, om_b2, bc2_sqrt, eps, neg_step_size, ema_omd;
    int use_wd, zero_grad;
};

// one parameter's update: torch's single-tensor AdamW op sequence with a bf16 rounding after every op
__device__ __forceinline__ void adamw_elem(float& pr, float g, float& mr_, float& vr_, float coef, bool clip, const AdamP& a) {
    const float gr = clip ? rbf(g * coef) : g;                      // grads.mul_(clip_coef_clamped)
    if (a.use_wd) pr = rbf(pr * a.wd_mul);                          // param.mul_(1 - lr*wd)
    const float mr = rbf(mr_ + a.w1 * (gr - mr_));                  // exp_avg.lerp_(grad, 1-beta1)
    float vr = rbf(vr_ * a.beta2);                                  // exp_avg_sq.mul_(beta2)
    vr = rbf(vr + a.om_b2 * gr * gr);                               //   .addcmul_(grad, grad, value=1-beta2)
    float den = rbf(sqrtf(vr));                                     // exp_avg_sq.sqrt()
    den = rbf(den / a.bc2_sqrt);                                    //   / bias_correction2_sqrt
    den = rbf(den + a.eps);                                         //   .add_(eps)
    pr = rbf(pr + a.neg_step_size * mr / den);                      // param.addcdiv_(exp_avg, denom, -step_size)
    mr_ = mr;
    vr_ = vr;
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void adamw_kernel(int64_t nvec, bf16_t* p, bf16_t* g, bf16_t* m, bf16_t* v,
                                                    const float* clip_coef, bf16_t* ema, AdamP a) {
    // U vectors of 8 parameters per thread and iteration, every load of the iteration issued before the first use (4 U
    // independent 16-byte loads in flight per lane); NT: the state streams through once per step -- non-temporal accesses
    // keep 22 GB of it from evicting what the forward that runs beside this kernel re-reads.
    const float coef = clip_coef ? clip_coef[0] : 1.0f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < nvec; i0 += stride * U) {
        u32x4 rp[U], rg[U], rm[U], rv[U], rs[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int64_t i = i0 + q * stride;
            if (i < nvec) {
                if (NT) {
                    rp[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + i * 8));
                    rg[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(g + i * 8));
                    rm[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(m + i * 8));
                    rv[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(v + i * 8));
                } else {
                    rp[q] = *reinterpret_cast<const u32x4*>(p + i * 8);
                    rg[q] = *reinterpret_cast<const u32x4*>(g + i * 8);
                    rm[q] = *reinterpret_cast<const u32x4*>(m + i * 8);
                    rv[q] = *reinterpret_cast<const u32x4*>(v + i * 8);
                }
                if (ema) rs[q] = *reinterpret_cast<const u32x4*>(ema + i * 8);
            }
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int64_t i = i0 + q * stride;
            if (i >= nvec) continue;
            float pp[8], gg[8], mm[8], vv[8], ss[8];
            unpack8(rp[q], pp);
            unpack8(rg[q], gg);
            unpack8(rm[q], mm);
            unpack8(rv[q], vv);
            if (ema) unpack8(rs[q], ss);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                adamw_elem(pp[e], gg[e], mm[e], vv[e], coef, clip_coef != nullptr, a);
                if (ema) {                                                  // s.sub_(one_minus_decay * (s - p))
                    const float d = rbf(ss[e] - pp[e]);
                    ss[e] = rbf(ss[e] - rbf(a.ema_omd * d));
                }
            }
            if (NT) {
                __builtin_nontemporal_store(pack8(pp), reinterpret_cast<u32x4*>(p + i * 8));
                __builtin_nontemporal_store(pack8(mm), reinterpret_cast<u32x4*>(m + i * 8));
                __builtin_nontemporal_store(pack8(vv), reinterpret_cast<u32x4*>(v + i * 8));
            } else {
                *reinterpret_cast<u32x4*>(p + i * 8) = pack8(pp);
                *reinterpret_cast<u32x4*>(m + i * 8) = pack8(mm);
                *reinterpret_cast<u32x4*>(v + i * 8) = pack8(vv);
            }
            if (ema) *reinterpret_cast<u32x4*>(ema + i * 8) = pack8(ss);
            if (a.zero_grad) *reinterpret_cast<u32x4*>(g + i * 8) = u32x4{0u, 0u, 0u, 0u};
        }
    }
}

// Background variant for the update that runs UNDER the next forward: one 256-thread workgroup per CU, 4 elements per
// thread and iteration, capped at 48 VGPRs -- what two resident waves of a 256-row GEMM workgroup (<= 232 VGPRs each)
// leave free on a SIMD, so it can share a CU with the GEMM instead of queueing for a whole one.  Same arithmetic.
__global__ __launch_bounds__(256, 10) void adamw_bg_kernel(int64_t nvec4, bf16_t* p, bf16_t* g, bf16_t* m, bf16_t* v,
                                                           const float* clip_coef, bf16_t* ema, AdamP a) {
    const float coef = clip_coef ? clip_coef[0] : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec4; i += (int64_t)gridDim.x * blockDim.x) {
        float pp[4], gg[4], mm[4], vv[4], ss[4];
        unpack4(*reinterpret_cast<const u32x2*>(p + i * 4), pp);
        unpack4(*reinterpret_cast<const u32x2*>(g + i * 4), gg);
        unpack4(*reinterpret_cast<const u32x2*>(m + i * 4), mm);
        unpack4(*reinterpret_cast<const u32x2*>(v + i * 4), vv);
        if (ema) unpack4(*reinterpret_cast<const u32x2*>(ema + i * 4), ss);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            adamw_elem(pp[e], gg[e], mm[e], vv[e], coef, clip_coef != nullptr, a);
            if (ema) {
                const float d = rbf(ss[e] - pp[e]);
                ss[e] = rbf(ss[e] - rbf(a.ema_omd * d));
            }
        }
        *reinterpret_cast<u32x2*>(p + i * 4) = pack4(pp[0], pp[1], pp[2], pp[3]);
        *reinterpret_cast<u32x2*>(m + i * 4) = pack4(mm[0], mm[1], mm[2], mm[3]);
        *reinterpret_cast<u32x2*>(v + i * 4) = pack4(vv[0], vv[1], vv[2], vv[3]);
        if (ema) *reinterpret_cast<u32x2*>(ema + i * 4) = pack4(ss[0], ss[1], ss[2], ss[3]);
        if (a.zero_grad) *reinterpret_cast<u32x2*>(g + i * 4) = u32x2{0u, 0u};
    }
}

}  // namespace

extern "C" {

static int norm_maxchunks(int64_t n) { return (int)((n + NORM_CHUNK - 1) / NORM_CHUNK); }

uint64_t yat_gradnorm_workspace_bytes(int64_t n, int nseg) {
    return (uint64_t)nseg * norm_maxchunks(n) * sizeof(float);
}

int yat_gradnorm_clip(int64_t n, const void* grad, int nseg, const int64_t* seg_start, float max_norm, float* norm_out,
                      float* clip_coef, void* workspace, yat_stream_t stream) {
    if (n <= 0 || nseg <= 0 || nseg > 65535 || !grad || !seg_start || !norm_out || !clip_coef || !workspace)
        return YAT_EINVAL;
    const int mc = norm_maxchunks(n);
    hipLaunchKernelGGL(gradnorm_partial_kernel, dim3(mc, nseg), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)grad,
                       seg_start, mc, (float*)workspace);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(gradnorm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, nseg, seg_start, mc,
                       (const float*)workspace, max_norm, norm_out, clip_coef);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

int yat_gradnorm_pieces_partial(const void* grad, int npiece, const int64_t* piece_start, const int* chunk_base,
                                int max_piece_chunks, const unsigned char* owned, float* partial, yat_stream_t stream) {
    if (!grad || npiece <= 0 || npiece > 65535 || !piece_start || !chunk_base || max_piece_chunks <= 0 || !partial)
        return YAT_EINVAL;
    hipLaunchKernelGGL(gradnorm_pieces_partial_kernel, dim3(max_piece_chunks, npiece), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)grad, piece_start, chunk_base, owned, partial);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

int yat_gradnorm_pieces_finish(int ntensor, const int* tensor_first_piece, const int* chunk_base, const float* partial,
                               float max_norm, float* norm_out, float* clip_coef, yat_stream_t stream) {
    if (ntensor <= 0 || !tensor_first_piece || !chunk_base || !partial || !norm_out || !clip_coef) return YAT_EINVAL;
    hipLaunchKernelGGL(gradnorm_pieces_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ntensor, tensor_first_piece,
                       chunk_base, partial, max_norm, norm_out, clip_coef);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

int yat_adamw_step(int64_t n, void* param, void* grad, void* exp_avg, void* exp_avg_sq, const float* clip_coef, double lr,
                   double beta1, double beta2, double eps, double weight_decay, int step, int zero_grad, void* ema_shadow,
                   double ema_decay, int background, yat_stream_t stream) {
    if (n <= 0 || (n & 7) || step < 1 || !param || !grad || !exp_avg || !exp_avg_sq) return YAT_EINVAL;
    // scalar prep in double exactly as torch's python does, then narrowed to the kernels' opmath (float)
    const double dlr = lr, db1 = beta1, db2 = beta2;
    AdamP a;
    a.use_wd = weight_decay != 0.0;
    a.wd_mul = (float)(1.0 - dlr * weight_decay);
    a.w1 = (float)(1.0 - db1);
    a.beta2 = (float)db2;
    a.om_b2 = (float)(1.0 - db2);
    const double bc1 = 1.0 - pow(db1, (double)step), bc2 = 1.0 - pow(db2, (double)step);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.neg_step_size = (float)(-(dlr / bc1));
    a.eps = (float)eps;
    a.ema_omd = (float)(1.0 - ema_decay);
    a.zero_grad = zero_grad;
    if (background) {
        const int64_t nvec4 = n >> 2;
        int64_t nbg = (nvec4 + 255) / 256;
        if (nbg > background) nbg = background;                 // `background` = workgroups (e.g. 256: one per CU)
        hipLaunchKernelGGL(adamw_bg_kernel, dim3((unsigned)nbg), dim3(256), 0, (hipStream_t)stream, nvec4, (bf16_t*)param,
                           (bf16_t*)grad, (bf16_t*)exp_avg, (bf16_t*)exp_avg_sq, clip_coef, (bf16_t*)ema_shadow, a);
        YAT_CHECK_LAUNCH();
        return YAT_OK;
    }
    const int64_t nvec = n >> 3;
    static const int variant = YAT_TUNE_INT("YAT_ADAMW_VARIANT", 3);      // 0: one vector per iteration; 1: two; 2: two, non-temporal; 3: one, non-temporal (alone all four are equal; in the step the non-temporal form is 0.3 ms shorter: the 22 GB of state do not sweep the Infinity Cache under the next forward)
    static const int max_blocks = YAT_TUNE_INT("YAT_ADAMW_BLOCKS", 8192);
    const int U = (variant == 1 || variant == 2) ? 2 : 1;
    int64_t nb = (nvec + 256 * U - 1) / (256 * U);
    if (nb > max_blocks) nb = max_blocks;
#define YAT_ADAMW_LAUNCH(UU, NTT)                                                                                            \
    hipLaunchKernelGGL((adamw_kernel<UU, NTT>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, nvec, (bf16_t*)param, \
                       (bf16_t*)grad, (bf16_t*)exp_avg, (bf16_t*)exp_avg_sq, clip_coef, (bf16_t*)ema_shadow, a)
    if (variant == 1) YAT_ADAMW_LAUNCH(2, false);
    else if (variant == 2) YAT_ADAMW_LAUNCH(2, true);
    else if (variant == 3) YAT_ADAMW_LAUNCH(1, true);
    else YAT_ADAMW_LAUNCH(1, false);
#undef YAT_ADAMW_LAUNCH
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

}  // extern "C"
