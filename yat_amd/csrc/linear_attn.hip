// ReLU linear attention (SANA self-attention, head dim 32) forward + backward for gfx950.
//
// Restates diffusers SanaLinearAttnProcessor2_0 (imported at
// /root/reference/utils/patch_sana_attention_layers.py:7; stock attn1 processor of the model that
// train_sana.py:210 trains):  q,k <- ReLU;  fp32:  S = [V;1] K   (33x32 per head),
// U = S Q (33 x N),  out = U[:32] / (U[32] + 1e-15).
//
// Two kinds of kernels:
//  * state kernels, one workgroup per (batch, head): the token reduction S = A^T B runs
//    register-tiled over LDS-staged token chunks (each thread owns a 4x1 strip of the 33x32 state:
//    one ds_read_b128 broadcast + one ds_read_b32 per 4 FMAs) and lands in a global fp32 workspace;
//  * apply kernels, one thread per token, grid (token chunks, heads, batch): the 33x32 state is
//    wave-uniform, so it is read through the scalar cache (s_load) and every product
//    (U = S q, dq = S^T dU, dv = dS k, dk = dS^T v') is a VALU FMA with an SGPR operand --
//    no LDS traffic, no register blow-up.
// fp32 throughout, as the reference up-casts q/k/v (0.2 % of block FLOPs; HBM/LDS-bound).
#include "common.hpp"
#include "../../include/yat_hip.h"

namespace {

constexpr int C = 32;          // head dim
constexpr int SS = 33 * C;     // floats per state
constexpr int CH = 128;        // tokens per staging chunk in the state kernel
constexpr int TB = 256;        // tokens per workgroup in the apply kernels

__device__ __forceinline__ void load32(const bf16_t* p, float* o) {
#pragma unroll
    for (int j = 0; j < 4; ++j) unpack8(*reinterpret_cast<const u32x4*>(p + j * 8), o + j * 8);
}
__device__ __forceinline__ void store32(bf16_t* p, const float* o) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(p + j * 8) = pack8(o + j * 8);
}

// s[c'][c] += sum_n a[n][c'] * b[n][c] over an LDS chunk; a rows have stride 36 floats (slot 32 =
// the "ones" row), b rows stride 32.  Thread (c = t&31, g = t>>5) owns c' = 4g..4g+3; g == 0 also c' = 32.
__device__ __forceinline__ void accum_state(const float* a_s, const float* b_s, int cnt, int c, int g, float (&s)[5]) {
    for (int n = 0; n < cnt; ++n) {
        const float bv = b_s[n * C + c];
        const f32x4 av = *reinterpret_cast<const f32x4*>(a_s + n * 36 + 4 * g);
        s[0] += av[0] * bv; s[1] += av[1] * bv; s[2] += av[2] * bv; s[3] += av[3] * bv;
        if (g == 0) s[4] += a_s[n * 36 + 32] * bv;
    }
}
__device__ __forceinline__ void store_state(float* S, int c, int g, const float (&s)[5]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) S[(4 * g + i) * C + c] = s[i];
    if (g == 0) S[32 * C + c] = s[4];
}

// S[b,h] = [V;1]^T relu(K), grid (H, B)
__global__ __launch_bounds__(256) void la_state_kv_kernel(int N, const bf16_t* qkv, int ld, int k_off, int v_off,
                                                          float* S_out) {
    __shared__ __attribute__((aligned(16))) float a_s[CH * 36];
    __shared__ __attribute__((aligned(16))) float b_s[CH * C];
    const int t = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
    const int c = t & 31, g = t >> 5;
    const bf16_t* kb = qkv + (int64_t)b * N * ld + h * C + k_off;
    const bf16_t* vb = qkv + (int64_t)b * N * ld + h * C + v_off;
    float s[5] = {0, 0, 0, 0, 0};
    const int tok = t >> 1, half = t & 1;   // staging: 2 threads per token, 16 channels each
    for (int n0 = 0; n0 < N; n0 += CH) {
        const int n = n0 + tok;
        float kk[16], vv[16];
        if (n < N) {
            unpack8(*reinterpret_cast<const u32x4*>(kb + (int64_t)n * ld + half * 16), kk);
            unpack8(*reinterpret_cast<const u32x4*>(kb + (int64_t)n * ld + half * 16 + 8), kk + 8);
            unpack8(*reinterpret_cast<const u32x4*>(vb + (int64_t)n * ld + half * 16), vv);
            unpack8(*reinterpret_cast<const u32x4*>(vb + (int64_t)n * ld + half * 16 + 8), vv + 8);
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) { kk[e] = 0.f; vv[e] = 0.f; }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            b_s[tok * C + half * 16 + e] = fmaxf(kk[e], 0.f);
            a_s[tok * 36 + half * 16 + e] = vv[e];
        }
        if (half == 0) a_s[tok * 36 + 32] = (n < N) ? 1.0f : 0.f;
        __syncthreads();
        accum_state(a_s, b_s, CH, c, g, s);
    }
    store_state(S_out + ((int64_t)b * gridDim.x + h) * SS, c, g, s);
}

// u[c'] = sum_c S[c'][c] x[c], c' < NR   (S wave-uniform -> scalar loads)
template <int NR>
__device__ __forceinline__ void state_times_vec(const float* __restrict__ S, const float* x, float* u) {
#pragma unroll
    for (int cp = 0; cp < NR; ++cp) {
        float a = 0.f;
#pragma unroll
        for (int cc = 0; cc < C; ++cc) a += S[cp * C + cc] * x[cc];
        u[cp] = a;
        // keep at most two state rows (64 SGPRs) in flight: stops the scheduler from hoisting all
        // 1056 scalar loads to the top and spilling them
        if (cp & 1) __builtin_amdgcn_sched_barrier(0);
    }
}
// y[c] = sum_{c' < NR} w[c'] S[c'][c]
template <int NR>
__device__ __forceinline__ void vec_times_state(const float* __restrict__ S, const float* w, float* y) {
#pragma unroll
    for (int cc = 0; cc < C; ++cc) y[cc] = 0.f;
#pragma unroll
    for (int cp = 0; cp < NR; ++cp) {
#pragma unroll
        for (int cc = 0; cc < C; ++cc) y[cc] += w[cp] * S[cp * C + cc];
        if (cp & 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// forward apply: grid (ceil(N/256), H, B)
__global__ __launch_bounds__(256) void la_apply_fwd_kernel(int N, int H, const bf16_t* __restrict__ qkv, int ld,
                                                           const float* __restrict__ S_all, bf16_t* __restrict__ out,
                                                           int ld_out) {
    const int h = blockIdx.y, b = blockIdx.z;
    const int n = blockIdx.x * TB + threadIdx.x;
    const float* __restrict__ S = S_all + ((int64_t)b * H + h) * SS;
    if (n >= N) return;
    float q[C], u[33];
    load32(qkv + ((int64_t)b * N + n) * ld + h * C, q);
#pragma unroll
    for (int e = 0; e < C; ++e) q[e] = fmaxf(q[e], 0.f);
    state_times_vec<33>(S, q, u);
    const float inv = 1.0f / (u[32] + 1e-15f);
#pragma unroll
    for (int e = 0; e < C; ++e) u[e] *= inv;
    store32(out + ((int64_t)b * N + n) * ld_out + h * C, u);
}

// backward apply 1: per token dU, dq; per workgroup partial dS slab = dU^T relu(Q)
// grid (nchunks, H, B); slabs dS_part[((b*H + h)*nchunks + chunk)][33*32]
__global__ __launch_bounds__(256) void la_apply_bwd_q_kernel(int N, int H, const bf16_t* __restrict__ qkv, int ld,
                                                             const bf16_t* __restrict__ dout, int ld_do,
                                                             const float* __restrict__ S_all,
                                                             const float* __restrict__ S_all_again,
                                                             bf16_t* __restrict__ dqkv, int ld_dq,
                                                             float* __restrict__ dS_part) {
    extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
    float* a_s = dyn_lds;               // [TB][36]
    float* b_s = dyn_lds + TB * 36;     // [TB][32]
    const int t = threadIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int n = blockIdx.x * TB + t;
    const float* __restrict__ S = S_all + ((int64_t)b * H + h) * SS;
    float q[C], du[33];
    if (n < N) {
        float dO[C];
        load32(qkv + ((int64_t)b * N + n) * ld + h * C, q);
        load32(dout + ((int64_t)b * N + n) * ld_do + h * C, dO);
#pragma unroll
        for (int e = 0; e < C; ++e) q[e] = fmaxf(q[e], 0.f);
        state_times_vec<33>(S, q, du);                 // du <- U
        const float inv = 1.0f / (du[32] + 1e-15f);
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < C; ++e) {
            dot += dO[e] * (du[e] * inv);              // dO . O
            du[e] = dO[e] * inv;                       // dU[e]
        }
        du[32] = -dot * inv;
        float dq[C];
        // second sweep through a second (equal) kernel argument: the compiler cannot merge the two
        // sweeps' loads, so the 33x32 state is re-read from the scalar cache instead of being kept
        // live in (spilled) SGPRs
        vec_times_state<33>(S_all_again + ((int64_t)b * H + h) * SS, du, dq);
#pragma unroll
        for (int e = 0; e < C; ++e) dq[e] = q[e] > 0.f ? dq[e] : 0.f;
        store32(dqkv + ((int64_t)b * N + n) * ld_dq + h * C, dq);
    } else {
#pragma unroll
        for (int e = 0; e < C; ++e) q[e] = 0.f;
#pragma unroll
        for (int e = 0; e < 33; ++e) du[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 33; ++e) a_s[t * 36 + e] = du[e];
#pragma unroll
    for (int e = 0; e < C; ++e) b_s[t * C + e] = q[e];
    __syncthreads();
    float ds[5] = {0, 0, 0, 0, 0};
    accum_state(a_s, b_s, TB, t & 31, t >> 5, ds);
    store_state(dS_part + (((int64_t)b * H + h) * gridDim.x + blockIdx.x) * SS, t & 31, t >> 5, ds);
}

// dS[bh] = sum_chunks dS_part[bh][chunk]
__global__ void la_reduce_slabs_kernel(int nbh, int nchunks, const float* part, float* dS) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nbh * SS) return;
    const int64_t bh = i / SS, e = i % SS;
    float s = 0.f;
    for (int c = 0; c < nchunks; ++c) s += part[(bh * nchunks + c) * SS + e];
    dS[i] = s;
}

// backward apply 2: dv = dS[:32] relu(k), dk = dS^T [v;1] masked by k > 0
__global__ __launch_bounds__(256) void la_apply_bwd_kv_kernel(int N, int H, const bf16_t* __restrict__ qkv, int ld,
                                                              int k_off, int v_off, const float* __restrict__ dS_all,
                                                              const float* __restrict__ dS_all_again,
                                                              bf16_t* __restrict__ dqkv, int ld_dq) {
    const int h = blockIdx.y, b = blockIdx.z;
    const int n = blockIdx.x * TB + threadIdx.x;
    const float* __restrict__ dS = dS_all + ((int64_t)b * H + h) * SS;
    if (n >= N) return;
    float k[C], kr[C], v[33], dv[C], dk[C];
    load32(qkv + ((int64_t)b * N + n) * ld + h * C + k_off, k);
    load32(qkv + ((int64_t)b * N + n) * ld + h * C + v_off, v);
    v[32] = 1.0f;
#pragma unroll
    for (int e = 0; e < C; ++e) kr[e] = fmaxf(k[e], 0.f);
    state_times_vec<32>(dS, kr, dv);
    vec_times_state<33>(dS_all_again + ((int64_t)b * H + h) * SS, v, dk);
#pragma unroll
    for (int e = 0; e < C; ++e) dk[e] = k[e] > 0.f ? dk[e] : 0.f;
    store32(dqkv + ((int64_t)b * N + n) * ld_dq + h * C + v_off, dv);
    store32(dqkv + ((int64_t)b * N + n) * ld_dq + h * C + k_off, dk);
}

}  // namespace

extern "C" {

uint64_t yat_linear_attn_workspace_bytes(int B, int N, int H) {
    const uint64_t nchunks = (N + TB - 1) / TB;
    return (uint64_t)B * H * SS * sizeof(float) * (2 + nchunks);
}

int yat_linear_attn_fwd(int B, int N, int H, const void* qkv, int ld, int k_off, int v_off, void* out, int ld_out,
                        void* workspace, yat_stream_t stream) {
    if (B <= 0 || N <= 0 || H <= 0 || (ld & 7) || (k_off & 7) || (v_off & 7) || (ld_out & 7) || !qkv || !out || !workspace)
        return YAT_EINVAL;
    float* S = (float*)workspace;
    hipLaunchKernelGGL(la_state_kv_kernel, dim3(H, B), dim3(256), 0, (hipStream_t)stream, N, (const bf16_t*)qkv, ld, k_off,
                       v_off, S);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(la_apply_fwd_kernel, dim3((N + TB - 1) / TB, H, B), dim3(256), 0, (hipStream_t)stream, N, H,
                       (const bf16_t*)qkv, ld, (const float*)S, (bf16_t*)out, ld_out);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

int yat_linear_attn_bwd(int B, int N, int H, const void* qkv, int ld, int k_off, int v_off, const void* dout, int ld_dout,
                        void* dqkv, int ld_dqkv, const float* state, void* workspace, yat_stream_t stream) {
    if (B <= 0 || N <= 0 || H <= 0 || (ld & 7) || (k_off & 7) || (v_off & 7) || (ld_dout & 7) || (ld_dqkv & 7) || !qkv ||
        !dout || !dqkv || !workspace)
        return YAT_EINVAL;
    const int nchunks = (N + TB - 1) / TB;
    float* S = (float*)workspace;
    float* dS = S + (int64_t)B * H * SS;
    float* part = dS + (int64_t)B * H * SS;
    hipStream_t st = (hipStream_t)stream;
    if (state) {
        S = const_cast<float*>(state);          // the forward's state (first B*H*33*32 floats of its workspace), kept by the caller
    } else {
        hipLaunchKernelGGL(la_state_kv_kernel, dim3(H, B), dim3(256), 0, st, N, (const bf16_t*)qkv, ld, k_off, v_off, S);
        YAT_CHECK_LAUNCH();
    }
    constexpr int BQ_LDS = TB * (36 + C) * (int)sizeof(float);
    static bool attr_set = false;   // idempotent one-time launch attribute (LDS > 64 KiB)
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)la_apply_bwd_q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BQ_LDS) !=
            hipSuccess)
            return YAT_EINVAL;
        attr_set = true;
    }
    hipLaunchKernelGGL(la_apply_bwd_q_kernel, dim3(nchunks, H, B), dim3(256), BQ_LDS, st, N, H, (const bf16_t*)qkv, ld,
                       (const bf16_t*)dout, ld_dout, (const float*)S, (const float*)S, (bf16_t*)dqkv, ld_dqkv, part);
    YAT_CHECK_LAUNCH();
    const int64_t tot = (int64_t)B * H * SS;
    hipLaunchKernelGGL(la_reduce_slabs_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, B * H, nchunks,
                       (const float*)part, dS);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(la_apply_bwd_kv_kernel, dim3(nchunks, H, B), dim3(256), 0, st, N, H, (const bf16_t*)qkv, ld, k_off,
                       v_off, (const float*)dS, (const float*)dS, (bf16_t*)dqkv, ld_dqkv);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

}  // extern "C"
