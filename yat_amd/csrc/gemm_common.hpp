// Shared by gemm.hip (128x128 tile) and gemm256.hip (256-row tile): launch parameters and the fused epilogue.
#pragma once
#include "common.hpp"

struct GemmP {
    const bf16_t* A; const bf16_t* B; bf16_t* C;
    int M, N, K; int lda, ldb, ldc;
    const bf16_t* bias;    // [N] or null
    const bf16_t* gate;    // [M/rows_per_batch][gate_ld] or null
    const bf16_t* res;     // [M, ldr] residual / accumulate input or null
    bf16_t* aux;           // [M, ldaux] pre-activation / pre-gate linear output or null
    int ldr, ldaux, gate_ld, rows_per_batch;
    int act;               // 0 none, 1 silu, 2 gelu_tanh
    int nbm, nbn;
    uint64_t a_bytes, b_bytes;
    int ksplit;            // gemm256 only: > 1 -> each workgroup reduces K/ksplit and stores an fp32 partial slab
    float* partial;        // [ksplit][M][N] fp32 (caller workspace)
    const bf16_t* glu_u;   // GLU backward epilogue: u = [u_a | u_g] of the depthwise conv, [M, ld_glu]; C is [M, 2N]
    int ld_glu;
    const bf16_t* pre_add; // adapter path: [M, ld_pre] added to the rounded Linear output before aux / activation / gate
    int ld_pre;
    const bf16_t* dact_z;  // activation-backward epilogue: z = the activation's input, [M, ld_z]; C = bf16(d) * act'(z)
    int ld_z;
    bf16_t* rowsum;        // wgrad layout: [M] row sums of the A operand over K (the Linear's bias gradient) or null
    int rowsum_acc;
    int group;             // gemm256: 256-row tiles per row group of the XCD-contiguous tile order (0 = the default, 4)
    const bf16_t* A2;      // gemm256, forward layout: second operand pair, C = epilogue(A B^T + A2 B2^T); A2 [M, .] with row
    const bf16_t* B2;      // stride lda, B2 [N, K2] with row stride ldb (yat_gemm_epilogue.a2 / b2 / k2)
    int K2;                // multiple of 64
    int a2_group;          // 0, or columns of C per block of K2 columns of A2 (fused q|k|v views: one T per target)
    uint64_t a2_bytes, b2_bytes;
};

// GLU backward fused into the producer of dy (= this GEMM's result d, rounded to bf16 like the Linear's output):
//   y = u_a * SiLU(u_g)  ->  du_a = d * bf16(SiLU(u_g)),  du_g = bf16(d * u_a) * SiLU'(u_g)
// -- exactly the arithmetic of the depthwise kernel's backward pass 1, which this replaces.
// Compiled only into the GLU instantiation of gemm256 (as a runtime branch of the shared epilogue it pushed the
// accumulators of every 256x320 kernel into scratch: +11 ms per step).
// Epilogue inputs of one 8-column unit, loaded AHEAD of the unit's slab round trip (gemm256: one pass of 32 rows ahead) so
// that their HBM latency is not paid four times per tile: a = residual | GLU u_a | activation input z, b = pre_add | GLU u_g.
struct EpiIn { u32x4 a, b; };

template <int W>
__device__ __forceinline__ void glu_bwd_store(const GemmP& p, const float (&v)[W], int m, int n, const EpiIn* pre = nullptr) {
    float ua[W], ug[W], da[W], dg[W];
    const bf16_t* up = p.glu_u + (int64_t)m * p.ld_glu + n;
    if (W == 8) {
        unpack8(pre ? pre->a : *reinterpret_cast<const u32x4*>(up), ua);
        unpack8(pre ? pre->b : *reinterpret_cast<const u32x4*>(up + p.N), ug);
    } else {
        unpack4(*reinterpret_cast<const u32x2*>(up), ua);
        unpack4(*reinterpret_cast<const u32x2*>(up + p.N), ug);
    }
#pragma unroll
    for (int e = 0; e < W; ++e) {
        const float d = rbf(v[e]);
        da[e] = d * rbf(silu_f(ug[e]));
        dg[e] = rbf(d * ua[e]) * dsilu_f(ug[e]);
    }
    bf16_t* cp = p.C + (int64_t)m * p.ldc + n;
    if (W == 8) {
        *reinterpret_cast<u32x4*>(cp) = pack8(da);
        *reinterpret_cast<u32x4*>(cp + p.N) = pack8(dg);
    } else {
        *reinterpret_cast<u32x2*>(cp) = pack4(da[0], da[1], da[2], da[3]);
        *reinterpret_cast<u32x2*>(cp + p.N) = pack4(dg[0], dg[1], dg[2], dg[3]);
    }
}

// Activation backward fused into the producer of the activation's output gradient (= this GEMM's result d, rounded to
// bf16 like the Linear's output): dz = d * act'(z) -- the arithmetic of ew_kernel<1> (yat_act_bwd), which this replaces.
// Its own template instantiation of gemm256 (EPI = 3), like the GLU one.
template <int W>
__device__ __forceinline__ void act_bwd_store(const GemmP& p, const float (&v)[W], int m, int n, const EpiIn* pre = nullptr) {
    float z[W], o[W];
    const bf16_t* zp = p.dact_z + (int64_t)m * p.ld_z + n;
    if (W == 8) unpack8(pre ? pre->a : *reinterpret_cast<const u32x4*>(zp), z);
    else unpack4(*reinterpret_cast<const u32x2*>(zp), z);
#pragma unroll
    for (int e = 0; e < W; ++e) o[e] = rbf(v[e]) * (p.act == 1 ? dsilu_f(z[e]) : dgelu_tanh_f(z[e]));
    bf16_t* cp = p.C + (int64_t)m * p.ldc + n;
    if (W == 8) *reinterpret_cast<u32x4*>(cp) = pack8(o);
    else *reinterpret_cast<u32x2*>(cp) = pack4(o[0], o[1], o[2], o[3]);
}

// One lane's 4 consecutive output columns of row m (swapped-operand MFMA result):
// +bias -> round bf16 (the Linear's output) -> aux store -> activation -> *gate (rounded) -> +residual -> store.
// PRE (own template instantiation, like the GLU one): result = bf16( bf16(acc + bias) + pre_add[m,n] ) -- a PEFT adapter's
// ``base_layer(x) + F.linear(x, delta_w)`` with the second term computed by a previous launch.
template <bool PRE = false>
__device__ __forceinline__ void gemm_epilogue_store(const GemmP& p, const f32x4& a, int m, int n, int b) {
    float v[4] = {a[0], a[1], a[2], a[3]};
    if (p.bias) {
        float bb[4];
        unpack4(*reinterpret_cast<const u32x2*>(p.bias + n), bb);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bb[e];
    }
    if (PRE) {
        float pa[4];
        unpack4(*reinterpret_cast<const u32x2*>(p.pre_add + (int64_t)m * p.ld_pre + n), pa);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rbf(v[e]) + pa[e];
    }
    if (p.aux || p.act || p.res) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rbf(v[e]);
    }
    if (p.aux) *reinterpret_cast<u32x2*>(p.aux + (int64_t)m * p.ldaux + n) = pack4(v[0], v[1], v[2], v[3]);
    if (p.act == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
    } else if (p.act == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(v[e]);
    }
    if (p.gate) {
        float g[4];
        unpack4(*reinterpret_cast<const u32x2*>(p.gate + (int64_t)b * p.gate_ld + n), g);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rbf(g[e] * v[e]);
    }
    if (p.res) {
        float r[4];
        unpack4(*reinterpret_cast<const u32x2*>(p.res + (int64_t)m * p.ldr + n), r);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
    }
    *reinterpret_cast<u32x2*>(p.C + (int64_t)m * p.ldc + n) = pack4(v[0], v[1], v[2], v[3]);
}

// Same for 8 consecutive columns (16-B accesses; n % 8 == 0 and every leading dimension % 8 == 0).
// NTC: the result is written non-temporally (weight gradients: next read by the all-reduce / the optimizer, milliseconds later);
// the pre-activation copy (aux) always is -- it is kept for the backward only (common.hpp YAT_AUX_NT).
template <bool PRE = false, bool NTC = false>
__device__ __forceinline__ void gemm_epilogue_store8(const GemmP& p, float (&v)[8], int m, int n, int b, const EpiIn* pre = nullptr) {
    if (p.bias) {
        float bb[8];
        unpack8(*reinterpret_cast<const u32x4*>(p.bias + n), bb);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bb[e];
    }
    if (PRE) {
        float pa[8];
        unpack8(pre ? pre->b : *reinterpret_cast<const u32x4*>(p.pre_add + (int64_t)m * p.ld_pre + n), pa);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rbf(v[e]) + pa[e];
    }
    if (p.aux || p.act || p.res) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rbf(v[e]);
    }
    if (p.aux) {
        __builtin_nontemporal_store(pack8(v), reinterpret_cast<u32x4*>(p.aux + (int64_t)m * p.ldaux + n));
    }
    if (p.act == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
    } else if (p.act == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_tanh_f(v[e]);
    }
    if (p.gate) {
        float g[8];
        unpack8(*reinterpret_cast<const u32x4*>(p.gate + (int64_t)b * p.gate_ld + n), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rbf(g[e] * v[e]);
    }
    if (p.res) {
        float r[8];
        unpack8(pre ? pre->a : *reinterpret_cast<const u32x4*>(p.res + (int64_t)m * p.ldr + n), r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e];
    }
    if (NTC) __builtin_nontemporal_store(pack8(v), reinterpret_cast<u32x4*>(p.C + (int64_t)m * p.ldc + n));
    else *reinterpret_cast<u32x4*>(p.C + (int64_t)m * p.ldc + n) = pack8(v);
}

// gemm256.hip
int yat_gemm256_launch(int a_t, int b_t, int nt_variant, const GemmP& p, hipStream_t stream);
// sums the ksplit fp32 slabs and applies the fused epilogue
int yat_gemm_splitk_reduce(const GemmP& p, hipStream_t stream);
