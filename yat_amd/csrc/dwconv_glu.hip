// GLUMBConv middle for gfx950, on the token-major (channels-last) layout:
//   s = SiLU(z);  u = dwconv3x3(s) + bias;  y = u[:Hc] * SiLU(u[Hc:])
// (diffusers GLUMBConv as called at /root/reference/utils/patch_sana_attention_layers.py:110-113;
// z is the conv_inverted output, so the reference's NCHW permutes at :110,112 disappear).
//
// HBM-bound (read z, write y).  Lanes run along channels (4 channels = 8 B per lane, 512 B
// contiguous per wave-instruction); each thread walks one image row left to right with a
// rolling-accumulator scheme: an input column is loaded once per (row-1,row,row+1) and scattered
// into the three output columns it touches, so no 3x3 window sits in registers and each z element
// is fetched 3x (neighbouring rows; L2 hits) instead of 9x.  SiLU is fused on load.
// Backward = two passes: (1) recompute u, emit du (bf16); (2) transposed conv of du -> dz (times
// SiLU'(z)) with the weight / bias gradient partials accumulated in registers over several rows.
#include "common.hpp"
#include "../../include/yat_hip.h"

namespace {

__device__ __forceinline__ void ld4(const bf16_t* p, float* o) { unpack4(*reinterpret_cast<const u32x2*>(p), o); }

// MODE 0: forward (writes y).  MODE 1: backward pass 1 (reads dy, writes du for both halves).
template <int MODE>
__global__ __launch_bounds__(256) void dwconv_glu_kernel(int h, int w, int Hc, const bf16_t* z, const bf16_t* wdw,
                                                         const bf16_t* bdw, const bf16_t* dy, bf16_t* out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;   // 4-channel group of the `a` half
    if (q * 4 >= Hc) return;
    const int i = blockIdx.y, b = blockIdx.z;
    const int C2 = 2 * Hc;
    const int ca = q * 4, cg = Hc + q * 4;
    float wa[9][4], wg[9][4], ba[4], bg[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            wa[t][e] = bf2f(wdw[(ca + e) * 9 + t]);
            wg[t][e] = bf2f(wdw[(cg + e) * 9 + t]);
        }
    }
    ld4(bdw + ca, ba);
    ld4(bdw + cg, bg);
    float aP[4], aC[4], aN[4], gP[4], gC[4], gN[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { aP[e] = aC[e] = aN[e] = ba[e]; gP[e] = gC[e] = gN[e] = bg[e]; }
    const bf16_t* zb = z + (int64_t)b * h * w * C2;
    for (int jj = 0; jj <= w; ++jj) {
        if (jj < w) {
#pragma unroll
            for (int r = -1; r <= 1; ++r) {
                const int ii = i + r;
                if (ii < 0 || ii >= h) continue;
                const bf16_t* zp = zb + ((int64_t)ii * w + jj) * C2;
                float za[4], zg[4];
                ld4(zp + ca, za);
                ld4(zp + cg, zg);
                const int tr = (r + 1) * 3;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sa = rbf(silu_f(za[e])), sg = rbf(silu_f(zg[e]));
                    aP[e] += wa[tr + 2][e] * sa; aC[e] += wa[tr + 1][e] * sa; aN[e] += wa[tr][e] * sa;
                    gP[e] += wg[tr + 2][e] * sg; gC[e] += wg[tr + 1][e] * sg; gN[e] += wg[tr][e] * sg;
                }
            }
        }
        if (jj >= 1) {   // output column jj-1 is complete
            const int64_t pix = ((int64_t)b * h + i) * w + (jj - 1);
            float ua[4], ug[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { ua[e] = rbf(aP[e]); ug[e] = rbf(gP[e]); }
            if (MODE == 0) {
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = ua[e] * rbf(silu_f(ug[e]));
                *reinterpret_cast<u32x2*>(out + pix * Hc + ca) = pack4(y[0], y[1], y[2], y[3]);
            } else {
                float d[4], da[4], dg[4];
                ld4(dy + pix * Hc + ca, d);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    da[e] = d[e] * rbf(silu_f(ug[e]));                 // d u_a
                    dg[e] = rbf(d[e] * ua[e]) * dsilu_f(ug[e]);        // d u_g
                }
                *reinterpret_cast<u32x2*>(out + pix * C2 + ca) = pack4(da[0], da[1], da[2], da[3]);
                *reinterpret_cast<u32x2*>(out + pix * C2 + cg) = pack4(dg[0], dg[1], dg[2], dg[3]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { aP[e] = aC[e]; aC[e] = aN[e]; aN[e] = ba[e]; gP[e] = gC[e]; gC[e] = gN[e]; gN[e] = bg[e]; }
    }
}

// backward pass 2: thread = 4 channels (either half), rows [i0, i0+R) of image b.
//   dz[i,j] = SiLU'(z[i,j]) * bf16( sum_taps W[tap] du[i-di, j-dj] )
//   dW[tap] += s(z[i,j]) * du[i-di, j-dj];   db += du[i,j]
// partials: ws[(b*nrg + rg)][2Hc*10]  (10 = 9 taps + bias per channel)
__global__ __launch_bounds__(256) void dwconv_bwd2_kernel(int h, int w, int Hc, int R, const bf16_t* z, const bf16_t* wdw,
                                                          const bf16_t* du, bf16_t* dz, float* ws) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int C2 = 2 * Hc;
    if (q * 4 >= C2) return;
    const int rg = blockIdx.y, b = blockIdx.z, c0 = q * 4;
    float wt[9][4], dW[9][4], db[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        db[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) { wt[t][e] = bf2f(wdw[(c0 + e) * 9 + t]); dW[t][e] = 0.f; }
    }
    const bf16_t* zb = z + (int64_t)b * h * w * C2;
    const bf16_t* dub = du + (int64_t)b * h * w * C2;
    bf16_t* dzb = dz + (int64_t)b * h * w * C2;
    for (int i = rg * R; i < min(h, rg * R + R); ++i) {
        float aP[4], aC[4], aN[4], sP[4], sC[4], sN[4], zP[4], zC[4], zN[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { aP[e] = aC[e] = aN[e] = 0.f; sP[e] = sC[e] = 0.f; zP[e] = zC[e] = 0.f; }
        // centre-row z window: columns jj-1 (P), jj (C), jj+1 (N)
        ld4(zb + ((int64_t)i * w + 0) * C2 + c0, zC);
#pragma unroll
        for (int e = 0; e < 4; ++e) sC[e] = rbf(silu_f(zC[e]));
        for (int jj = 0; jj <= w; ++jj) {
            if (jj + 1 < w) {
                ld4(zb + ((int64_t)i * w + jj + 1) * C2 + c0, zN);
#pragma unroll
                for (int e = 0; e < 4; ++e) sN[e] = rbf(silu_f(zN[e]));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { zN[e] = 0.f; sN[e] = 0.f; }
            }
            if (jj < w) {
#pragma unroll
                for (int r = -1; r <= 1; ++r) {
                    const int ii = i + r;          // du row; di = -r
                    if (ii < 0 || ii >= h) continue;
                    float d[4];
                    ld4(dub + ((int64_t)ii * w + jj) * C2 + c0, d);
                    const int tr = (1 - r) * 3;    // (di+1)*3
                    // output column jc = jj-1: dj = jc - jj = -1 -> tap tr+0 ; jc = jj: tap tr+1 ; jc = jj+1: tap tr+2
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        aP[e] += wt[tr][e] * d[e];     dW[tr][e] += sP[e] * d[e];
                        aC[e] += wt[tr + 1][e] * d[e]; dW[tr + 1][e] += sC[e] * d[e];
                        aN[e] += wt[tr + 2][e] * d[e]; dW[tr + 2][e] += sN[e] * d[e];
                        if (r == 0) db[e] += d[e];
                    }
                }
            }
            if (jj >= 1) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rbf(aP[e]) * dsilu_f(zP[e]);
                *reinterpret_cast<u32x2*>(dzb + ((int64_t)i * w + jj - 1) * C2 + c0) = pack4(o[0], o[1], o[2], o[3]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                aP[e] = aC[e]; aC[e] = aN[e]; aN[e] = 0.f;
                sP[e] = sC[e]; sC[e] = sN[e]; zP[e] = zC[e]; zC[e] = zN[e];
            }
        }
    }
    float* wp = ws + ((int64_t)b * gridDim.y + rg) * C2 * 10 + (int64_t)c0 * 10;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wp[e * 10 + t] = dW[t][e];
        wp[e * 10 + 9] = db[e];
    }
}

__global__ void dwconv_reduce_kernel(int P, int C2, const float* ws, bf16_t* dw, bf16_t* dbias, int accumulate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over C2*10
    if (idx >= C2 * 10) return;
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += ws[(int64_t)p * C2 * 10 + idx];
    const int ch = idx / 10, t = idx % 10;
    bf16_t* dst = t < 9 ? dw + ch * 9 + t : dbias + ch;
    if (accumulate) s = rbf(s) + bf2f(*dst);
    *dst = f2bf(s);
}

constexpr int ROWS_PER_THREAD = 8;

}  // namespace

extern "C" {

int yat_dwconv_glu_fwd(int B, int h, int w, int Hc, const void* z, const void* wdw, const void* bdw, void* y,
                       yat_stream_t stream) {
    if (B <= 0 || h <= 0 || w <= 0 || Hc <= 0 || (Hc & 3) || !z || !wdw || !bdw || !y) return YAT_EINVAL;
    const int groups = Hc / 4;
    hipLaunchKernelGGL((dwconv_glu_kernel<0>), dim3((groups + 255) / 256, h, B), dim3(256), 0, (hipStream_t)stream, h, w,
                       Hc, (const bf16_t*)z, (const bf16_t*)wdw, (const bf16_t*)bdw, (const bf16_t*)nullptr, (bf16_t*)y);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

uint64_t yat_dwconv_glu_bwd_workspace_bytes(int B, int h, int w, int Hc) {
    const uint64_t nrg = (h + ROWS_PER_THREAD - 1) / ROWS_PER_THREAD;
    // du (bf16 [B,h,w,2Hc]) followed by the fp32 partials
    const uint64_t du_bytes = ((uint64_t)B * h * w * 2 * Hc * 2 + 255) & ~255ull;
    return du_bytes + (uint64_t)B * nrg * 2 * Hc * 10 * sizeof(float);
}

int yat_dwconv_glu_bwd(int B, int h, int w, int Hc, const void* z, const void* wdw, const void* bdw, const void* dy,
                       void* dz, void* dwdw, void* dbdw, int accumulate, void* workspace, yat_stream_t stream) {
    if (B <= 0 || h <= 0 || w <= 0 || Hc <= 0 || (Hc & 3) || !z || !wdw || !bdw || !dy || !dz || !dwdw || !dbdw ||
        !workspace)
        return YAT_EINVAL;
    const int C2 = 2 * Hc;
    const int nrg = (h + ROWS_PER_THREAD - 1) / ROWS_PER_THREAD;
    bf16_t* du = (bf16_t*)workspace;
    const uint64_t du_bytes = ((uint64_t)B * h * w * C2 * 2 + 255) & ~255ull;
    float* ws = (float*)((char*)workspace + du_bytes);
    hipLaunchKernelGGL((dwconv_glu_kernel<1>), dim3((Hc / 4 + 255) / 256, h, B), dim3(256), 0, (hipStream_t)stream, h, w,
                       Hc, (const bf16_t*)z, (const bf16_t*)wdw, (const bf16_t*)bdw, (const bf16_t*)dy, du);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(dwconv_bwd2_kernel, dim3((C2 / 4 + 255) / 256, nrg, B), dim3(256), 0, (hipStream_t)stream, h, w, Hc,
                       ROWS_PER_THREAD, (const bf16_t*)z, (const bf16_t*)wdw, (const bf16_t*)du, (bf16_t*)dz, ws);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(dwconv_reduce_kernel, dim3((C2 * 10 + 255) / 256), dim3(256), 0, (hipStream_t)stream, B * nrg, C2,
                       (const float*)ws, (bf16_t*)dwdw, (bf16_t*)dbdw, accumulate);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

}  // extern "C"
