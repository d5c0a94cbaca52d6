// GLUMBConv middle for gfx950, on the token-major (channels-last) layout:
//   s = SiLU(z) [fused into the conv_inverted GEMM epilogue, which stores both s and z];
//   u = dwconv3x3(s) + bias;  y = u[:Hc] * SiLU(u[Hc:])
// (diffusers GLUMBConv as called at /root/reference/utils/patch_sana_attention_layers.py:110-113;
// z is the conv_inverted output, so the reference's NCHW permutes at :110,112 disappear).
//
// HBM-bound (read z, write y).  Lanes run along channels (4 channels = 8 B per lane, 512 B
// contiguous per wave-instruction).  A thread owns one image row segment of SEG output columns and
// walks it left to right with rolling accumulators: every input column (rows i-1, i, i+1) is
// loaded once and scattered into the three output columns it touches, so no 3x3 window lives in
// registers and each z element is fetched ~3.75x from L2 instead of 9x.  The next column's loads
// are issued before the current column is consumed (register double buffer) so L2/HBM latency
// hides under the FMA work.  (Recomputing SiLU on every load made the kernel VALU-bound: 3.75 SiLUs per
// element; the producer GEMM now applies it once.)
// Backward = two passes: (1) recompute u, emit du (bf16); (2) transposed conv of du -> dz (times
// SiLU'(z)), with the weight / bias gradient partials accumulated in registers over ROWS x SEG
// pixels, reduced across the block's segments in LDS, then across blocks by a small kernel.
#include "common.hpp"
#include "../../include/yat_hip.h"

namespace {

constexpr int SEG = 8;      // output columns per thread
constexpr int ROWS = 4;     // rows per thread in backward pass 2

struct Col6 { u32x2 v[6]; };   // rows (i-1, i, i+1) x (half a, half g), packed bf16x4

__device__ __forceinline__ u32x2 ld_or_zero(const bf16_t* p, bool ok) {
    u32x2 z = {0u, 0u};
    if (ok) z = *reinterpret_cast<const u32x2*>(p);
    return z;
}

// Work-unit order for a 1-D grid: workgroups are dealt round-robin to the 8 XCDs (each with a private L2), so unit
// u = xcd * ceil(total/8) + slot gives every XCD one contiguous run of units.  With (segment, row) fastest inside a
// (channel chunk, image) the three-row halo a unit re-reads was fetched by its neighbour on the SAME L2 moments before.
__device__ __forceinline__ int xcd_unit(int total) {
    const int per = (total + 7) >> 3;
    return (blockIdx.x & 7) * per + (blockIdx.x >> 3);
}

// MODE 0: forward (writes y).  MODE 1: backward pass 1 (reads dy, writes du for both halves).
// grid = 8 * ceil(nx * h * nseg * B / 8), nx = ceil(Hc/4 / 256)
template <int MODE>
__global__ __launch_bounds__(256) void dwconv_glu_kernel(int h, int w, int Hc, int nseg, int nx, int B,
                                                         const bf16_t* z /* = SiLU(conv_inverted) */,
                                                         const bf16_t* wdw, const bf16_t* bdw, const bf16_t* dy,
                                                         bf16_t* out) {
    const int total = nx * h * nseg * B;
    int u = xcd_unit(total);
    if (u >= total) return;
    const int seg = u % nseg; u /= nseg;
    const int i = u % h; u /= h;
    const int q = (u % nx) * 256 + threadIdx.x;            // 4-channel group of the `a` half
    const int b = u / nx;
    if (q * 4 >= Hc) return;
    const int j0 = seg * SEG, j1 = min(w, j0 + SEG);
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
    unpack4(*reinterpret_cast<const u32x2*>(bdw + ca), ba);
    unpack4(*reinterpret_cast<const u32x2*>(bdw + cg), bg);
    float aP[4], aC[4], aN[4], gP[4], gC[4], gN[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { aP[e] = aC[e] = aN[e] = ba[e]; gP[e] = gC[e] = gN[e] = bg[e]; }
    const bf16_t* zb = z + (int64_t)b * h * w * C2;

    auto load_col = [&](int jj) {
        Col6 c;
        const bool colok = jj >= 0 && jj < w;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ii = i + r - 1;
            const bool ok = colok && ii >= 0 && ii < h;
            const bf16_t* zp = zb + ((int64_t)ii * w + jj) * C2;
            c.v[2 * r] = ld_or_zero(zp + ca, ok);
            c.v[2 * r + 1] = ld_or_zero(zp + cg, ok);
        }
        return c;
    };

    Col6 nxt = load_col(j0 - 1);
    for (int jj = j0 - 1; jj <= j1; ++jj) {
        const Col6 cur = nxt;
        if (jj < j1) nxt = load_col(jj + 1);           // prefetch before consuming `cur`
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float za[4], zg[4];
            unpack4(cur.v[2 * r], za);
            unpack4(cur.v[2 * r + 1], zg);
            const int tr = r * 3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sa = za[e], sg = zg[e];   // already SiLU(z) in bf16; zero padding of s
                aP[e] += wa[tr + 2][e] * sa; aC[e] += wa[tr + 1][e] * sa; aN[e] += wa[tr][e] * sa;
                gP[e] += wg[tr + 2][e] * sg; gC[e] += wg[tr + 1][e] * sg; gN[e] += wg[tr][e] * sg;
            }
        }
        if (jj - 1 >= j0) {   // output column jj-1 has seen inputs jj-2 .. jj
            const int64_t pix = ((int64_t)b * h + i) * w + (jj - 1);
            float ua[4], ug[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { ua[e] = rbf(aP[e]); ug[e] = rbf(gP[e]); }
            if (MODE == 0) {
                *reinterpret_cast<u32x2*>(out + pix * Hc + ca) =
                    pack4(ua[0] * rbf(silu_f(ug[0])), ua[1] * rbf(silu_f(ug[1])), ua[2] * rbf(silu_f(ug[2])),
                          ua[3] * rbf(silu_f(ug[3])));
            } else {
                float d[4], da[4], dg[4];
                unpack4(*reinterpret_cast<const u32x2*>(dy + pix * Hc + ca), d);
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

// backward pass 2.  Block = 64 channel groups (4 channels each, either half) x 4 column segments;
// thread = rows [i0, i0+ROWS) x columns [j0, j0+SEG) of image b.
//   dz[i,j] = SiLU'(z[i,j]) * bf16( sum_taps W[tap] du[i-di, j-dj] )
//   dW[tap] += s(z[i,j]) * du[i-di, j-dj];   db += du[i,j]
// partials: ws[((b*nrg + rg)*nsb + sb)][2Hc*10]  (10 = 9 taps + bias per channel)
__global__ __launch_bounds__(256) void dwconv_bwd2_kernel(int h, int w, int Hc, int nx, int nrg, int B, const bf16_t* sact, const bf16_t* z,
                                                          const bf16_t* wdw, const bf16_t* du, bf16_t* dz, float* ws) {
    __shared__ float red[4][64][41];
    const int C2 = 2 * Hc;
    const int lg = threadIdx.x & 63, lseg = threadIdx.x >> 6;
    const int nsb = (w + 4 * SEG - 1) / (4 * SEG);           // segment-blocks per row
    const int total = nx * nrg * nsb * B;
    int u = xcd_unit(total);
    if (u >= total) return;                                   // whole block leaves together (before any barrier)
    const int sb = u % nsb; u /= nsb;
    const int rg = u % nrg; u /= nrg;
    const int bx = u % nx, b = u / nx;
    const int q = bx * 64 + lg;
    const bool active = q * 4 < C2;
    const int c0 = active ? q * 4 : 0;
    const int j0 = (sb * 4 + lseg) * SEG, j1 = min(w, j0 + SEG);
    float wt[9][4], dW[9][4], db[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        db[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) { wt[t][e] = bf2f(wdw[(c0 + e) * 9 + t]); dW[t][e] = 0.f; }
    }
    const bf16_t* zb = z + (int64_t)b * h * w * C2;
    const bf16_t* sb_ = sact + (int64_t)b * h * w * C2;
    const bf16_t* dub = du + (int64_t)b * h * w * C2;
    bf16_t* dzb = dz + (int64_t)b * h * w * C2;
    if (active && j0 < w) {
        for (int i = rg * ROWS; i < min(h, rg * ROWS + ROWS); ++i) {
            auto load_du = [&](int jj, u32x2 (&d)[3]) {
                const bool colok = jj >= 0 && jj < w;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int ii = i + r - 1;
                    d[r] = ld_or_zero(dub + ((int64_t)ii * w + jj) * C2 + c0, colok && ii >= 0 && ii < h);
                }
            };
            auto load_z = [&](int jj) { return ld_or_zero(zb + ((int64_t)i * w + jj) * C2 + c0, jj >= 0 && jj < w); };
            auto load_s = [&](int jj) { return ld_or_zero(sb_ + ((int64_t)i * w + jj) * C2 + c0, jj >= 0 && jj < w); };
            float aP[4], aC[4], aN[4], sP[4], sC[4], sN[4], zP[4], zC[4], zN[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { aP[e] = aC[e] = aN[e] = 0.f; }
            // centre-row z window around input column jj: P = jj-1, C = jj, N = jj+1
            unpack4(load_z(j0 - 2), zP);
            unpack4(load_z(j0 - 1), zC);
            unpack4(load_s(j0 - 2), sP);
            unpack4(load_s(j0 - 1), sC);
            u32x2 dn[3];
            load_du(j0 - 1, dn);
            u32x2 zn = load_z(j0), sn = load_s(j0);
            for (int jj = j0 - 1; jj <= j1; ++jj) {
                u32x2 dc[3] = {dn[0], dn[1], dn[2]};
                unpack4(zn, zN);
                unpack4(sn, sN);
                if (jj < j1) { load_du(jj + 1, dn); zn = load_z(jj + 2); sn = load_s(jj + 2); }      // prefetch
                // only contributions to output columns inside [j0, j1) count for dW (each (pixel, tap) pair once)
                const bool inP = jj - 1 >= j0 && jj - 1 < j1, inC = jj >= j0 && jj < j1, inN = jj + 1 >= j0 && jj + 1 < j1;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    float d[4];
                    unpack4(dc[r], d);
                    const int tr = (2 - r) * 3;      // du row ii = i + r - 1  ->  di = 1 - r  ->  (di+1)*3
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        aP[e] += wt[tr][e] * d[e];
                        aC[e] += wt[tr + 1][e] * d[e];
                        aN[e] += wt[tr + 2][e] * d[e];
                        if (inP) dW[tr][e] += sP[e] * d[e];
                        if (inC) dW[tr + 1][e] += sC[e] * d[e];
                        if (inN) dW[tr + 2][e] += sN[e] * d[e];
                        if (r == 1 && inC) db[e] += d[e];
                    }
                }
                if (inP) {
                    *reinterpret_cast<u32x2*>(dzb + ((int64_t)i * w + jj - 1) * C2 + c0) =
                        pack4(rbf(aP[0]) * dsilu_f(zP[0]), rbf(aP[1]) * dsilu_f(zP[1]), rbf(aP[2]) * dsilu_f(zP[2]),
                              rbf(aP[3]) * dsilu_f(zP[3]));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    aP[e] = aC[e]; aC[e] = aN[e]; aN[e] = 0.f;
                    sP[e] = sC[e]; sC[e] = sN[e]; zP[e] = zC[e]; zC[e] = zN[e];
                }
            }
        }
    }
    // reduce the 4 segment threads of each channel group in LDS, then one partial row per block
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int t = 0; t < 9; ++t) red[lseg][lg][e * 10 + t] = dW[t][e];
        red[lseg][lg][e * 10 + 9] = db[e];
    }
    __syncthreads();
    if (lseg == 0 && active) {
        float* wp = ws + (((int64_t)b * nrg * nsb + rg * nsb + sb)) * C2 * 10 + (int64_t)c0 * 10;
#pragma unroll
        for (int k = 0; k < 40; ++k) wp[k] = red[0][lg][k] + red[1][lg][k] + red[2][lg][k] + red[3][lg][k];
    }
}

__global__ void dwconv_reduce_kernel(int P, int C2, const float* ws, bf16_t* dw, bf16_t* dbias, int accumulate) {
    __shared__ float red[4][64];
    const int idx = blockIdx.x * 64 + (threadIdx.x & 63);    // over C2*10
    const int part = threadIdx.x >> 6;
    float s = 0.f;
    if (idx < C2 * 10)
        for (int p = part; p < P; p += 4) s += ws[(int64_t)p * C2 * 10 + idx];
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part != 0 || idx >= C2 * 10) return;
    s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    const int ch = idx / 10, t = idx % 10;
    bf16_t* dst = t < 9 ? dw + ch * 9 + t : dbias + ch;
    if (accumulate) s = rbf(s) + bf2f(*dst);
    *dst = f2bf(s);
}

inline int nseg_of(int w) { return (w + SEG - 1) / SEG; }
inline int nsb_of(int w) { return (w + 4 * SEG - 1) / (4 * SEG); }
inline int nrg_of(int h) { return (h + ROWS - 1) / ROWS; }
inline unsigned grid8(int64_t total) { return (unsigned)(((total + 7) / 8) * 8); }

}  // namespace

extern "C" {

int yat_dwconv_glu_fwd(int B, int h, int w, int Hc, const void* s, const void* wdw, const void* bdw, void* y,
                       yat_stream_t stream) {
    const void* z = s;
    if (B <= 0 || h <= 0 || w <= 0 || Hc <= 0 || (Hc & 3) || !z || !wdw || !bdw || !y) return YAT_EINVAL;
    if ((int64_t)h * nseg_of(w) * B * ((Hc / 4 + 63) / 64) * 2 > 0x7fffff00ll) return YAT_EINVAL;
    const int nx = (Hc / 4 + 255) / 256;
    hipLaunchKernelGGL((dwconv_glu_kernel<0>), dim3(grid8((int64_t)nx * h * nseg_of(w) * B)), dim3(256), 0,
                       (hipStream_t)stream, h, w, Hc, nseg_of(w), nx, B, (const bf16_t*)z, (const bf16_t*)wdw,
                       (const bf16_t*)bdw, (const bf16_t*)nullptr, (bf16_t*)y);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

uint64_t yat_dwconv_glu_bwd_workspace_bytes(int B, int h, int w, int Hc) {
    // du (bf16 [B,h,w,2Hc]) followed by the fp32 partials
    const uint64_t du_bytes = ((uint64_t)B * h * w * 2 * Hc * 2 + 255) & ~255ull;
    return du_bytes + (uint64_t)B * nrg_of(h) * nsb_of(w) * 2 * Hc * 10 * sizeof(float);
}

int yat_dwconv_glu_bwd(int B, int h, int w, int Hc, const void* s, const void* z, const void* wdw, const void* bdw,
                       const void* dy, void* dz, void* dwdw, void* dbdw, int accumulate, void* workspace,
                       yat_stream_t stream) {
    if (B <= 0 || h <= 0 || w <= 0 || Hc <= 0 || (Hc & 3) || !s || !z || !wdw || !bdw || !dy || !dz || !dwdw || !dbdw ||
        !workspace)
        return YAT_EINVAL;
    if ((int64_t)h * nseg_of(w) * B * ((Hc / 4 + 63) / 64) * 2 > 0x7fffff00ll) return YAT_EINVAL;
    const int C2 = 2 * Hc;
    const int gy2 = nrg_of(h) * nsb_of(w);
    bf16_t* du = (bf16_t*)workspace;
    const uint64_t du_bytes = ((uint64_t)B * h * w * C2 * 2 + 255) & ~255ull;
    float* ws = (float*)((char*)workspace + du_bytes);
    const int nx = (Hc / 4 + 255) / 256, nx2 = (C2 / 4 + 63) / 64;
    hipLaunchKernelGGL((dwconv_glu_kernel<1>), dim3(grid8((int64_t)nx * h * nseg_of(w) * B)), dim3(256), 0,
                       (hipStream_t)stream, h, w, Hc, nseg_of(w), nx, B, (const bf16_t*)s, (const bf16_t*)wdw,
                       (const bf16_t*)bdw, (const bf16_t*)dy, du);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(dwconv_bwd2_kernel, dim3(grid8((int64_t)nx2 * gy2 * B)), dim3(256), 0, (hipStream_t)stream, h, w,
                       Hc, nx2, nrg_of(h), B, (const bf16_t*)s, (const bf16_t*)z, (const bf16_t*)wdw, (const bf16_t*)du,
                       (bf16_t*)dz, ws);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(dwconv_reduce_kernel, dim3((C2 * 10 + 63) / 64), dim3(256), 0, (hipStream_t)stream, B * gy2, C2,
                       (const float*)ws, (bf16_t*)dwdw, (bf16_t*)dbdw, accumulate);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

}  // extern "C"
