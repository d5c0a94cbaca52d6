// HBM-bound row kernels of the SANA block for gfx950: LayerNorm+adaLN modulate, RMSNorm, the
// modulation table, gated-residual backward and bias-gradient column sums.
//
// Common shape: one 64-lane wave owns one row (<= 4096 bf16), 16-B loads per lane, the row lives
// in registers between the statistics pass and the output pass (one HBM read, one write), fp32
// math, wave shuffles for the row reductions.  Column reductions (dshift/dscale/dgate/dbias/dw)
// are accumulated per wave in registers over a group of rows, written as fp32 partial rows to a
// caller-provided workspace and finished by one small reduce kernel -- no atomics, bit-reproducible.
#include <type_traits>
#include "common.hpp"
#include "../../include/yat_hip.h"

namespace {

constexpr int ROWS_PER_WAVE = 4;     // rows a wave walks in the backward kernels
constexpr int WAVES = 4;             // 256-thread blocks

template <int MAXV>
__device__ __forceinline__ void load_row(const bf16_t* p, int nchunk, int lane, float (&v)[MAXV][8]) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            unpack8(*reinterpret_cast<const u32x4*>(p + c * 8), v[i]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        }
    }
}

// ------------------------------------------------------------------ LN + modulate forward
template <int MAXV>
__global__ __launch_bounds__(256) void ln_mod_fwd_kernel(int M, int D, int rpb, float eps, const bf16_t* x,
                                                         const bf16_t* shift, const bf16_t* scale, int mod_ld,
                                                         bf16_t* y, float* mean_out, float* rstd_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * WAVES + wave;
    if (row >= M) return;
    const int nchunk = D >> 3;
    float v[MAXV][8];
    load_row<MAXV>(x + (int64_t)row * D, nchunk, lane, v);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { float d = v[i][e] - mean; q += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    const int b = row / rpb;
    const bf16_t* sh = shift + (int64_t)b * mod_ld;
    const bf16_t* sc = scale + (int64_t)b * mod_ld;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float a[8], g[8], o[8];
            unpack8(*reinterpret_cast<const u32x4*>(sh + c * 8), a);
            unpack8(*reinterpret_cast<const u32x4*>(sc + c * 8), g);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = rbf((v[i][e] - mean) * rstd);      // F.layer_norm output (bf16)
                o[e] = rbf(xh * rbf(1.0f + g[e])) + a[e];           // * (1 + scale) + shift
            }
            *reinterpret_cast<u32x4*>(y + (int64_t)row * D + c * 8) = pack8(o);
        }
    }
}

// ------------------------------------------------------------------ LN + modulate backward
// Two light kernels instead of one register-heavy one (the fused version held x, dy, (1+scale) and both column
// accumulators per lane: ~230 VGPRs, 1-2 waves/SIMD, 2 TB/s):
//  (a) row pass, one wave per row, two streaming sweeps (the second one hits L1/L2): dx = dres + LN'(dy * (1+scale));
//  (b) column pass (strip layout: a lane owns 8 columns and walks 32 rows, row statistics are wave-uniform scalars):
//      dshift += sum_n dy, dscale += sum_n dy * bf16(xhat).  x and dy come back from L2 / Infinity Cache.
template <int MAXV>
__global__ __launch_bounds__(256) void ln_mod_bwd_rows_kernel(int M, int D, int rpb, const bf16_t* x, const float* mean_in,
                                                              const float* rstd_in, const bf16_t* scale, int mod_ld,
                                                              const bf16_t* dy, const bf16_t* dres, bf16_t* dx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * WAVES + wave;
    if (row >= M) return;
    const int nchunk = D >> 3;
    const bf16_t* sc = scale + (int64_t)(row / rpb) * mod_ld;
    const bf16_t* xr = x + (int64_t)row * D;
    const bf16_t* gr = dy + (int64_t)row * D;
    const float mean = mean_in[row], rstd = rstd_in[row];
    // pass 1: the two row reductions (streamed, nothing kept)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float xv[8], gv[8], sv[8];
            unpack8(*reinterpret_cast<const u32x4*>(xr + c * 8), xv);
            unpack8(*reinterpret_cast<const u32x4*>(gr + c * 8), gv);
            unpack8(*reinterpret_cast<const u32x4*>(sc + c * 8), sv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (xv[e] - mean) * rstd;
                const float gg = rbf(gv[e] * rbf(1.0f + sv[e]));       // grad wrt the LN output (bf16 mul in autograd)
                s1 += gg;
                s2 += gg * xh;
            }
        }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    // pass 2: re-read the row (L1/L2 hit) rather than holding 80 floats across the reduction -- this kernel lives on
    // occupancy.  The empty asm keeps the compiler from merging the two passes' loads.
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float xv[8], gv[8], sv[8], o[8];
            unpack8(*reinterpret_cast<const u32x4*>(xr + c * 8), xv);
            unpack8(*reinterpret_cast<const u32x4*>(gr + c * 8), gv);
            unpack8(*reinterpret_cast<const u32x4*>(sc + c * 8), sv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (xv[e] - mean) * rstd;
                const float gg = rbf(gv[e] * rbf(1.0f + sv[e]));
                o[e] = rstd * (gg - s1 - xh * s2);
            }
            if (dres) {
                float r[8];
                unpack8(*reinterpret_cast<const u32x4*>(dres + (int64_t)row * D + c * 8), r);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rbf(o[e]) + r[e];
            }
            *reinterpret_cast<u32x4*>(dx + (int64_t)row * D + c * 8) = pack8(o);
        }
    }
}

// grid = (ceil(D/512), ceil(rpb / (WAVES * LN_COL_ROWS)), B); one partial row per workgroup: ws[(b*gridDim.y + by)][2][D].
// Eight rows' loads (x, dy, mean, rstd) are issued before the first use: with one row in flight per wave the 1280 waves of
// a 8192 x 2240 gradient kept 2.6 MB in flight, a quarter of what the HBM pipe needs (3.7 TB/s measured).
constexpr int LN_COL_ROWS = 16;
__global__ __launch_bounds__(256) void ln_mod_bwd_cols_kernel(int rpb, int D, const bf16_t* x, const float* mean_in,
                                                              const float* rstd_in, const bf16_t* dy, float* ws) {
    __shared__ float red[2 * WAVES * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 512 + lane * 8;
    const int b = blockIdx.z;
    const int rg = blockIdx.y * WAVES + wave;
    float a1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c0 < D) {
        for (int r0 = 0; r0 < LN_COL_ROWS; r0 += 8) {
            u32x4 vx[8], vg[8];
            float mean[8], rstd[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int rl = rg * LN_COL_ROWS + r0 + q;
                const bool in = rl < rpb;
                const int64_t row = (int64_t)b * rpb + (in ? rl : 0);
                vx[q] = *reinterpret_cast<const u32x4*>(x + row * D + c0);
                vg[q] = in ? *reinterpret_cast<const u32x4*>(dy + row * D + c0) : u32x4{0u, 0u, 0u, 0u};
                mean[q] = mean_in[row];
                rstd[q] = rstd_in[row];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float xv[8], gv[8];
                unpack8(vx[q], xv);
                unpack8(vg[q], gv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    a1[e] += gv[e];
                    a2[e] += gv[e] * rbf((xv[e] - mean[q]) * rstd[q]);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[wave * 512 + lane * 8 + e] = a1[e];
        red[(WAVES + wave) * 512 + lane * 8 + e] = a2[e];
    }
    __syncthreads();
    float* wp = ws + ((int64_t)b * gridDim.y + blockIdx.y) * 2 * D;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int cl = threadIdx.x + 256 * k, c = blockIdx.x * 512 + cl;
        if (c >= D) continue;
        wp[c] = red[cl] + red[512 + cl] + red[1024 + cl] + red[1536 + cl];
        wp[D + c] = red[2048 + cl] + red[2560 + cl] + red[3072 + cl] + red[3584 + cl];
    }
}

// Partial-row reductions: a 256-thread block owns 64 columns; 4 thread rows split the G partial rows and meet in LDS
// (fixed order -> bit-reproducible).  grid.x = ceil(W / 64).
__device__ __forceinline__ float reduce_rows64(const float* ws, int G, int W, int j, bool valid) {
    __shared__ float red[4][64];
    const int part = threadIdx.x >> 6;
    float s = 0.f;
    if (valid)
        for (int g = part; g < G; g += 4) s += ws[(int64_t)g * W + j];
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    return red[0][threadIdx.x & 63] + red[1][threadIdx.x & 63] + red[2][threadIdx.x & 63] + red[3][threadIdx.x & 63];
}

// out[b*ld_out + j] += sum_g ws[(b*G + g)*W + j]   (fp32 accumulate) -- j over W columns
__global__ __launch_bounds__(256) void reduce_partials_f32_kernel(int B, int G, int W, const float* ws, float* out0,
                                                                  float* out1, int half, int ld_out) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), b = blockIdx.y;
    const float s = reduce_rows64(ws + (int64_t)b * G * W, G, W, j, j < W);
    if (threadIdx.x >= 64 || j >= W) return;
    // columns [0, half) go to out0, [half, W) to out1 (dshift / dscale live in different slots)
    if (j < half) out0[(int64_t)b * ld_out + j] += s;
    else if (out1) out1[(int64_t)b * ld_out + (j - half)] += s;
}

// bf16 output variant (bias / weight gradients): out[j] = (accumulate ? out[j] : 0) + sum_g ws[g*W + j]
__global__ __launch_bounds__(256) void reduce_partials_bf16_kernel(int G, int W, const float* ws, bf16_t* out,
                                                                   int accumulate) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    float s = reduce_rows64(ws, G, W, j, j < W);
    if (threadIdx.x >= 64 || j >= W) return;
    if (accumulate) s = rbf(s) + bf2f(out[j]);
    out[j] = f2bf(s);
}

// ------------------------------------------------------------------ RMSNorm
template <int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(int M, int D, float eps, const bf16_t* x, const bf16_t* w,
                                                          bf16_t* y, float* rstd_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * WAVES + wave;
    if (row >= M) return;
    const int nchunk = D >> 3;
    float v[MAXV][8];
    load_row<MAXV>(x + (int64_t)row * D, nchunk, lane, v);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) q += v[i][e] * v[i][e];
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) rstd_out[row] = rstd;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float ww[8], o[8];
            unpack8(*reinterpret_cast<const u32x4*>(w + c * 8), ww);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rbf(v[i][e] * rstd) * ww[e];
            *reinterpret_cast<u32x4*>(y + (int64_t)row * D + c * 8) = pack8(o);
        }
    }
}

// grid.x = ceil(M / 16); partial rows ws[(blk*4 + wave)][D] hold dw partials
template <int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(int M, int D, const bf16_t* x, const bf16_t* w,
                                                          const float* rstd_in, const bf16_t* dy, bf16_t* dx,
                                                          float* ws) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = D >> 3;
    float ww[MAXV][8], aw[MAXV][8];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) unpack8(*reinterpret_cast<const u32x4*>(w + c * 8), ww[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) { aw[i][e] = 0.f; if (c >= nchunk) ww[i][e] = 0.f; }
    }
    const int r_begin = blockIdx.x * (WAVES * ROWS_PER_WAVE) + wave * ROWS_PER_WAVE;
    for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
        const int64_t row = r_begin + rr;
        if (row >= M) break;
        float v[MAXV][8], g[MAXV][8];
        load_row<MAXV>(x + row * D, nchunk, lane, v);
        load_row<MAXV>(dy + row * D, nchunk, lane, g);
        const float rstd = rstd_in[row];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xn = v[i][e] * rstd;
                aw[i][e] += g[i][e] * rbf(xn);
                const float gg = rbf(g[i][e] * ww[i][e]);
                v[i][e] = xn; g[i][e] = gg;
                s += gg * xn;
            }
        s = wave_sum(s) / (float)D;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rstd * (g[i][e] - v[i][e] * s);
                *reinterpret_cast<u32x4*>(dx + row * D + c * 8) = pack8(o);
            }
        }
    }
    float* wp = ws + ((int64_t)blockIdx.x * WAVES + wave) * D;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; e += 4)
                *reinterpret_cast<f32x4*>(wp + c * 8 + e) = f32x4{aw[i][e], aw[i][e + 1], aw[i][e + 2], aw[i][e + 3]};
        }
    }
}

// ------------------------------------------------------------------ strip kernels (512 columns x 32 rows per wave)
constexpr int STRIP_ROWS = 32;            // rows per wave of the column sum
constexpr int GATE_ROWS = 16;             // ... of the gate backward (three streams per row: twice the workgroups)
// MODE 0: column sum of x.  MODE 1: gate backward: dlin = bf16(gate*dout), partial = dout*lin, and (ws2 != null) the
// column sum of dlin, i.e. the bias gradient of the gated Linear, for free.
// The four waves of a workgroup are summed in LDS: one partial row per workgroup (the follow-up reduction reads 4x less).
template <int MODE>
__global__ __launch_bounds__(256) void strip_kernel(int rows_per_batch, int cols, const bf16_t* x, int ld,
                                                    const bf16_t* lin, const bf16_t* gate, int gate_ld, bf16_t* dlin,
                                                    float* ws, float* ws2) {
    // grid = (ceil(cols/512), ceil(rpb / 128), B)
    __shared__ float red[(MODE == 1 ? 2 : 1) * WAVES * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 512 + lane * 8;
    const int b = blockIdx.z;
    const int rg = blockIdx.y * WAVES + wave;                 // row group within the batch
    constexpr int SR = MODE == 1 ? GATE_ROWS : STRIP_ROWS;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, acc2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float g[8];
    if (MODE == 1 && c0 < cols) unpack8(*reinterpret_cast<const u32x4*>(gate + (int64_t)b * gate_ld + c0), g);
    if (MODE == 0 && c0 < cols) {
        // eight rows' loads in flight per lane before the first add (a wave of this kernel is alone on its SIMD more often
        // than not: 320 workgroups on 256 CUs for an 8192 x 2240 gradient); the adds keep the row order, so the sums are
        // bit-identical to the one-row-at-a-time loop
        for (int r0 = 0; r0 < STRIP_ROWS; r0 += 8) {
            u32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int rl = rg * STRIP_ROWS + r0 + q;
                v[q] = rl < rows_per_batch ? *reinterpret_cast<const u32x4*>(x + ((int64_t)b * rows_per_batch + rl) * ld + c0)
                                           : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float a[8];
                unpack8(v[q], a);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += a[e];
            }
        }
    } else if (c0 < cols) {
        // gate backward: the same eight-rows-in-flight shape (x and lin: 16 loads per lane before the first multiply)
        for (int r0 = 0; r0 < SR; r0 += 8) {
            u32x4 va[8], vl[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int rl = rg * SR + r0 + q;
                const int64_t row = (int64_t)b * rows_per_batch + rl;
                const bool in = rl < rows_per_batch;
                va[q] = in ? *reinterpret_cast<const u32x4*>(x + row * ld + c0) : u32x4{0u, 0u, 0u, 0u};
                vl[q] = in ? *reinterpret_cast<const u32x4*>(lin + row * ld + c0) : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int rl = rg * SR + r0 + q;
                if (rl >= rows_per_batch) continue;
                const int64_t row = (int64_t)b * rows_per_batch + rl;
                float a[8], l[8], o[8];
                unpack8(va[q], a);
                unpack8(vl[q], l);
#pragma unroll
                for (int e = 0; e < 8; ++e) { acc[e] += rbf(a[e] * l[e]); o[e] = rbf(g[e] * a[e]); acc2[e] += o[e]; }
                *reinterpret_cast<u32x4*>(dlin + row * ld + c0) = pack8(o);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[wave * 512 + lane * 8 + e] = acc[e];
        if (MODE == 1) red[(WAVES + wave) * 512 + lane * 8 + e] = acc2[e];
    }
    __syncthreads();
    // 256 threads x 2 columns
    const int part = (int)(gridDim.y * blockIdx.z + blockIdx.y);     // partial row: (batch, row block)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int cl = threadIdx.x + 256 * k, c = blockIdx.x * 512 + cl;
        if (c >= cols) continue;
        ws[(int64_t)part * cols + c] = red[cl] + red[512 + cl] + red[1024 + cl] + red[1536 + cl];
        if (MODE == 1 && ws2)
            ws2[(int64_t)part * cols + c] = red[2048 + cl] + red[2560 + cl] + red[3072 + cl] + red[3584 + cl];
    }
}

// ------------------------------------------------------------------ modulation table
__global__ void modulation_fwd_kernel(int B, int S, int D, const bf16_t* table, const bf16_t* tmod, int tmod_ld,
                                      int slot_stride, bf16_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)B * S * D;
    if (i >= total) return;
    const int d = i % D, s = (i / D) % S, b = i / ((int64_t)D * S);
    out[i] = f2bf(bf2f(table[s * D + d]) + bf2f(tmod[(int64_t)b * tmod_ld + s * slot_stride + d]));
}

__global__ void modulation_bwd_kernel(int B, int S, int D, const float* dmod, bf16_t* dtable, int acc_table,
                                      float* dtmod, int tmod_ld, int slot_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over S*D
    if (i >= S * D) return;
    const int d = i % D, s = i / D;
    float t = 0.f;
    for (int b = 0; b < B; ++b) {
        const float g = dmod[((int64_t)b * S + s) * D + d];
        // the reference's autograd hands each consumer a bf16 gradient
        t += rbf(g);
        if (slot_stride != 0) dtmod[(int64_t)b * tmod_ld + s * slot_stride + d] += rbf(g);
    }
    if (acc_table) t = rbf(t) + bf2f(dtable[i]);
    dtable[i] = f2bf(t);
}
// slot_stride == 0 (final norm: every slot adds the same embedded_timestep): dtmod[b, d] += sum_s dmod[b,s,d]
__global__ void modulation_bwd_shared_kernel(int B, int S, int D, const float* dmod, float* dtmod, int tmod_ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over B*D
    if (i >= B * D) return;
    const int d = i % D, b = i / D;
    float t = 0.f;
    for (int s = 0; s < S; ++s) t += rbf(dmod[((int64_t)b * S + s) * D + d]);
    dtmod[(int64_t)b * tmod_ld + d] += t;
}

template <typename F>
int dispatch_maxv(int D, F&& f) {
    if (D & 7) return YAT_EINVAL;
    if (D <= 512 * 2) return f(std::integral_constant<int, 2>{});
    if (D <= 512 * 5) return f(std::integral_constant<int, 5>{});
    if (D <= 512 * 8) return f(std::integral_constant<int, 8>{});
    return YAT_EINVAL;
}

}  // namespace

extern "C" {

int yat_version(void) { return 1; }

int yat_ln_modulate_fwd(int M, int D, int rpb, float eps, const void* x, const void* shift, const void* scale,
                        int mod_ld, void* y, float* mean, float* rstd, yat_stream_t stream) {
    if (M <= 0 || rpb <= 0 || (mod_ld & 7) || !x || !shift || !scale || !y || !mean || !rstd) return YAT_EINVAL;
    return dispatch_maxv(D, [&](auto mv) {
        constexpr int MV = decltype(mv)::value;
        hipLaunchKernelGGL((ln_mod_fwd_kernel<MV>), dim3((M + WAVES - 1) / WAVES), dim3(256), 0, (hipStream_t)stream, M, D,
                           rpb, eps, (const bf16_t*)x, (const bf16_t*)shift, (const bf16_t*)scale, mod_ld, (bf16_t*)y,
                           mean, rstd);
        YAT_CHECK_LAUNCH();
        return YAT_OK;
    });
}

uint64_t yat_ln_bwd_workspace_bytes(int M, int D, int rpb) {
    if (M <= 0 || rpb <= 0) return 0;
    const uint64_t gy = (rpb + WAVES * LN_COL_ROWS - 1) / (WAVES * LN_COL_ROWS);
    return (uint64_t)(M / rpb) * gy * 2 * D * sizeof(float);
}

int yat_ln_modulate_bwd(int M, int D, int rpb, const void* x, const float* mean, const float* rstd, const void* scale,
                        int mod_ld, const void* dy, const void* dres, void* dx, float* dshift_acc, float* dscale_acc,
                        int acc_ld, void* workspace, int parts, yat_stream_t stream) {
    if (M <= 0 || rpb <= 0 || M % rpb || (mod_ld & 7) || !x || !dy || parts < 1 || parts > 3) return YAT_EINVAL;
    if ((parts & 1) && !dx) return YAT_EINVAL;
    if ((parts & 2) && (!workspace || !dshift_acc || !dscale_acc)) return YAT_EINVAL;
    const int B = M / rpb;
    int rc = !(parts & 1) ? YAT_OK : dispatch_maxv(D, [&](auto mv) {
        constexpr int MV = decltype(mv)::value;
        hipLaunchKernelGGL((ln_mod_bwd_rows_kernel<MV>), dim3((M + WAVES - 1) / WAVES), dim3(256), 0, (hipStream_t)stream, M,
                           D, rpb, (const bf16_t*)x, mean, rstd, (const bf16_t*)scale, mod_ld, (const bf16_t*)dy,
                           (const bf16_t*)dres, (bf16_t*)dx);
        YAT_CHECK_LAUNCH();
        return YAT_OK;
    });
    if (rc || !(parts & 2)) return rc;
    const int gy = (rpb + WAVES * LN_COL_ROWS - 1) / (WAVES * LN_COL_ROWS);
    hipLaunchKernelGGL(ln_mod_bwd_cols_kernel, dim3((D + 511) / 512, gy, B), dim3(256), 0, (hipStream_t)stream, rpb, D,
                       (const bf16_t*)x, mean, rstd, (const bf16_t*)dy, (float*)workspace);
    YAT_CHECK_LAUNCH();
    const int W = 2 * D, G = gy;
    hipLaunchKernelGGL(reduce_partials_f32_kernel, dim3((W + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, B, G, W,
                       (const float*)workspace, dshift_acc, dscale_acc, D, acc_ld);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

int yat_rmsnorm_fwd(int M, int D, float eps, const void* x, const void* w, void* y, float* rstd, yat_stream_t stream) {
    if (M <= 0 || !x || !w || !y || !rstd) return YAT_EINVAL;
    return dispatch_maxv(D, [&](auto mv) {
        constexpr int MV = decltype(mv)::value;
        hipLaunchKernelGGL((rmsnorm_fwd_kernel<MV>), dim3((M + WAVES - 1) / WAVES), dim3(256), 0, (hipStream_t)stream, M,
                           D, eps, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd);
        YAT_CHECK_LAUNCH();
        return YAT_OK;
    });
}

uint64_t yat_rmsnorm_bwd_workspace_bytes(int M, int D) {
    return (uint64_t)((M + WAVES * ROWS_PER_WAVE - 1) / (WAVES * ROWS_PER_WAVE)) * WAVES * D * sizeof(float);
}

int yat_rmsnorm_bwd(int M, int D, const void* x, const void* w, const float* rstd, const void* dy, void* dx,
                    void* dw, int accumulate_dw, void* workspace, yat_stream_t stream) {
    if (M <= 0 || !x || !w || !rstd || !dy || !dx || !dw || !workspace) return YAT_EINVAL;
    const int nblk = (M + WAVES * ROWS_PER_WAVE - 1) / (WAVES * ROWS_PER_WAVE);
    int rc = dispatch_maxv(D, [&](auto mv) {
        constexpr int MV = decltype(mv)::value;
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<MV>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, M, D,
                           (const bf16_t*)x, (const bf16_t*)w, rstd, (const bf16_t*)dy, (bf16_t*)dx, (float*)workspace);
        YAT_CHECK_LAUNCH();
        return YAT_OK;
    });
    if (rc) return rc;
    // rows past M never wrote their partial row: zero-filled? no -- every wave writes its (possibly zero) partials
    hipLaunchKernelGGL(reduce_partials_bf16_kernel, dim3((D + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                       nblk * WAVES, D, (const float*)workspace, (bf16_t*)dw, accumulate_dw);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

uint64_t yat_colsum_workspace_bytes(int rows, int cols) {
    const int gy = (rows + WAVES * STRIP_ROWS - 1) / (WAVES * STRIP_ROWS);
    return (uint64_t)gy * cols * sizeof(float);
}

int yat_colsum_bf16(int rows, int cols, const void* x, int ld, void* out, int accumulate, void* workspace,
                    yat_stream_t stream) {
    if (rows <= 0 || cols <= 0 || (cols & 7) || (ld & 7) || !x || !out || !workspace) return YAT_EINVAL;
    const int gy = (rows + WAVES * STRIP_ROWS - 1) / (WAVES * STRIP_ROWS);
    hipLaunchKernelGGL((strip_kernel<0>), dim3((cols + 511) / 512, gy, 1), dim3(256), 0, (hipStream_t)stream, rows, cols,
                       (const bf16_t*)x, ld, (const bf16_t*)nullptr, (const bf16_t*)nullptr, 0, (bf16_t*)nullptr,
                       (float*)workspace, (float*)nullptr);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_partials_bf16_kernel, dim3((cols + 63) / 64), dim3(256), 0, (hipStream_t)stream, gy, cols,
                       (const float*)workspace, (bf16_t*)out, accumulate);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

uint64_t yat_gate_bwd_workspace_bytes(int M, int D, int rpb) {
    if (M <= 0 || rpb <= 0) return 0;
    const uint64_t gy = (rpb + WAVES * GATE_ROWS - 1) / (WAVES * GATE_ROWS);
    return 2 * (uint64_t)(M / rpb) * gy * D * sizeof(float);        // gate partials + bias-gradient partials
}

int yat_gate_bwd(int M, int D, int rpb, const void* dout, const void* lin, const void* gate, int gate_ld, void* dlin,
                 float* dgate_acc, int acc_ld, void* dbias, int accumulate_bias, void* workspace, yat_stream_t stream) {
    if (M <= 0 || rpb <= 0 || M % rpb || (D & 7) || (gate_ld & 7) || !dout || !lin || !gate || !dlin || !dgate_acc ||
        !workspace)
        return YAT_EINVAL;
    const int B = M / rpb;
    const int gy = (rpb + WAVES * GATE_ROWS - 1) / (WAVES * GATE_ROWS);
    float* ws2 = dbias ? (float*)workspace + (int64_t)B * gy * D : nullptr;
    hipLaunchKernelGGL((strip_kernel<1>), dim3((D + 511) / 512, gy, B), dim3(256), 0, (hipStream_t)stream, rpb, D,
                       (const bf16_t*)dout, D, (const bf16_t*)lin, (const bf16_t*)gate, gate_ld, (bf16_t*)dlin,
                       (float*)workspace, ws2);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_partials_f32_kernel, dim3((D + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, B, gy, D,
                       (const float*)workspace, dgate_acc, (float*)nullptr, D, acc_ld);
    YAT_CHECK_LAUNCH();
    if (dbias) {                                              // bias gradient of the gated Linear = column sum of dlin
        hipLaunchKernelGGL(reduce_partials_bf16_kernel, dim3((D + 63) / 64), dim3(256), 0, (hipStream_t)stream, B * gy, D,
                           (const float*)ws2, (bf16_t*)dbias, accumulate_bias);
        YAT_CHECK_LAUNCH();
    }
    return YAT_OK;
}

int yat_modulation_fwd(int B, int S, int D, const void* table, const void* tmod, int tmod_ld, int slot_stride,
                       void* mod_out, yat_stream_t stream) {
    if (B <= 0 || S <= 0 || D <= 0 || !table || !tmod || !mod_out) return YAT_EINVAL;
    const int64_t total = (int64_t)B * S * D;
    hipLaunchKernelGGL(modulation_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B,
                       S, D, (const bf16_t*)table, (const bf16_t*)tmod, tmod_ld, slot_stride, (bf16_t*)mod_out);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

int yat_modulation_bwd(int B, int S, int D, const float* dmod, void* dtable, int accumulate_table, float* dtmod_acc,
                       int tmod_ld, int slot_stride, yat_stream_t stream) {
    if (B <= 0 || S <= 0 || D <= 0 || !dmod || !dtable || !dtmod_acc) return YAT_EINVAL;
    hipLaunchKernelGGL(modulation_bwd_kernel, dim3((S * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, S, D, dmod,
                       (bf16_t*)dtable, accumulate_table, dtmod_acc, tmod_ld, slot_stride);
    YAT_CHECK_LAUNCH();
    if (slot_stride == 0) {
        hipLaunchKernelGGL(modulation_bwd_shared_kernel, dim3((B * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, B,
                           S, D, dmod, dtmod_acc, tmod_ld);
        YAT_CHECK_LAUNCH();
    }
    return YAT_OK;
}

}  // extern "C"
