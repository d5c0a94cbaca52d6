// GLUMBConv middle for gfx950, on the token-major (channels-last) layout:
//   s = SiLU(z) [fused into the conv_inverted GEMM epilogue, which stores both s and z];
//   u = dwconv3x3(s) + bias;  y = u[:Hc] * SiLU(u[Hc:])
// (diffusers GLUMBConv as called at /root/reference/utils/patch_sana_attention_layers.py:110-113;
// z is the conv_inverted output, so the reference's NCHW permutes at :110,112 disappear).
// Backward = two passes: (1) recompute u, emit du (bf16); (2) transposed conv of du -> dz (times SiLU'(z)) with the
// weight / bias gradient partials accumulated in registers, reduced per workgroup in LDS, then by a small kernel.
//
// Two generations of kernels live here:
//  * LDS-tiled kernels (dwglu_tile_kernel<0/1>, dwglu_bwd2_tile_kernel) for w <= 64 -- the product path at every SANA
//    aspect bucket.  A workgroup stages (R+2) image rows x w columns x 32 channels with LDS-DMA and computes from LDS;
//    a thread owns 4 channels of a run of SEG output columns and walks it with three statically rotated accumulators,
//    so each tap is applied exactly once and nothing is shuffled between registers.  Measured (B=8, 32x32, Hc=5600):
//    forward ~89 us, backward ~350 us; PMC shows the forward at ~75 % VALU utilisation (unpack + FMA + SiLU), i.e. these
//    are VALU-bound at ~3 TB/s of algorithmic traffic, not HBM-bound.
//  * a streaming forward (dwglu_stream_kernel, dwconv_tile_fwd.inc) for the shapes whose rows fill its 16 run slots (32 x 32):
//    ring of 2R+2 tile rows, the next R rows' LDS-DMA under the current rows' arithmetic; same tap order -> same bits.
//  * direct kernels (dwconv_glu_kernel<0/1>, dwconv_bwd2_kernel) for wider images: lanes along channels (8 B per lane),
//    one guarded global load per (row, column, half), register double buffer.  ~25 % slower; kept as the general path.
#include "common.hpp"
#include "../../include/yat_hip.h"
#include <cstdlib>

namespace {

#ifndef YAT_DW_SEG
#define YAT_DW_SEG 8
#endif
constexpr int SEG = YAT_DW_SEG;      // output columns per thread
constexpr int PK = 11;      // partial values per channel: 9 taps, conv bias, column sum of dz
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
                                                         bf16_t* out, bf16_t* u_out) {
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
                if (u_out) {
                    *reinterpret_cast<u32x2*>(u_out + pix * C2 + ca) = pack4(ua[0], ua[1], ua[2], ua[3]);
                    *reinterpret_cast<u32x2*>(u_out + pix * C2 + cg) = pack4(ug[0], ug[1], ug[2], ug[3]);
                }
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

inline int nseg_of(int w) { return (w + SEG - 1) / SEG; }
inline int nsb_of(int w) { return (w + 4 * SEG - 1) / (4 * SEG); }
inline int nrg_of(int h) { return (h + ROWS - 1) / ROWS; }
inline unsigned grid8(int64_t total) { return (unsigned)(((total + 7) / 8) * 8); }

// ------------------------------------------------------------------------------------------------------------------
// LDS-tiled version (w <= 64, Hc % 8 == 0): the kernels above issue one guarded 8-byte global load per (row, column,
// half) and keep the loads in flight in VGPRs, which caps occupancy at 3 waves/SIMD and leaves them latency-bound at
// ~2.5 TB/s.  Here a workgroup stages a whole (R+2)-row x w-column x 32-channel tile of both halves with LDS-DMA
// (16 B/lane, no VGPRs, zero fill at the image border from the buffer range check) and computes from LDS.
//   tile[half][(R+2) rows][WP = w+1 columns][32 ch]: column 0 of a row is the zero left halo AND (being the element after
//   column w of the previous row) the zero right halo, so no second halo column is stored.  WP is odd for even w:
//   consecutive rows start 64 B (mod 256 B) apart and the four runs one ds_read_b64 serves per cycle hit disjoint banks.
//   9 pad pixels after each half absorb the over-read of a partial last column segment.
// A thread owns 4 channels of one run = SEG output columns of one row; the column loop is fully unrolled with the three
// rolling accumulators renamed statically (no register shuffling), every tap is applied exactly once.
// MODE 0: forward (writes y).  MODE 1: backward pass 1 (reads dy, writes du for both halves).
// 9 taps of 4 consecutive channels = 36 contiguous bf16 (8-byte aligned since the channel index is a multiple of 4):
// nine 8-byte loads issued together, then regrouped as channel pairs per tap
__device__ __forceinline__ void load_taps(const bf16_t* wdw, int c, f32x2 (&wv)[9][2]) {
    u32x2 raw[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) raw[k] = *reinterpret_cast<const u32x2*>(wdw + (int64_t)c * 9 + k * 4);
    float flat[36];
#pragma unroll
    for (int k = 0; k < 9; ++k) unpack4(raw[k], flat + 4 * k);      // flat[e*9 + t]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) wv[t][pr] = f32x2{flat[(2 * pr) * 9 + t], flat[(2 * pr + 1) * 9 + t]};
    // opaque to the optimizer (after ALL loads): otherwise it keeps the packed words and re-unpacks each weight in the run loop
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) asm volatile("" : "+v"(wv[t][pr]));
}

// The forward tile kernel in four (channels per tile, workgroups per CU) variants; results are bit-identical across them
// (every output's taps are applied in the same order), so the choice per shape is pure scheduling.
#define DW_TCH 32
#define DW_WPS 3
namespace v32w3 {
#include "dwconv_tile_fwd.inc"
}
#undef DW_TCH
#undef DW_WPS
#define DW_TCH 64
#define DW_WPS 3
namespace v64w3 {
#include "dwconv_tile_fwd.inc"
}
#undef DW_WPS
#define DW_WPS 2
namespace v64w2 {
#include "dwconv_tile_fwd.inc"
}
#undef DW_TCH
#undef DW_WPS
using namespace v32w3;      // pass 2 below shares this variant's staging helpers and tile constants

// Forward: 128-byte pixel slices always; two workgroups per CU (taller bands) for wide rows.  Measured (B = 8, Hc = 5600,
// forward with the u store, us): 32x32 154 -> 119, 16x64 165 -> 140, 24x42 176 -> 136, 44x22 186 -> 123.
// Pass 1 (only when u is not kept) stays on the 32-channel variant it was tuned on.
template <int MODE>
int launch_tile_best(int B, int h, int w, int Hc, const bf16_t* s, const bf16_t* wdw, const bf16_t* bdw, const bf16_t* dy,
                     bf16_t* out, bf16_t* u_out, hipStream_t stream) {
    static const int force = YAT_TUNE_INT("YAT_DW_VARIANT", 0);     // 1: v32w3, 2: v64w3, 3: v64w2
    int v = MODE == 0 ? (w > 48 ? 3 : 2) : 1;
    if (force) v = force;
    static const int stream_on = YAT_TUNE_INT("YAT_DW_STREAM", 1);
    if (MODE == 0 && stream_on && v64w3::launch_stream(B, h, w, Hc, s, wdw, bdw, out, u_out, stream) == 0) return 0;
    if (v == 3 && v64w2::launch_tile<MODE>(B, h, w, Hc, s, wdw, bdw, dy, out, u_out, stream) == 0) return 0;
    if (v >= 2 && v64w3::launch_tile<MODE>(B, h, w, Hc, s, wdw, bdw, dy, out, u_out, stream) == 0) return 0;
    return v32w3::launch_tile<MODE>(B, h, w, Hc, s, wdw, bdw, dy, out, u_out, stream);
}

// LDS-tiled backward pass 2 (same staging scheme; all 2*Hc channels are independent here, 32 per workgroup):
//   du tile: rows i0-1 .. i0+R (zero outside the image), s and z tiles: rows i0 .. i0+R-1, all with the shared zero column.
//   dz[i,j] = SiLU'(z[i,j]) * bf16( sum_taps W[tap] du[i-di, j-dj] );  dW[tap] += s[i,j] * du[i-di, j-dj];  db += du[i,j]
// The thread's dW/db registers are summed over the 32 run slots through the (then free) tile memory: one partial row
// per workgroup, ws[(b*nbands + band)][2Hc*10].
// (two workgroups per CU; sized for three -- 53 KB tiles, which the kernel's 166 registers would allow -- it is 5 .. 8 % slower:
// shorter bands re-read more halo; profiles/r05_q_*)
__global__ __launch_bounds__(256, 2) void dwglu_bwd2_tile_kernel(int h, int w, int Hc, int B, int R, int rmagic, int nbands,
                                                                 int bpb, int nchunk, const bf16_t* sact, const bf16_t* z,
                                                                 const bf16_t* du, uint64_t bytes, const bf16_t* wdw,
                                                                 bf16_t* dz, float* ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tile[];
    const int ngrp = (nbands + bpb - 1) / bpb;
    const int total = ngrp * nchunk * B;
    int u = xcd_unit(total);
    if (u >= total) return;
    const int bg = u % ngrp; u /= ngrp;
    const int cx = u % nchunk, b = u / nchunk;
    const int ch0 = cx * TCH, C2 = 2 * Hc;
    const int WP = w + 1;
    const int PHd = ((R + 2) * WP + 1 + TILE_PAD + 15) & ~15, PHc = (R * WP + 1 + TILE_PAD + 15) & ~15;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6, nslots = blockDim.x / NCG;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    clear_tile(tile, (PHd + PHc) * TCH * 2);
    const int cg = lane & (NCG - 1);
    const int c0 = ch0 + cg * 4;
    const bool chan_ok = c0 < C2;
    // channel pairs as explicit 2-vectors: every multiply-add below is one v_pk_fma_f32 with a fixed register pairing
    // (left to the SLP vectorizer, taps get paired across different weights and the weight set is kept twice)
    f32x2 wt[9][2], dW[9][2], db[2], dzs[2];                 // dzs: column sum of dz = bias gradient of conv_inverted
    {
        const int cs = chan_ok ? c0 : 0;
        load_taps(wdw, cs, wt);
#pragma unroll
        for (int t = 0; t < 9; ++t) dW[t][0] = dW[t][1] = f32x2{0.f, 0.f};
        db[0] = db[1] = dzs[0] = dzs[1] = f32x2{0.f, 0.f};
    }
    const int nseg = (w + SEG - 1) / SEG, nruns = R * nseg;
    const int slot = threadIdx.x / NCG;
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(du, bytes), rz = make_rsrc(z, bytes);
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(dz, bytes);
    const bool lane_ch_ok = ch0 + (lane % PPP) * 8 < C2;

  for (int rb = bg * bpb; rb < min(nbands, (bg + 1) * bpb); ++rb) {
    const int i0 = rb * R;
    __syncthreads();
    stage_rows(rd, tile, 0, R + 2, i0 - 1, b, h, w, WP, C2, ch0, lane_ch_ok, wave_s, nwaves, lane);
    stage_rows(rz, tile, PHd, R, i0, b, h, w, WP, C2, ch0, lane_ch_ok, wave_s, nwaves, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int run = slot; run < nruns; run += nslots) {
        const int seg = (run * rmagic) >> 16, row = run - seg * R;
        const int i = i0 + row;
        if (i >= h || !chan_ok) continue;
        const int j0 = seg * SEG;
        const unsigned char* pd = tile + ((row * WP + j0) * TCH + cg * 4) * 2;                 // du rows row .. row+2
        const unsigned char* pz = tile + ((PHd + row * WP + j0 + 1) * TCH + cg * 4) * 2;       // z at output column j0
        const int64_t pix0 = ((int64_t)b * h + i) * w + j0;
        f32x2 acc[3][2], S[3][2], SG[3][2], ZC[3][2];
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[m][0] = acc[m][1] = f32x2{0.f, 0.f};
        u32x2 nxt[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) nxt[r] = *reinterpret_cast<const u32x2*>(pd + (r * WP) * TCH * 2);
#pragma unroll
        for (int t = 0; t < SEG + 2; ++t) {                 // du column j0 - 1 + t
            u32x2 cur[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) cur[r] = nxt[r];
            if (t + 1 < SEG + 2) {
#pragma unroll
                for (int r = 0; r < 3; ++r) nxt[r] = *reinterpret_cast<const u32x2*>(pd + (r * WP + t + 1) * TCH * 2);
            }
            if (t < SEG) {                                  // s of output column t (zero past the image: it must not count)
                // s = bf16(z sigmoid(z)) is what the conv_inverted GEMM stored (gemm_common.hpp: silu_f on the rounded z, rounded
                // again by the store).  Round 5: recomputed here, bit for bit, from the z this pass reads anyway -- the sigmoid is
                // kept for the SiLU' of the same column two iterations on -- so s is not read at all: a quarter of the pass's
                // bytes (181 -> 163 us at 32 x 32, 239 -> 175 at 16 x 64; profiles/r05_o_*).
                unpack22(*reinterpret_cast<const u32x2*>(pz + t * TCH * 2), ZC[t % 3]);
                const bool in_img = j0 + t < w;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float zz = ZC[t % 3][pr][e], sg = sigmoid_f(zz);
                        SG[t % 3][pr][e] = sg;
                        S[t % 3][pr][e] = in_img ? rbf(zz * sg) : 0.f;
                    }
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                f32x2 d[2];
                unpack22(cur[r], d);
                const int tr = (2 - r) * 3;                 // du row i + r - 1 -> tap row 2 - r
#pragma unroll
                for (int tc = 0; tc < 3; ++tc) {            // output o = t + tc - 2 takes tap column tc
                    const int o = t + tc - 2;
                    if (o < 0 || o >= SEG) continue;
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        acc[o % 3][pr] += wt[tr + tc][pr] * d[pr];
                        dW[tr + tc][pr] += S[o % 3][pr] * d[pr];
                    }
                }
                if (r == 1 && t >= 1 && t <= SEG) {         // du[i, j0 + t - 1]: the bias gradient; past the image edge the
                    const float m = j0 + t - 1 < w ? 1.f : 0.f;                      // tile wraps to real data: mask it
                    db[0] += m * d[0];
                    db[1] += m * d[1];
                }
            }
            const int o = t - 2;
            if (o >= 0) {
                float ds[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) ds[e] = dsilu_from_sigmoid(ZC[o % 3][e >> 1][e & 1], SG[o % 3][e >> 1][e & 1]);
                const f32x2 a0 = acc[o % 3][0], a1 = acc[o % 3][1];
                const u32x2 v = pack4(rbf(a0[0]) * ds[0], rbf(a0[1]) * ds[1], rbf(a1[0]) * ds[2], rbf(a1[1]) * ds[3]);
                acc[o % 3][0] = acc[o % 3][1] = f32x2{0.f, 0.f};
                const bool live = j0 + o < w;
                f32x2 vz[2];
                unpack22(live ? v : u32x2{0u, 0u}, vz);     // the rounded values, as a later column sum over dz would see them
                dzs[0] += vz[0];
                dzs[1] += vz[1];
                __builtin_amdgcn_raw_buffer_store_b64(v, rout, live ? (uint32_t)(((pix0 + o) * C2 + c0) * 2) : YAT_OOB, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
  }
    // ---- sum the run slots through LDS: red[slot][cg*4*PK + e*PK + k]; one partial row per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(tile);
    {
        float* mine = red + slot * (NCG * 4 * PK) + cg * 4 * PK;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int t = 0; t < 9; ++t) mine[e * PK + t] = dW[t][e >> 1][e & 1];
            mine[e * PK + 9] = db[e >> 1][e & 1];
            mine[e * PK + 10] = dzs[e >> 1][e & 1];
        }
    }
    __syncthreads();
    float* wp = ws + ((int64_t)b * ngrp + bg) * C2 * PK + (int64_t)ch0 * PK;
    const int nvalid = min(TCH, C2 - ch0) * PK;
    for (int idx = threadIdx.x; idx < nvalid; idx += blockDim.x) {
        float t = 0.f;
        for (int sl = 0; sl < nslots; ++sl) t += red[sl * (NCG * 4 * PK) + idx];
        wp[idx] = t;
    }
}

// tile geometry for pass 2 (R rows per band, two workgroups per CU)
inline int pick_band_rows_bwd2(int h, int w, size_t* lds_bytes, int* threads) {
    const int nseg = (w + SEG - 1) / SEG, WP = w + 1;
    int best = 0;
    double best_score = 0;
    *threads = 256;
    const int ns = *threads / NCG;
    for (int R = 4; R <= 16 && R <= ((h + 1) & ~1); ++R) {     // >= ROWS: the partial rows fit the workspace
        const int PHd = ((R + 2) * WP + 1 + TILE_PAD + 15) & ~15, PHc = (R * WP + 1 + TILE_PAD + 15) & ~15;
        size_t bytes = (size_t)(PHd + PHc) * TCH * 2;
        if (bytes < ns * NCG * 4 * PK * sizeof(float)) bytes = ns * NCG * 4 * PK * sizeof(float);
        if (bytes > 65536) break;
        const int nruns = R * nseg, passes = (nruns + ns - 1) / ns;
        const int nb = (h + R - 1) / R;
        // du is a third of the traffic: its halo re-read weighs a third
        const double score = (double)nruns / (passes * ns) * (3.0 * R / (3.0 * R + 2.0)) * h / (nb * R);
        if (score > best_score) { best_score = score; best = R; *lds_bytes = bytes; }
    }
    return best;
}

// backward pass 2.  Block = 64 channel groups (4 channels each, either half) x 4 column segments;
// thread = rows [i0, i0+ROWS) x columns [j0, j0+SEG) of image b.
//   dz[i,j] = SiLU'(z[i,j]) * bf16( sum_taps W[tap] du[i-di, j-dj] )
//   dW[tap] += s(z[i,j]) * du[i-di, j-dj];   db += du[i,j]
// partials: ws[((b*nrg + rg)*nsb + sb)][2Hc*10]  (10 = 9 taps + bias per channel)
__global__ __launch_bounds__(256) void dwconv_bwd2_kernel(int h, int w, int Hc, int nx, int nrg, int B, const bf16_t* sact, const bf16_t* z,
                                                          const bf16_t* wdw, const bf16_t* du, bf16_t* dz, float* ws) {
    __shared__ float red[4][64][4 * PK + 1];
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
    (void)sact;
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
            // s = bf16(z sigmoid(z)), recomputed from z like the band / global-z kernels do (round 6: one definition of s for
            // every shape, the `s` argument is not read by pass 2 at all; zero past the image, where z loads as zero)
            auto s_of = [&](u32x2 zz, float (&o)[4]) {
                unpack4(zz, o);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rbf(o[e] * sigmoid_f(o[e]));
            };
            float aP[4], aC[4], aN[4], sP[4], sC[4], sN[4], zP[4], zC[4], zN[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { aP[e] = aC[e] = aN[e] = 0.f; }
            // centre-row z window around input column jj: P = jj-1, C = jj, N = jj+1
            unpack4(load_z(j0 - 2), zP);
            unpack4(load_z(j0 - 1), zC);
            s_of(load_z(j0 - 2), sP);
            s_of(load_z(j0 - 1), sC);
            u32x2 dn[3];
            load_du(j0 - 1, dn);
            u32x2 zn = load_z(j0);
            for (int jj = j0 - 1; jj <= j1; ++jj) {
                u32x2 dc[3] = {dn[0], dn[1], dn[2]};
                unpack4(zn, zN);
                s_of(zn, sN);
                if (jj < j1) { load_du(jj + 1, dn); zn = load_z(jj + 2); }      // prefetch
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
        for (int t = 0; t < 9; ++t) red[lseg][lg][e * PK + t] = dW[t][e];
        red[lseg][lg][e * PK + 9] = db[e];
        red[lseg][lg][e * PK + 10] = 0.f;                    // dz column sum: taken by a separate pass on this path
    }
    __syncthreads();
    if (lseg == 0 && active) {
        float* wp = ws + (((int64_t)b * nrg * nsb + rg * nsb + sb)) * C2 * PK + (int64_t)c0 * PK;
#pragma unroll
        for (int k = 0; k < 4 * PK; ++k) wp[k] = red[0][lg][k] + red[1][lg][k] + red[2][lg][k] + red[3][lg][k];
    }
}

__global__ void dwconv_reduce_kernel(int P, int C2, const float* ws, bf16_t* dw, bf16_t* dbias, bf16_t* dzsum,
                                     int accumulate) {
    __shared__ float red[4][64];
    const int idx = blockIdx.x * 64 + (threadIdx.x & 63);    // over C2*PK
    const int part = threadIdx.x >> 6;
    float s = 0.f;
    if (idx < C2 * PK)
        for (int p = part; p < P; p += 4) s += ws[(int64_t)p * C2 * PK + idx];
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part != 0 || idx >= C2 * PK) return;
    s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    const int ch = idx / PK, t = idx % PK;
    bf16_t* dst = t < 9 ? dw + ch * 9 + t : (t == 9 ? dbias + ch : (dzsum ? dzsum + ch : nullptr));
    if (!dst) return;
    if (accumulate) s = rbf(s) + bf2f(*dst);
    *dst = f2bf(s);
}


}  // namespace

extern "C" {

int yat_dwconv_glu_fwd(int B, int h, int w, int Hc, const void* s, const void* wdw, const void* bdw, void* y,
                       void* u_out, yat_stream_t stream) {
    const void* z = s;
    if (B <= 0 || h <= 0 || w <= 0 || Hc <= 0 || (Hc & 3) || !z || !wdw || !bdw || !y) return YAT_EINVAL;
    if ((int64_t)h * nseg_of(w) * B * ((Hc / 4 + 63) / 64) * 2 > 0x7fffff00ll) return YAT_EINVAL;
    if (w <= 64 && launch_tile_best<0>(B, h, w, Hc, (const bf16_t*)z, (const bf16_t*)wdw, (const bf16_t*)bdw, nullptr,
                                  (bf16_t*)y, (bf16_t*)u_out, (hipStream_t)stream) == 0) {
        YAT_CHECK_LAUNCH();
        return YAT_OK;
    }
    const int nx = (Hc / 4 + 255) / 256;
    hipLaunchKernelGGL((dwconv_glu_kernel<0>), dim3(grid8((int64_t)nx * h * nseg_of(w) * B)), dim3(256), 0,
                       (hipStream_t)stream, h, w, Hc, nseg_of(w), nx, B, (const bf16_t*)z, (const bf16_t*)wdw,
                       (const bf16_t*)bdw, (const bf16_t*)nullptr, (bf16_t*)y, (bf16_t*)u_out);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

uint64_t yat_dwconv_glu_bwd_workspace_bytes(int B, int h, int w, int Hc) {
    // du (bf16 [B,h,w,2Hc]) followed by the fp32 partials
    const uint64_t du_bytes = ((uint64_t)B * h * w * 2 * Hc * 2 + 255) & ~255ull;
    return du_bytes + (uint64_t)B * nrg_of(h) * nsb_of(w) * 2 * Hc * PK * sizeof(float);
}

int yat_dwconv_glu_bwd(int B, int h, int w, int Hc, const void* s, const void* z, const void* wdw, const void* bdw,
                       const void* dy, void* dz, void* dwdw, void* dbdw, void* dz_colsum, int accumulate, void* workspace,
                       const void* du_in, yat_stream_t stream) {
    if (B <= 0 || h <= 0 || w <= 0 || Hc <= 0 || (Hc & 3) || !s || !z || !wdw || !bdw || (!dy && !du_in) || !dz || !dwdw ||
        !dbdw || !workspace)
        return YAT_EINVAL;
    if ((int64_t)h * nseg_of(w) * B * ((Hc / 4 + 63) / 64) * 2 > 0x7fffff00ll) return YAT_EINVAL;
    const int C2 = 2 * Hc;
    const int gy2 = nrg_of(h) * nsb_of(w);
    const bf16_t* du = du_in ? (const bf16_t*)du_in : (const bf16_t*)workspace;
    const uint64_t du_bytes = ((uint64_t)B * h * w * C2 * 2 + 255) & ~255ull;
    float* ws = (float*)((char*)workspace + du_bytes);
    const int nx = (Hc / 4 + 255) / 256, nx2 = (C2 / 4 + 63) / 64;
    if (!du_in) {                                             // pass 1: recompute u, GLU backward -> du (workspace)
        bf16_t* duw = (bf16_t*)workspace;
        if (!(w <= 64 && launch_tile_best<1>(B, h, w, Hc, (const bf16_t*)s, (const bf16_t*)wdw, (const bf16_t*)bdw,
                                        (const bf16_t*)dy, duw, nullptr, (hipStream_t)stream) == 0))
            hipLaunchKernelGGL((dwconv_glu_kernel<1>), dim3(grid8((int64_t)nx * h * nseg_of(w) * B)), dim3(256), 0,
                               (hipStream_t)stream, h, w, Hc, nseg_of(w), nx, B, (const bf16_t*)s, (const bf16_t*)wdw,
                               (const bf16_t*)bdw, (const bf16_t*)dy, duw, (bf16_t*)nullptr);
        YAT_CHECK_LAUNCH();
    }
    size_t lds2 = 0;
    int threads2 = 256;
    // pass 2 variants: 64 channels per tile with s, z straight from global (128-byte pixel slices everywhere; 1), or the
    // 32-channel kernel with all three operands tiled in LDS (2); YAT_DW_BWD2=2 forces the latter
    static const int bwd2_force = YAT_TUNE_INT("YAT_DW_BWD2", 0);
    // measured again after the s read went away (round 5, B = 8, Hc = 5600, us, band kernel vs global-z kernel,
    // profiles/r05_q_*): 32x32 171 vs 162, 44x22 211 vs 225, 24x42 196 vs 223, 16x64 180 vs 227, 7x9 40 vs 30: the global-z
    // kernel where a row is exactly four 8-column segments or the image is tiny, the band kernel otherwise
    const bool gs_shape = (w > 24 && w <= 32) || w < 16;
    int nparts = (bwd2_force == 2 || (bwd2_force == 0 && !gs_shape)) ? 0 : v64w3::launch_bwd2_gs(B, h, w, Hc, (const bf16_t*)s, (const bf16_t*)z, du,
                                                              (const bf16_t*)wdw, (bf16_t*)dz, ws, (hipStream_t)stream);
    const bool gs = nparts > 0;
    const int R2 = !gs && w <= 64 && !(C2 & 7) && du_bytes <= 0x7fffffffull ? pick_band_rows_bwd2(h, w, &lds2, &threads2) : 0;
    if (gs) {
    } else if (R2) {
        const int nbands = (h + R2 - 1) / R2, nchunk = (C2 + TCH - 1) / TCH;
        int bpb = nbands;
        while (bpb > 1 && (int64_t)((nbands + bpb - 1) / bpb) * nchunk * B < 4 * 512) --bpb;
        const int ngrp = (nbands + bpb - 1) / bpb;
        nparts = B * ngrp;                                     // <= B * nrg_of(h) * nsb_of(w): fits the same workspace
        hipLaunchKernelGGL(dwglu_bwd2_tile_kernel, dim3(grid8((int64_t)ngrp * nchunk * B)), dim3(threads2), lds2,
                           (hipStream_t)stream, h, w, Hc, B, R2, (65536 + R2 - 1) / R2, nbands, bpb, nchunk,
                           (const bf16_t*)s, (const bf16_t*)z,
                           (const bf16_t*)du, (uint64_t)B * h * w * C2 * 2, (const bf16_t*)wdw, (bf16_t*)dz, ws);
    } else {
        nparts = B * gy2;
        hipLaunchKernelGGL(dwconv_bwd2_kernel, dim3(grid8((int64_t)nx2 * gy2 * B)), dim3(256), 0, (hipStream_t)stream, h,
                           w, Hc, nx2, nrg_of(h), B, (const bf16_t*)s, (const bf16_t*)z, (const bf16_t*)wdw,
                           (const bf16_t*)du, (bf16_t*)dz, ws);
    }
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(dwconv_reduce_kernel, dim3((C2 * PK + 63) / 64), dim3(256), 0, (hipStream_t)stream, nparts, C2,
                       (const float*)ws, (bf16_t*)dwdw, (bf16_t*)dbdw, (R2 || gs) ? (bf16_t*)dz_colsum : (bf16_t*)nullptr, accumulate);
    YAT_CHECK_LAUNCH();
    if (dz_colsum && !R2 && !gs)                                     // direct path: the column sum of dz is its own pass
        return yat_colsum_bf16(B * h * w, C2, dz, C2, dz_colsum, accumulate, ws, stream);
    return YAT_OK;
}

}  // extern "C"
