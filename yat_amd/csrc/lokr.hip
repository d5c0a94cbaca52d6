// LoKr adapter support kernels for gfx950 (BASELINE config 5: ``lora_algo: lokr`` -> peft LoKrConfig(r, alpha,
// module_dropout, target_modules) wrapped around the model at /root/reference/common/trainer.py:212-238).
//
// [RECALL peft/tuners/lokr/layer.py]  For a target Linear / 1x1 Conv with weight [out, in]:
//   (out_l, out_k) = factorization(out), (in_m, in_n) = factorization(in);  w1 [out_l, in_m] (zeros at init),
//   w2 = w2_a [out_k, r] @ w2_b [r, in_n] (both kaiming-uniform);  delta_w = kron(w1, w2) * (alpha / r);
//   forward: base_layer(x) + F.linear(x, delta_w)   -- all in the module dtype (bf16), every op rounding.
// The dense products (x delta_w^T, dy delta_w, d_delta = dy^T x) run on the GEMM kernels of this library; what is left
// is HBM-bound glue, one launch per adapter:
//   yat_lokr_delta   : w2 = bf16(w2_a w2_b);  delta[(i,k),(j,n)] = bf16( bf16(w1[i,j] * w2[k,n]) * scale )
//   yat_lokr_project : autograd of the above from d_delta: d_w1, d_w2 (fp32 sums, fixed order), d_w2_a, d_w2_b.
#include "common.hpp"
#include "../../include/yat_hip.h"

namespace {

struct LokrP {
    int out_l, out_k, in_m, in_n, r, ld;
    float scale;
    const bf16_t* w1; const bf16_t* w2a; const bf16_t* w2b;
};

__device__ __forceinline__ float w2_elem(const LokrP& p, int k, int n) {
    float s = 0.f;
    for (int q = 0; q < p.r; ++q) s += bf2f(p.w2a[k * p.r + q]) * bf2f(p.w2b[q * p.in_n + n]);
    return rbf(s);                                             // the bf16 matmul w2_a @ w2_b
}

// grid (ceil(in / 256), out): one thread per element, consecutive threads along the row
__global__ __launch_bounds__(256) void lokr_delta_kernel(LokrP p, bf16_t* delta) {
    const int col = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    const int in = p.in_m * p.in_n;
    if (col >= in) return;
    const int i = row / p.out_k, k = row - i * p.out_k, j = col / p.in_n, n = col - j * p.in_n;
    float v = rbf(bf2f(p.w1[i * p.in_m + j]) * w2_elem(p, k, n));          // torch.kron in bf16
    if (p.scale != 1.0f) v = rbf(v * p.scale);                             // make_kron: rebuild * scale (skipped at 1)
    delta[(int64_t)row * p.ld + col] = f2bf(v);
}

// d_w1[i,j] = sum_{k,n} dR[(i,k),(j,n)] * w2[k,n],  dR = bf16(d_delta * scale).  One workgroup per (i,j).
__global__ __launch_bounds__(256) void lokr_dw1_kernel(LokrP p, const bf16_t* dd, bf16_t* dw1) {
    __shared__ float red[4];
    const int j = blockIdx.x, i = blockIdx.y;
    float s = 0.f;
    const int per = p.out_k * p.in_n;
    for (int e = threadIdx.x; e < per; e += 256) {
        const int k = e / p.in_n, n = e - k * p.in_n;
        float g = bf2f(dd[((int64_t)i * p.out_k + k) * p.ld + j * p.in_n + n]);
        if (p.scale != 1.0f) g = rbf(g * p.scale);
        s += g * w2_elem(p, k, n);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) dw1[i * p.in_m + j] = f2bf(red[0] + red[1] + red[2] + red[3]);
}

// partial[i][k*in_n + n] = sum_j dR[(i,k),(j,n)] * w1[i,j]   (one workgroup per i; summed over i by the next kernel)
__global__ __launch_bounds__(256) void lokr_dw2_partial_kernel(LokrP p, const bf16_t* dd, float* partial) {
    const int i = blockIdx.x;
    const int per = p.out_k * p.in_n;
    for (int e = threadIdx.x; e < per; e += 256) {
        const int k = e / p.in_n, n = e - k * p.in_n;
        const bf16_t* row = dd + ((int64_t)i * p.out_k + k) * p.ld + n;
        float s = 0.f;
        for (int j = 0; j < p.in_m; ++j) {
            float g = bf2f(row[j * p.in_n]);
            if (p.scale != 1.0f) g = rbf(g * p.scale);
            s += g * bf2f(p.w1[i * p.in_m + j]);
        }
        partial[(int64_t)i * per + e] = s;
    }
}

// d_w2 = bf16(sum_i partial[i]);  then d_w2_a = bf16(d_w2 w2_b^T), d_w2_b = bf16(w2_a^T d_w2) -- one workgroup, the
// d_w2 tile staged in LDS (out_k * in_n <= 16 K elements for every SANA target).
__global__ __launch_bounds__(256) void lokr_dw2_final_kernel(LokrP p, const float* partial, bf16_t* dw2a, bf16_t* dw2b) {
    extern __shared__ float dw2[];
    const int per = p.out_k * p.in_n;
    for (int e = threadIdx.x; e < per; e += 256) {
        float s = 0.f;
        for (int i = 0; i < p.out_l; ++i) s += partial[(int64_t)i * per + e];
        dw2[e] = rbf(s);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < p.out_k * p.r; e += 256) {          // d_w2_a[k, q] = sum_n d_w2[k, n] * w2_b[q, n]
        const int k = e / p.r, q = e - k * p.r;
        float s = 0.f;
        for (int n = 0; n < p.in_n; ++n) s += dw2[k * p.in_n + n] * bf2f(p.w2b[q * p.in_n + n]);
        dw2a[e] = f2bf(s);
    }
    for (int e = threadIdx.x; e < p.r * p.in_n; e += 256) {           // d_w2_b[q, n] = sum_k w2_a[k, q] * d_w2[k, n]
        const int q = e / p.in_n, n = e - q * p.in_n;
        float s = 0.f;
        for (int k = 0; k < p.out_k; ++k) s += bf2f(p.w2a[k * p.r + q]) * dw2[k * p.in_n + n];
        dw2b[e] = f2bf(s);
    }
}

// Small-output weight gradient of the factored adapter path: out[q, n] (+)= sum_row A[row, q] * X[row, n] with a few
// (R <= 16) x (N <= 128) outputs and a million-row reduction (rows = B*N_tokens*in_m) -- far outside what a tiled GEMM is
// for.  HBM-bound streaming pass; every workgroup leaves one fp32 partial, a second launch sums the partials in a fixed
// order (no atomics).  (v1 -- a wave per row, a lane per column pair -- was latency-bound: 226-477 us per call.)
// v3: the reduction runs on the matrix cores.  out^T tile = a^T x is a 16 x 128 MFMA accumulator block (q padded to 16, eight
// 16-column tiles), k = 32 rows per v_mfma_f32_16x16x32_bf16; both operands are "k-strided" (k = the row index is the slow
// dimension of a and x), so a workgroup stages CH rows of each as row-major LDS images with 16-byte loads and the waves
// fetch fragments with the hardware-transposing ds_read_b64_tr_b16 (the x image XOR-swizzled like gemm256's k-strided
// operand, the 32-byte rows of the a image need no swizzle).  The pass is a stream through LDS: 16 MFMAs per wave per 256 rows.
// (v1, a wave per row: 226-477 us per call; v2, scalar FMAs from LDS: ~200 us in the step; traffic alone is ~35 us.)
__device__ __forceinline__ uint32_t sw_swz(uint32_t krow) { return ((krow & 3) | (((krow >> 3) & 1) << 2)) << 1; }

__global__ __launch_bounds__(256) void lokr_small_wgrad_kernel(int64_t rows, int R, int N, int ldx, int npad, const bf16_t* A,
                                                               const bf16_t* X, float* partial) {
    constexpr int CH = 256;                                  // rows per staged chunk = 8 k-steps of 32
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ximg = smem;                                       // [CH][128] bf16, 256-byte rows, chunk ^= sw_swz(row)
    char* aimg = smem + CH * 256;                            // [CH][16] bf16, 32-byte rows (columns >= R stay zero)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.y * 128;
    const int Nb = N - n0 < 128 ? N - n0 : 128;
    const int VPR = Nb >> 3, ntile = (Nb + 15) >> 4;
    for (int v = threadIdx.x; v < CH * 2; v += 256) *reinterpret_cast<u32x4*>(aimg + v * 16) = u32x4{0u, 0u, 0u, 0u};
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int64_t nchunk = (rows + CH - 1) / CH;
    for (int64_t ck = blockIdx.x; ck < nchunk; ck += gridDim.x) {
        const int64_t r0 = ck * CH;
        const int nr = (int)(rows - r0 < CH ? rows - r0 : CH);
        // Stage the chunk with every global load in flight before the first LDS store: as a plain loop (runtime trip count,
        // load -> store per iteration) the compiler serialised up to 16 dependent round trips to memory per chunk and the
        // kernel ran at 0.9 TB/s of its 168 MB (round 4; the rank-8, in_n = 56 adapter gradients of BASELINE config 5).
        u32x4 xv[16], av[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int v = threadIdx.x + 256 * i;
            const int r = v / VPR, c = v - r * VPR;
            xv[i] = u32x4{0u, 0u, 0u, 0u};                   // rows past the end contribute zeros (never stale LDS bits)
            if (v < CH * VPR && r < nr) xv[i] = *reinterpret_cast<const u32x4*>(X + (r0 + r) * ldx + n0 + c * 8);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = threadIdx.x + 256 * i;
            const int r = v / (R >> 3), c = v - r * (R >> 3);
            av[i] = u32x4{0u, 0u, 0u, 0u};
            if (v < CH * (R >> 3) && r < nr) av[i] = *reinterpret_cast<const u32x4*>(A + (r0 + r) * R + c * 8);
        }
        __syncthreads();                                     // previous chunk fully consumed (and the zero fill done)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int v = threadIdx.x + 256 * i;
            const int r = v / VPR, c = v - r * VPR;
            if (v < CH * VPR) *reinterpret_cast<u32x4*>(ximg + r * 256 + ((c ^ sw_swz(r)) << 4)) = xv[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = threadIdx.x + 256 * i;
            const int r = v / (R >> 3), c = v - r * (R >> 3);
            if (v < CH * (R >> 3)) *reinterpret_cast<u32x4*>(aimg + r * 32 + c * 16) = av[i];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint32_t kk = wave + 4 * ks;               // this wave's 32-row k-steps of the chunk
            const uint32_t ra = kk * 32 + 8 * g + q, rb = ra + 4;
            const bf16x8 af = cat4(lds_read_tr4(aimg, ra * 32 + p * 8), lds_read_tr4(aimg, rb * 32 + p * 8));
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t < ntile) {
                    const uint32_t col = t * 16 + 4 * p;
                    const uint32_t ca = (col >> 3) ^ sw_swz(ra), cb = (col >> 3) ^ sw_swz(rb);
                    const bf16x8 xf = cat4(lds_read_tr4(ximg, ra * 256 + ca * 16 + (p & 1) * 8),
                                           lds_read_tr4(ximg, rb * 256 + cb * 16 + (p & 1) * 8));
                    acc[t] = mfma16(af, xf, acc[t]);         // acc[t][r] @ lane (g, li): out[q = 4g + r][n = 16t + li]
                }
            }
        }
    }
    __syncthreads();                                         // every wave is done reading the images
    float* red = reinterpret_cast<float*>(smem);             // [4 waves][16 q][128 n] fp32 = 32 KiB
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * g + r) * 128 + t * 16 + (lane & 15)] = acc[t][r];
    __syncthreads();
    for (int e = threadIdx.x; e < R * Nb; e += 256) {
        const int qq = e / Nb, n = e - qq * Nb;
        partial[((int64_t)blockIdx.x * R + qq) * npad + n0 + n] =
            red[(0 * 16 + qq) * 128 + n] + red[(1 * 16 + qq) * 128 + n] + red[(2 * 16 + qq) * 128 + n] + red[(3 * 16 + qq) * 128 + n];
    }
}
// out[q, n] = (accumulate ? out : 0) + bf16(scale * sum_g partial[g][q][n]), q < r_out.  One workgroup per 16 outputs: 16
// slices of the partial list (4 waves x 4 lane groups) x 16 consecutive outputs, fixed order, two shuffles + one LDS step.
__global__ __launch_bounds__(256) void lokr_small_wgrad_final_kernel(int G, int R, int N, int npad, int r_out, float scale,
                                                                     const float* partial, bf16_t* out, int ldo,
                                                                     int accumulate) {
    __shared__ float red[4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 16 + (lane & 15), slice = wave * 4 + (lane >> 4);
    const int q = e / N, n = e - q * N;
    float s = 0.f;
    if (q < r_out)
        for (int g = slice; g < G; g += 16) s += partial[((int64_t)g * R + q) * npad + n];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < 16) red[wave][lane] = s;
    __syncthreads();
    if (threadIdx.x >= 16 || q >= r_out) return;
    s = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) * scale;
    bf16_t* o = out + (int64_t)q * ldo + n;
    if (accumulate) s = rbf(s) + bf2f(*o);
    *o = f2bf(s);
}

// The two row-streaming products of the factored path, rows = M*in_m (a million at B = 32), N = in_n <= 128, R <= 16:
//   FWD: t1[row, q]  = bf16( sum_n x[row, n] * wb[q, n] )                       (T1 = x' w2_b^T)
//   BWD: dx[row, n]  = bf16( bf16( sum_q h[row, q] * wb[q, n] ) + dx[row, n] )  (dx' += H' w2_b, the GEMM's residual rounding)
// A tiled GEMM spends a 128-wide tile on the 8 useful columns (measured 124 / 202 us per call); both are HBM-bound passes
// (168 / 315 MB at D = 2240, B = 32).
//
// FWD on the matrix cores (round 5): a wave takes 16 rows per step; x' is k-contiguous, so a lane's operand fragment of
// v_mfma_f32_16x16x32_bf16 -- 8 consecutive columns of one row -- is one 16-byte global load, w2_b's fragments (q = lane & 15,
// zero rows past R) sit in registers for the whole kernel, and with w2_b as the first operand a lane ends up with 4
// consecutive q of ONE row: an 8-byte store, 16 bytes per row from two lanes.  No LDS, no cross-lane step; four row tiles of
// loads are in flight per wave.  (v2 -- 8 lanes per row, 64 scalar FMAs + 24 shuffles per 16 bytes loaded, w2_b broadcast from
// LDS -- was bound by vector issue: 48.8 us for the 168 MB of a D = 2240 target at B = 32 against 27 us of traffic.)
// With im > 0 the result is written as T1_flat [rows / im, im * R] with row stride ldt -- row (m, j) at io[m * ldt + j * R] -- so
// that it can be handed to the base Linear's GEMM as its second operand (yat_gemm_epilogue.a2: K2 further columns with A's stride).
template <int KS>                                              // k-steps of 32 columns: N <= 32 KS
__global__ __launch_bounds__(256) void lokr_rows_fwd_kernel(int64_t rows, int N, int R, const bf16_t* wb, const bf16_t* x,
                                                            bf16_t* io, int im, int ldt) {
    constexpr int U = 4;                                       // row tiles per wave step
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    bf16x8 zero;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero[e] = (__bf16)0.0f;
    bf16x8 wf[KS];
    bool live[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c = 32 * ks + 8 * g;
        live[ks] = c < N;
        wf[ks] = (li < R && live[ks]) ? *reinterpret_cast<const bf16x8*>(wb + li * N + c) : zero;
    }
    const int64_t tiles = (rows + 15) >> 4;
    for (int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * U; t0 < tiles; t0 += (int64_t)gridDim.x * 4 * U) {
        bf16x8 xf[U][KS];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = (t0 + u) * 16 + li;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                xf[u][ks] = (row < rows && live[ks]) ? *reinterpret_cast<const bf16x8*>(x + row * N + 32 * ks + 8 * g) : zero;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = mfma16(wf[ks], xf[u][ks], acc);      // acc[r] = t1[row][q = 4 g + r]
            const int64_t row = (t0 + u) * 16 + li;
            if (row < rows && 4 * g < R) {
                bf16_t* dst = io + row * R;
                if (im > 0) { const int64_t m = row / im; dst = io + m * ldt + (row - m * im) * R; }
                *reinterpret_cast<u32x2*>(dst + 4 * g) = pack4(acc[0], acc[1], acc[2], acc[3]);
            }
        }
    }
}

// BWD: dx' is rows of N / 8 16-byte chunks back to back, so the pass is one linear stream of chunks -- thread i of a step owns chunk
// i (row = i / (N / 8), no idle lanes whatever N / 8 is), reads its row's h (16 or 32 bytes, the same for the row's threads), forms
// its 8 columns from w2_b in LDS (fp32, per-chunk columns) and updates the chunk in place, no exchange.  (v1 gave a lane a whole
// row: every 16-byte load of a wave touched 64 different rows, 245 us per call against ~65 us of traffic; v2 padded a row to 8 or
// 16 lanes: 6 of 16 idle at in_n = 80, 3.2 TB/s.)
template <int R>
__global__ __launch_bounds__(256) void lokr_rows_bwd_kernel(uint32_t chunks, int N, const bf16_t* wb, const bf16_t* a, bf16_t* io) {
    __shared__ float w[R][128];
    for (int e = threadIdx.x; e < R * 128; e += 256) w[e >> 7][e & 127] = (e & 127) < N ? bf2f(wb[(e >> 7) * N + (e & 127)]) : 0.f;
    __syncthreads();
    const uint32_t cpr = (uint32_t)N >> 3;                     // chunks per row
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < chunks; i += gridDim.x * 256u) {
        const uint32_t row = i / cpr, n = (i - row * cpr) * 8;
        float h[R], o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, d[8];
#pragma unroll
        for (int v = 0; v < R / 8; ++v) unpack8(*reinterpret_cast<const u32x4*>(a + (int64_t)row * R + v * 8), h + v * 8);
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(&w[q][n]), hi = *reinterpret_cast<const f32x4*>(&w[q][n + 4]);
            o[0] += h[q] * lo[0]; o[1] += h[q] * lo[1]; o[2] += h[q] * lo[2]; o[3] += h[q] * lo[3];
            o[4] += h[q] * hi[0]; o[5] += h[q] * hi[1]; o[6] += h[q] * hi[2]; o[7] += h[q] * hi[3];
        }
        bf16_t* dp = io + (int64_t)i * 8;
        unpack8(*reinterpret_cast<const u32x4*>(dp), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rbf(o[e]) + d[e];
        *reinterpret_cast<u32x4*>(dp) = pack8(o);
    }
}

// Plain LoRA through the base GEMM's second operand pair (yat_gemm_epilogue.b2): lora_B^T [R, out] of EVERY adapter, times
// `scale`, into the first R columns of its target's rows in a shadow of the model's flat weights (row stride = the target's in):
// dst[n, q] = bf16(scale * src[q, n]).  One launch for the whole adapter set: table[4 e ..] = {offset of B^T in `src`, offset of the
// target's first row in `dst`, out, in} (elements).  Reads run along n (coalesced), a thread writes its row's R values at once.
template <int R>
__global__ __launch_bounds__(256) void lora_scatter_b_kernel(const int64_t* table, float scale, const bf16_t* src, bf16_t* dst) {
    const int64_t* t = table + 4 * (int64_t)blockIdx.y;
    const int out = (int)t[2], in = (int)t[3];
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= out) return;
    const bf16_t* sp = src + t[0] + n;
    float v[R];
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = bf2f(sp[(int64_t)q * out]) * scale;
    bf16_t* dp = dst + t[1] + (int64_t)n * in;
#pragma unroll
    for (int c = 0; c < R / 8; ++c) *reinterpret_cast<u32x4*>(dp + c * 8) = pack8(v + c * 8);
}
// Rank-R expansion over a whole layer width (plain LoRA: u = T B^T with K = R, dx += dT A): a GEMM with an 8-deep reduction
// is all epilogue, so it runs as a stream instead -- io[row, n] = f(sum_q h[row, q] * w[q, n]) with w's column block
// (CB = 512 columns, fp32) in LDS, a lane per 16-byte chunk of the output row, a wave per 512 contiguous columns.
//   residual = 0: io = bf16(bf16(sum) * scale)      (peft: lora_B(..) * scaling, each op rounded)
//   residual = 1: io = bf16(bf16(sum) + io)         (accumulation into an input gradient, the GEMM's residual rounding)
template <int R>
__global__ __launch_bounds__(256) void lokr_wide_rows_kernel(int64_t rows, int N, int ldio, const bf16_t* wsrc, const bf16_t* h,
                                                             bf16_t* io, float scale, int residual) {
    constexpr int CB = 512;
    __shared__ float w[R][CB];
    const int n0 = blockIdx.y * CB;
    for (int e = threadIdx.x; e < R * CB; e += 256) {
        const int q = e / CB, n = n0 + (e - q * CB);
        w[q][e - q * CB] = n < N ? bf2f(wsrc[(int64_t)q * N + n]) : 0.f;
    }
    __syncthreads();
    const int c = threadIdx.x & 63, n = n0 + c * 8;          // this lane's 8 columns
    if (n >= N) return;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        float hv[R], o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < R / 8; ++v) unpack8(*reinterpret_cast<const u32x4*>(h + row * R + v * 8), hv + v * 8);
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(&w[q][c * 8]), hi = *reinterpret_cast<const f32x4*>(&w[q][c * 8 + 4]);
            o[0] += hv[q] * lo[0]; o[1] += hv[q] * lo[1]; o[2] += hv[q] * lo[2]; o[3] += hv[q] * lo[3];
            o[4] += hv[q] * hi[0]; o[5] += hv[q] * hi[1]; o[6] += hv[q] * hi[2]; o[7] += hv[q] * hi[3];
        }
        bf16_t* dp = io + row * ldio + n;
        if (residual) {
            float d[8];
            unpack8(*reinterpret_cast<const u32x4*>(dp), d);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rbf(o[e]) + d[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rbf(o[e]) * scale;
        }
        *reinterpret_cast<u32x4*>(dp) = pack8(o);
    }
}

int fill(LokrP& p, int out_l, int out_k, int in_m, int in_n, int r, const void* w1, const void* w2a, const void* w2b,
         float scale, int ld) {
    if (out_l <= 0 || out_k <= 0 || in_m <= 0 || in_n <= 0 || r <= 0 || r > 64 || !w1 || !w2a || !w2b ||
        ld < in_m * in_n || out_l * out_k > 65535)
        return YAT_EINVAL;
    p.out_l = out_l; p.out_k = out_k; p.in_m = in_m; p.in_n = in_n; p.r = r; p.ld = ld; p.scale = scale;
    p.w1 = (const bf16_t*)w1; p.w2a = (const bf16_t*)w2a; p.w2b = (const bf16_t*)w2b;
    return YAT_OK;
}


// MODE 0: o1 = bf16(bf16(x * y) * scale).   MODE 1: g = bf16(x * scale); o1 = bf16(g * z); o2 = bf16(g * y)
//         (x = d_delta, y = A1, z = A2 -> o1 = t1, o2 = t2).
template <int MODE>
__global__ __launch_bounds__(256) void hadamard_kernel(int rows, int cols, const bf16_t* x, int ldx, const bf16_t* y, int ldy,
                                                       const bf16_t* z, int ldz, float scale, bf16_t* o1, int ld1, bf16_t* o2,
                                                       int ld2) {
    const int cpr = cols >> 3;
    const int64_t total = (int64_t)rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr;
        const int c = (int)(i - r * cpr) * 8;
        float xv[8], yv[8], zv[8], a[8], b[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + r * ldx + c), xv);
        unpack8(*reinterpret_cast<const u32x4*>(y + r * ldy + c), yv);
        if (MODE == 1) unpack8(*reinterpret_cast<const u32x4*>(z + r * ldz + c), zv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (MODE == 0) a[e] = rbf(rbf(xv[e] * yv[e]) * scale);
            else {
                const float g = rbf(xv[e] * scale);
                a[e] = rbf(g * zv[e]);
                b[e] = rbf(g * yv[e]);
            }
        }
        *reinterpret_cast<u32x4*>(o1 + r * ld1 + c) = pack8(a);
        if (MODE == 1) *reinterpret_cast<u32x4*>(o2 + r * ld2 + c) = pack8(b);
    }
}

// DoRA (peft LoraConfig(use_dora=True), common/trainer.py:215-220) [RECALL peft/tuners/lora/dora.py DoraLinearLayer]: with
// lw = lora_B lora_A (bf16), u = bf16(W + bf16(scaling * lw)), n_j = bf16(||u_j||_2) (detached), s_j = bf16(m_j / n_j):
//   result = base(x) + (s - 1) (x W^T) + s scaling lora(x)  =  base(x) + x delta^T  with  delta_j = s_j (W_j + scaling lw_j) - W_j.
// One workgroup per output row: pass 1 the norm, pass 2 the delta row (the row is re-read from L2).
// MODE 0: delta + (s, n) out.  MODE 1 (backward): dm_j = (sum_l dd[j,l] u[j,l]) / n_j and t1 = bf16(s_j * scaling * dd).
template <int MODE>
__global__ __launch_bounds__(256) void dora_kernel(int cols, const bf16_t* W, int ldw, const bf16_t* lw, int ldl,
                                                   const bf16_t* mag, float scaling, bf16_t* out, int ldo, float* s_buf,
                                                   float* n_buf, const bf16_t* dd, int ldd, bf16_t* dmag) {
    __shared__ float red[4];
    const int j = blockIdx.x, cpr = cols >> 3;
    const bf16_t* wr = W + (int64_t)j * ldw;
    const bf16_t* lr = lw + (int64_t)j * ldl;
    float acc = 0.f;
    for (int c = threadIdx.x; c < cpr; c += 256) {
        float wv[8], lv[8], dv[8];
        unpack8(*reinterpret_cast<const u32x4*>(wr + c * 8), wv);
        unpack8(*reinterpret_cast<const u32x4*>(lr + c * 8), lv);
        if (MODE == 1) unpack8(*reinterpret_cast<const u32x4*>(dd + (int64_t)j * ldd + c * 8), dv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float u = rbf(wv[e] + rbf(scaling * lv[e]));
            acc += MODE == 0 ? u * u : dv[e] * u;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    float s, n;
    if (MODE == 0) {
        n = rbf(sqrtf(tot));
        s = rbf(bf2f(mag[j]) / n);
        if (threadIdx.x == 0) { s_buf[j] = s; n_buf[j] = n; }
    } else {
        s = s_buf[j];
        n = n_buf[j];
        if (threadIdx.x == 0) dmag[j] = f2bf(tot / n);
    }
    for (int c = threadIdx.x; c < cpr; c += 256) {
        float a[8];
        if (MODE == 0) {
            float wv[8], lv[8];
            unpack8(*reinterpret_cast<const u32x4*>(wr + c * 8), wv);
            unpack8(*reinterpret_cast<const u32x4*>(lr + c * 8), lv);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = s * (wv[e] + scaling * lv[e]) - wv[e];
        } else {
            float dv[8];
            unpack8(*reinterpret_cast<const u32x4*>(dd + (int64_t)j * ldd + c * 8), dv);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = rbf(rbf(s * scaling) * dv[e]);
        }
        *reinterpret_cast<u32x4*>(out + (int64_t)j * ldo + c * 8) = pack8(a);
    }
}

}  // namespace

extern "C" {

int yat_lokr_delta(int out_l, int out_k, int in_m, int in_n, int r, const void* w1, const void* w2_a, const void* w2_b,
                   float scale, void* delta, int ld, yat_stream_t stream) {
    LokrP p;
    if (fill(p, out_l, out_k, in_m, in_n, r, w1, w2_a, w2_b, scale, ld) || !delta) return YAT_EINVAL;
    hipLaunchKernelGGL(lokr_delta_kernel, dim3((in_m * in_n + 255) / 256, out_l * out_k), dim3(256), 0, (hipStream_t)stream, p,
                       (bf16_t*)delta);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

uint64_t yat_lokr_project_workspace_bytes(int out_l, int out_k, int in_n) {
    return (uint64_t)out_l * out_k * in_n * sizeof(float);
}

int yat_lokr_project(int out_l, int out_k, int in_m, int in_n, int r, const void* w1, const void* w2_a, const void* w2_b,
                     float scale, const void* d_delta, int ld, void* d_w1, void* d_w2_a, void* d_w2_b, void* workspace,
                     yat_stream_t stream) {
    LokrP p;
    if (fill(p, out_l, out_k, in_m, in_n, r, w1, w2_a, w2_b, scale, ld) || !d_delta || !d_w1 || !d_w2_a || !d_w2_b ||
        !workspace || (uint64_t)out_k * in_n * sizeof(float) > 65536)
        return YAT_EINVAL;
    hipLaunchKernelGGL(lokr_dw1_kernel, dim3(in_m, out_l), dim3(256), 0, (hipStream_t)stream, p, (const bf16_t*)d_delta,
                       (bf16_t*)d_w1);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(lokr_dw2_partial_kernel, dim3(out_l), dim3(256), 0, (hipStream_t)stream, p, (const bf16_t*)d_delta,
                       (float*)workspace);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(lokr_dw2_final_kernel, dim3(1), dim3(256), out_k * in_n * sizeof(float), (hipStream_t)stream, p,
                       (const float*)workspace, (bf16_t*)d_w2_a, (bf16_t*)d_w2_b);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

static int lokr_rows_launch(int64_t rows, int N, int R, int backward, const void* w2_b, const void* a, void* io, int im, int ldt,
                            yat_stream_t stream);
int yat_lokr_rows(int64_t rows, int N, int R, int backward, const void* w2_b, const void* a, void* io, yat_stream_t stream) {
    return lokr_rows_launch(rows, N, R, backward, w2_b, a, io, 0, 0, stream);
}
int yat_lokr_rows_fwd_flat(int64_t rows, int N, int R, int in_m, const void* w2_b, const void* x, void* t1_flat, int ldt,
                           yat_stream_t stream) {
    if (in_m <= 0 || rows % in_m || (ldt & 7) || ldt < in_m * R) return YAT_EINVAL;
    return lokr_rows_launch(rows, N, R, 0, w2_b, x, t1_flat, in_m, ldt, stream);
}
static int lokr_rows_launch(int64_t rows, int N, int R, int backward, const void* w2_b, const void* a, void* io, int im, int ldt,
                            yat_stream_t stream) {
    if (rows <= 0 || (R != 8 && R != 16) || N <= 0 || N > 128 || (N & 7) || !w2_b || !a || !io) return YAT_EINVAL;
    const bf16_t* wb = (const bf16_t*)w2_b;
    const bf16_t* ap = (const bf16_t*)a;
    bf16_t* iop = (bf16_t*)io;
    hipStream_t st = (hipStream_t)stream;
    const dim3 block(256);
    if (!backward) {
        const int64_t g64 = ((rows + 15) / 16 + 15) / 16;                   // 4 waves x 4 row tiles per workgroup step
        const dim3 grid((unsigned)(g64 > 8192 ? 8192 : g64));
#define YAT_FWD(KS) hipLaunchKernelGGL((lokr_rows_fwd_kernel<KS>), grid, block, 0, st, rows, N, R, wb, ap, iop, im, ldt)
        if (N <= 32) YAT_FWD(1); else if (N <= 64) YAT_FWD(2); else if (N <= 96) YAT_FWD(3); else YAT_FWD(4);
#undef YAT_FWD
        YAT_CHECK_LAUNCH();
        return YAT_OK;
    }
    const int64_t chunks = rows * (N >> 3);
    if (chunks > 0xffffffffll) return YAT_EINVAL;
    const int64_t g64 = (chunks + 255) / 256;
    const dim3 grid((unsigned)(g64 > 8192 ? 8192 : g64));
    if (R == 8) hipLaunchKernelGGL((lokr_rows_bwd_kernel<8>), grid, block, 0, st, (uint32_t)chunks, N, wb, ap, iop);
    else hipLaunchKernelGGL((lokr_rows_bwd_kernel<16>), grid, block, 0, st, (uint32_t)chunks, N, wb, ap, iop);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

int yat_lora_scatter_b(int entries, int max_out, int R, float scale, const void* table, const void* src, void* dst,
                       yat_stream_t stream) {
    if (entries <= 0 || entries > 65535 || max_out <= 0 || (R != 8 && R != 16) || !table || !src || !dst) return YAT_EINVAL;
    const dim3 grid((max_out + 255) / 256, entries), block(256);
    if (R == 8) hipLaunchKernelGGL((lora_scatter_b_kernel<8>), grid, block, 0, (hipStream_t)stream, (const int64_t*)table, scale,
                                   (const bf16_t*)src, (bf16_t*)dst);
    else hipLaunchKernelGGL((lora_scatter_b_kernel<16>), grid, block, 0, (hipStream_t)stream, (const int64_t*)table, scale,
                            (const bf16_t*)src, (bf16_t*)dst);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

int yat_rank_expand(int64_t rows, int N, int R, const void* w, const void* h, void* io, int ldio, float scale, int residual,
                    yat_stream_t stream) {
    if (rows <= 0 || (R != 8 && R != 16) || N <= 0 || (N & 7) || (ldio & 7) || ldio < N || !w || !h || !io) return YAT_EINVAL;
    const int nblk = (N + 511) / 512;
    int64_t g64 = (rows + 3) / 4;
    const int gx = (int)(g64 > 2048 / nblk + 1 ? 2048 / nblk + 1 : g64);
    if (R == 8)
        hipLaunchKernelGGL((lokr_wide_rows_kernel<8>), dim3(gx, nblk), dim3(256), 0, (hipStream_t)stream, rows, N, ldio,
                           (const bf16_t*)w, (const bf16_t*)h, (bf16_t*)io, scale, residual);
    else
        hipLaunchKernelGGL((lokr_wide_rows_kernel<16>), dim3(gx, nblk), dim3(256), 0, (hipStream_t)stream, rows, N, ldio,
                           (const bf16_t*)w, (const bf16_t*)h, (bf16_t*)io, scale, residual);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

static int small_wgrad_groups(int64_t rows, int nblk) {
    const int64_t nchunk = (rows + 255) / 256;
    int64_t g = 512 / nblk;                                  // ~two workgroups per CU over all column blocks
    if (g < 32) g = 32;
    return (int)(nchunk < g ? nchunk : g);
}

uint64_t yat_lokr_small_wgrad_workspace_bytes(int64_t rows, int R, int N) {
    const int nblk = (N + 127) / 128;
    return (uint64_t)small_wgrad_groups(rows, nblk) * R * nblk * 128 * sizeof(float);
}

int yat_lokr_small_wgrad(int64_t rows, int R, int N, int r_out, const void* a, const void* x, int ldx, void* out, int ldo,
                         float scale, int accumulate, void* workspace, yat_stream_t stream) {
    if (rows <= 0 || (R != 8 && R != 16) || N <= 0 || (N & 7) || (ldx & 7) || ldx < N || ldo < N || r_out <= 0 || r_out > R ||
        !a || !x || !out || !workspace)
        return YAT_EINVAL;
    const int nblk = (N + 127) / 128, npad = nblk * 128;
    const int G = small_wgrad_groups(rows, nblk);
    const int lds = 256 * (128 + 16) * 2;                    // x image 64 KiB + a image 8 KiB
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)lokr_small_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 73728) !=
            hipSuccess)
            return YAT_EINVAL;
        attr_set = true;
    }
    hipLaunchKernelGGL(lokr_small_wgrad_kernel, dim3(G, nblk), dim3(256), lds, (hipStream_t)stream, rows, R, N, ldx, npad,
                       (const bf16_t*)a, (const bf16_t*)x, (float*)workspace);
    YAT_CHECK_LAUNCH();
    const int n_out = r_out * N;
    hipLaunchKernelGGL(lokr_small_wgrad_final_kernel, dim3((n_out + 15) / 16), dim3(256), 0, (hipStream_t)stream, G, R, N, npad,
                       r_out, scale, (const float*)workspace, (bf16_t*)out, ldo, accumulate);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}


// peft LoHa (HadaWeight [RECALL peft/tuners/loha/layer.py]): delta_w = ((w1a w1b) * (w2a w2b)) * scale, bf16 op by op, and
// the element-wise half of its hand-written backward: g = dd * scale; t1 = g * A2; t2 = g * A1.  rows x cols views with row strides.
int yat_hadamard_scale(int rows, int cols, const void* a, int lda, const void* b, int ldb, float scale, void* out, int ldo,
                       yat_stream_t stream) {
    if (rows <= 0 || cols <= 0 || (cols & 7) || (lda & 7) || (ldb & 7) || (ldo & 7) || lda < cols || ldb < cols || ldo < cols ||
        !a || !b || !out)
        return YAT_EINVAL;
    const int64_t chunks = (int64_t)rows * (cols >> 3);
    int64_t nb = (chunks + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(hadamard_kernel<0>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, rows, cols, (const bf16_t*)a,
                       lda, (const bf16_t*)b, ldb, (const bf16_t*)nullptr, 0, scale, (bf16_t*)out, ldo, (bf16_t*)nullptr, 0);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

int yat_hadamard_bwd(int rows, int cols, const void* dd, int ldd, const void* a1, int ld1, const void* a2, int ld2, float scale,
                     void* t1, int ldt1, void* t2, int ldt2, yat_stream_t stream) {
    if (rows <= 0 || cols <= 0 || (cols & 7) || ((ldd | ld1 | ld2 | ldt1 | ldt2) & 7) || ldd < cols || ld1 < cols ||
        ld2 < cols || ldt1 < cols || ldt2 < cols || !dd || !a1 || !a2 || !t1 || !t2)
        return YAT_EINVAL;
    const int64_t chunks = (int64_t)rows * (cols >> 3);
    int64_t nb = (chunks + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(hadamard_kernel<1>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, rows, cols, (const bf16_t*)dd,
                       ldd, (const bf16_t*)a1, ld1, (const bf16_t*)a2, ld2, scale, (bf16_t*)t1, ldt1, (bf16_t*)t2, ldt2);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

// DoRA: see dora_kernel above.  W, lw (= lora_B lora_A), delta / t1 are rows x cols bf16 views (cols % 8 == 0) with row strides
// in elements; mag / dmag bf16 [rows]; s_buf / n_buf fp32 [rows] written by yat_dora_delta and read by yat_dora_bwd.
int yat_dora_delta(int rows, int cols, const void* W, int ldw, const void* lw, int ldl, const void* mag, float scaling,
                   void* delta, int ldd, float* s_buf, float* n_buf, yat_stream_t stream) {
    if (rows <= 0 || cols <= 0 || (cols & 7) || ((ldw | ldl | ldd) & 7) || ldw < cols || ldl < cols || ldd < cols || !W || !lw ||
        !mag || !delta || !s_buf || !n_buf)
        return YAT_EINVAL;
    hipLaunchKernelGGL(dora_kernel<0>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, cols, (const bf16_t*)W, ldw,
                       (const bf16_t*)lw, ldl, (const bf16_t*)mag, scaling, (bf16_t*)delta, ldd, s_buf, n_buf,
                       (const bf16_t*)nullptr, 0, (bf16_t*)nullptr);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

int yat_dora_bwd(int rows, int cols, const void* dd, int ldd, const void* W, int ldw, const void* lw, int ldl, float scaling,
                 const float* s_buf, const float* n_buf, void* t1, int ldt, void* dmag, yat_stream_t stream) {
    if (rows <= 0 || cols <= 0 || (cols & 7) || ((ldw | ldl | ldd | ldt) & 7) || ldw < cols || ldl < cols || ldd < cols ||
        ldt < cols || !W || !lw || !dd || !t1 || !dmag || !s_buf || !n_buf)
        return YAT_EINVAL;
    hipLaunchKernelGGL(dora_kernel<1>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, cols, (const bf16_t*)W, ldw,
                       (const bf16_t*)lw, ldl, (const bf16_t*)nullptr, scaling, (bf16_t*)t1, ldt, (float*)s_buf, (float*)n_buf,
                       (const bf16_t*)dd, ldd, (bf16_t*)dmag);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

}  // extern "C"
