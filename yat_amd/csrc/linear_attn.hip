// ReLU linear attention (SANA self-attention, head dim 32) forward + backward on MFMA for gfx950.
//
// Restates diffusers SanaLinearAttnProcessor2_0 (imported at
// /root/reference/utils/patch_sana_attention_layers.py:7; stock attn1 processor of the model that
// train_sana.py:210 trains):  q,k <- ReLU;  fp32:  S = [V;1]^T K  (33x32 per head),
// U = Q S^T (N x 33),  out = U[:, :32] / (U[:, 32] + 1e-15).
//
// The reference up-casts q/k/v to fp32.  q, relu(k), v ARE bf16 values, so products with them are
// exact in the fp32 accumulator of a bf16 MFMA; the fp32 state S (and dS, dU) is fed as a bf16
// hi + lo pair (x = hi + lo + O(2^-17 x)), i.e. 2-3 MFMAs per product instead of an fp32 pipeline
// at 1/16 of the rate.  All kernels are HBM-bound after that (0.2 % of the block's FLOPs).
//
//  state   : S[b,h] = sum_n [v_n;1] relu(k_n)^T.  Tokens are the contraction index, so both operands
//            must be transposed: 64-token tiles of k and v are LDS-DMA'd (wave-private, 64-B rows)
//            and fetched with ds_read_b64_tr_b16.  One workgroup per (b,h), 4 waves split the tokens.
//  fwd     : per 16 tokens 6 MFMAs: U^T = [S_hi + S_lo] relu(q)^T, swapped so a lane owns 4 consecutive
//            c' of one token (8-B stores); the denominator row is broadcast with one shuffle.
//  bwd q   : recompute U, dU = dO/den, dU[32] = -(dO.O)/den, dq = dU S (accumulator-as-operand, S^T
//            fragments in the accumulator's k order), and the partial dS = dU^T relu(q) of the
//            workgroup's 256 tokens (dU goes through a wave-private LDS image to get tokens onto k).
//  bwd kv  : dv = relu(k) dS[:32]^T, dk = [v;1] dS masked by k > 0.
// (An earlier VALU version with the state read through the scalar cache took 9 ms per SANA-1.6B step.)
#include "common.hpp"
#include "../../include/yat_hip.h"

namespace {

constexpr int C = 32;          // head dim
constexpr int SS = 33 * C;     // floats per state
constexpr int TB = 256;        // tokens per workgroup in the apply kernels (4 waves x 64)

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.0f;
    return z;
}
__device__ __forceinline__ bf16x8 relu8(bf16x8 v) {
    u32x4 u = __builtin_bit_cast(u32x4, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t neg = (u[i] >> 15) & 0x00010001u;       // sign bits of the two halves
        u[i] &= ~(neg * 0xffffu);
    }
    return __builtin_bit_cast(bf16x8, u);
}
// fp32 x[8] -> bf16 hi and lo fragments with hi + lo ~= x
__device__ __forceinline__ void split8(const float* x, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)x[e];
        hi[e] = h;
        lo[e] = (__bf16)(x[e] - (float)h);
    }
}
// rows of a [33][32] fp32 state as an operand fragment with NATURAL k order:
// idx = row0 + (lane & 15) (zero beyond `nrows`), k = 8*(lane>>4) + j.
__device__ __forceinline__ void state_frag_rows(const float* S, int row0, int nrows, int lane, bf16x8& hi, bf16x8& lo) {
    float x[8];
    const int r = row0 + (lane & 15);
    if (r < nrows) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(S + r * C + 8 * (lane >> 4));
        const f32x4 b = *reinterpret_cast<const f32x4*>(S + r * C + 8 * (lane >> 4) + 4);
        x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 0.f;
    }
    split8(x, hi, lo);
}
// columns of the state (S^T) as an operand fragment: idx = c = col0 + (lane & 15); the contraction index is
// the state ROW c' in {acc order: 4g+j (j<4), 16+4g+(j-4)} or {natural order: 8g+j}, rows >= 32 excluded.
template <bool ACC_ORDER>
__device__ __forceinline__ void state_frag_cols(const float* S, int col0, int lane, bf16x8& hi, bf16x8& lo) {
    float x[8];
    const int c = col0 + (lane & 15), g = lane >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cp = ACC_ORDER ? (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4)) : 8 * g + j;
        x[j] = S[cp * C + c];
    }
    split8(x, hi, lo);
}
__device__ __forceinline__ void acc2frag_hilo(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    split8(x, hi, lo);
}
// row-operand fragment of a token-major [n][32] head slice straight from global: idx = token, k = channel
__device__ __forceinline__ bf16x8 tok_frag(const bf16_t* base, int ld, int tok, int N, int lane) {
    bf16x8 z = zero8();
    if (tok < N) z = *reinterpret_cast<const bf16x8*>(base + (int64_t)tok * ld + 8 * (lane >> 4));
    return z;
}
// the same four channels as raw bf16 pairs (to be unpacked later: lets a kernel issue every global load of its token range
// up front -- these kernels run two waves per SIMD and a load consumed right where it is issued costs its full latency)
__device__ __forceinline__ u32x2 tok4_raw(const bf16_t* base, int ld, int tok, int N, int c0) {
    return tok < N ? *reinterpret_cast<const u32x2*>(base + (int64_t)tok * ld + c0) : u32x2{0u, 0u};
}
// 4 consecutive channels (c0..c0+3) of token `tok` as floats (output layout of the swapped MFMA)
__device__ __forceinline__ void tok4(const bf16_t* base, int ld, int tok, int N, int c0, float* o) {
    if (tok < N) unpack4(*reinterpret_cast<const u32x2*>(base + (int64_t)tok * ld + c0), o);
    else { o[0] = o[1] = o[2] = o[3] = 0.f; }
}

// ------------------------------------------------------------------------------------------ state
// grid (H, B); 4 waves, wave w reduces tokens [w*per, (w+1)*per) in 64-token tiles.
// wave-private LDS: K tile [64][32] and V tile [64][32] (64-B rows, 32-B halves swapped on rows with bit 3 set).
__global__ __launch_bounds__(256) void la_state_kernel(int N, const bf16_t* qkv, int ld, int k_off, int v_off,
                                                       uint64_t bytes, float* S_out) {
    __shared__ __attribute__((aligned(16))) char lds[4 * 8192 + 4 * 6 * 64 * 16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int h, b, unused_z;
    xcd_contiguous3(h, b, unused_z);            // adjacent heads (the two halves of a 128-B line) on one XCD
    char* kt = lds + wave * 8192;
    char* vt = kt + 4096;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(qkv, bytes);
    const int per = ((N + 3) / 4 + 63) / 64 * 64;           // tokens per wave, multiple of 64
    const int n_begin = wave * per, n_end = min(N, n_begin + per);
    const int64_t head_off = (int64_t)b * N * ld + h * C;

    f32x4 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones = zero8();
    if ((lane & 15) == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    }
    const uint32_t g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    for (int n0 = n_begin; n0 < n_end; n0 += 64) {
        // DMA: 4 pieces per tile, piece = 16 tokens x 64 B; lane -> (token = l>>2, 16-B slot = l&3)
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const int r = pc * 16 + (lane >> 2);
            const int slot = lane & 3;
            const int chunk = slot ^ (((r >> 3) & 1) << 1);
            const int n = n0 + r;
            const uint32_t vk = n < N ? (uint32_t)((head_off + (int64_t)n * ld + k_off + chunk * 8) * 2) : YAT_OOB;
            const uint32_t vv = n < N ? (uint32_t)((head_off + (int64_t)n * ld + v_off + chunk * 8) * 2) : YAT_OOB;
            lds_dma16(rs, (YAT_LDS void*)(kt + pc * 1024), vk);
            lds_dma16(rs, (YAT_LDS void*)(vt + pc * 1024), vv);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // wave-private tiles: the issuing wave's wait is the ordering
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                       // 32 tokens per MFMA k-step
            bf16x8 kf[2], vf[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const uint32_t col = ct * 16 + 4 * p4;
                const uint32_t r0 = ks * 32 + 8 * g + q4, r1 = r0 + 4;
                const uint32_t c0 = (col >> 3) ^ (((r0 >> 3) & 1) << 1), c1 = (col >> 3) ^ (((r1 >> 3) & 1) << 1);
                kf[ct] = relu8(cat4(lds_read_tr4(kt, r0 * 64 + c0 * 16 + (p4 & 1) * 8), lds_read_tr4(kt, r1 * 64 + c1 * 16 + (p4 & 1) * 8)));
                vf[ct] = cat4(lds_read_tr4(vt, r0 * 64 + c0 * 16 + (p4 & 1) * 8), lds_read_tr4(vt, r1 * 64 + c1 * 16 + (p4 & 1) * 8));
            }
            // tokens past N were zero-filled in K, so the ones row needs no mask
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {                   // D[c][c'] += sum_n relu(k)[n][c] * v'[n][c']
                acc[ct][0] = mfma16(kf[ct], vf[0], acc[ct][0]);
                acc[ct][1] = mfma16(kf[ct], vf[1], acc[ct][1]);
                acc[ct][2] = mfma16(kf[ct], ones, acc[ct][2]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // reads retired before the next DMA overwrites the tiles
    }
    // cross-wave reduction through LDS, fixed order
    float* red = reinterpret_cast<float*>(lds + 4 * 8192);     // [4 waves][6 tiles][64 lanes][4]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4*>(red + ((wave * 6 + ct * 3 + j) * 64 + lane) * 4) = acc[ct][j];
    __syncthreads();
    if (wave == 0) {
        float* S = S_out + ((int64_t)b * gridDim.x + h) * SS;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(red + ((w * 6 + ct * 3 + j) * 64 + lane) * 4);
                    s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
                }
                const int cp = j * 16 + (lane & 15);           // D[row = c = ct*16 + 4g + r][col = c']
                if (cp < 33) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) S[cp * C + ct * 16 + 4 * g + r] = s[r];
                }
            }
    }
}

// U^T tiles for 16 tokens: ut[t][r] = U[n = lane&15][c' = 16t + 4(lane>>4) + r]; 6 MFMAs
__device__ __forceinline__ void u_tiles(const bf16x8 (&shi)[3], const bf16x8 (&slo)[3], bf16x8 qf, f32x4 (&ut)[3]) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        ut[t] = mfma16(shi[t], qf, f32x4{0.f, 0.f, 0.f, 0.f});
        ut[t] = mfma16(slo[t], qf, ut[t]);
    }
}

// ------------------------------------------------------------------------------------------ forward apply
// grid (ceil(N/256), H, B); wave = 64 tokens = 4 groups of 16
__global__ __launch_bounds__(256) void la_fwd_kernel(int N, int H, const bf16_t* qkv, int ld, const float* S_all,
                                                     bf16_t* out, int ld_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int cx, h, b;
    xcd_contiguous3(cx, h, b);                  // chunks of a head, then the next head, on one XCD
    const float* S = S_all + ((int64_t)b * H + h) * SS;
    bf16x8 shi[3], slo[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) state_frag_rows(S, t * 16, 33, lane, shi[t], slo[t]);
    const bf16_t* qb = qkv + (int64_t)b * N * ld + h * C;
    bf16_t* ob = out + (int64_t)b * N * ld_out + h * C;
    const int g = lane >> 4, li = lane & 15;
    // every load of the wave's 64 tokens first (tokens past N read as zero; their stores are guarded): one latency, not four
    bf16x8 qfr[4];
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) qfr[grp] = tok_frag(qb, ld, cx * TB + wave * 64 + grp * 16 + li, N, lane);
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
        const int n = cx * TB + wave * 64 + grp * 16 + li;
        f32x4 ut[3];
        u_tiles(shi, slo, relu8(qfr[grp]), ut);
        const float den = __shfl(ut[2][0], li, 64) + 1e-15f;     // U[32] sits on lanes 0..15 (g = 0, r = 0)
        const float inv = 1.0f / den;
        if (n < N) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
                *reinterpret_cast<u32x2*>(ob + (int64_t)n * ld_out + t * 16 + 4 * g) =
                    pack4(ut[t][0] * inv, ut[t][1] * inv, ut[t][2] * inv, ut[t][3] * inv);
        }
    }
}

// Bank-conflict swizzles of the wave-private images of the backward q kernel (rows = tokens).  Both images are written
// row-wise and read with ds_read_b64_tr_b16, whose 32-lane halves fetch a 32-B block from token rows {0..3, 8..11} + 4n:
//  * dU images, 128-B rows: consecutive rows already alternate between the two halves of the 256-B bank row; XOR the
//    16-B chunk index with 2 bits taken from row bits 1 and 3 (bit 0 of the chunk untouched: a 32-B block stays whole);
//  * q image, 64-B rows: rows r and r+8 would hit the same banks -> swap the row's two 32-B blocks when bit 3 is set.
// (unswizzled, 86 % of this kernel's LDS cycles were bank conflicts: PMC, profiles/r01_g_pmc_per_kernel.txt)
__device__ __forceinline__ uint32_t du_off(uint32_t row, uint32_t byte) {
    const uint32_t sw = (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;
    return row * 128 + ((((byte >> 4) ^ sw) << 4) | (byte & 15));
}
__device__ __forceinline__ uint32_t q_off(uint32_t row, uint32_t byte) {
    return row * 64 + (byte ^ (((row >> 3) & 1) << 5));
}

// ------------------------------------------------------------------------------------------ backward apply (q side)
// grid (nchunks, H, B).  Per wave: 64 tokens.  dS partial slab per workgroup.
__global__ __launch_bounds__(256, 3) void la_bwd_q_kernel(int N, int H, const bf16_t* qkv, int ld, const bf16_t* dout, int ld_do,
                                                       const float* S_all, bf16_t* dqkv, int ld_dq, float* dS_part) {
    // wave-private: dU hi and lo images [32 tokens][64 c'] bf16 (128-B rows) + q image [32 tokens][32 c] bf16 (64-B rows)
    // (the workgroup's final reduction reuses the images' bytes: 40 KB instead of 64 KB per workgroup = three workgroups per CU
    //  instead of two for a kernel that waits on memory)
    __shared__ __attribute__((aligned(16))) char lds[4 * (2 * 4096 + 2048)];
    static_assert(4 * 6 * 64 * 16 <= 4 * (2 * 4096 + 2048), "reduction slab must fit in the images");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int cx, h, b;
    xcd_contiguous3(cx, h, b);                  // chunks of a head, then the next head, on one XCD
    const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = lane & 3;
    const float* S = S_all + ((int64_t)b * H + h) * SS;
    char* du_img = lds + wave * 10240;
    char* du_lo = du_img + 4096;
    char* q_img = du_img + 8192;
    bf16x8 shi[3], slo[3], xhi[2], xlo[2];
#pragma unroll
    for (int t = 0; t < 3; ++t) state_frag_rows(S, t * 16, 33, lane, shi[t], slo[t]);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) state_frag_cols<true>(S, ct * 16, lane, xhi[ct], xlo[ct]);
    float s32[2][4];                                             // S[32][c] for the lane's output channels
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) s32[ct][r] = S[32 * C + ct * 16 + 4 * g + r];
    const bf16_t* qb = qkv + (int64_t)b * N * ld + h * C;
    const bf16_t* dob = dout + (int64_t)b * N * ld_do + h * C;
    bf16_t* dqb = dqkv + (int64_t)b * N * ld_dq + h * C;

    f32x4 ds[3][2];                                              // dS tiles D[c' = 16t + 4g + r][c = 16ct + li]
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) ds[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    // every global load of the wave's 64 tokens first (see tok4_raw): q fragments, dO and q in the output layout
    bf16x8 qfr[4];
    u32x2 d4r[4][2], q4r[4][2];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int n = cx * TB + wave * 64 + s4 * 16 + li;
        qfr[s4] = tok_frag(qb, ld, n, N, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            d4r[s4][t] = tok4_raw(dob, ld_do, n, N, t * 16 + 4 * g);
            q4r[s4][t] = tok4_raw(qb, ld, n, N, t * 16 + 4 * g);
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {                       // 32 tokens per dS k-step
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int n0 = cx * TB + wave * 64 + half * 32 + sub * 16;
            const int n = n0 + li;
            const bf16x8 qf = relu8(qfr[half * 2 + sub]);
            f32x4 ut[3];
            u_tiles(shi, slo, qf, ut);
            const float den = __shfl(ut[2][0], li, 64) + 1e-15f;
            const float inv = 1.0f / den;
            float dot = 0.f;
            f32x4 du[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float d4[4];
                unpack4(d4r[half * 2 + sub][t], d4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dot += d4[r] * (ut[t][r] * inv);             // dO . O
                    du[t][r] = d4[r] * inv;
                }
            }
            dot += __shfl_xor(dot, 16, 64);
            dot += __shfl_xor(dot, 32, 64);
            const float du32 = -dot * inv;                       // dU[n][32], known to every lane of token n
            // dq^T[c][n] = sum_{c' < 32} S[c'][c] dU[n][c']  (accumulator as B operand, hi/lo on both sides) + S[32][c] dU[n][32]
            bf16x8 dhi, dlo;
            acc2frag_hilo(du[0], du[1], dhi, dlo);
            const bool valid = n < N;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                f32x4 dq = mfma16(xhi[ct], dhi, f32x4{0.f, 0.f, 0.f, 0.f});
                dq = mfma16(xhi[ct], dlo, dq);
                dq = mfma16(xlo[ct], dhi, dq);
                float q4v[4];
                unpack4(q4r[half * 2 + sub][ct], q4v);
                if (valid) {
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = q4v[r] > 0.f ? dq[r] + s32[ct][r] * du32 : 0.f;
                    *reinterpret_cast<u32x2*>(dqb + (int64_t)n * ld_dq + ct * 16 + 4 * g) = pack4(o[0], o[1], o[2], o[3]);
                }
            }
            // stash bf16 dU (64 c' per token; c' = 32 holds dU32, the rest of 33..63 zero) and relu(q) for the dS product
            const int trow = sub * 16 + li;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float hi4[4], lo4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { hi4[r] = valid ? rbf(du[t][r]) : 0.f; lo4[r] = valid ? du[t][r] - hi4[r] : 0.f; }
                *reinterpret_cast<u32x2*>(du_img + du_off(trow, (t * 16 + 4 * g) * 2)) = pack4(hi4[0], hi4[1], hi4[2], hi4[3]);
                *reinterpret_cast<u32x2*>(du_lo + du_off(trow, (t * 16 + 4 * g) * 2)) = pack4(lo4[0], lo4[1], lo4[2], lo4[3]);
            }
            {
                const float h32 = (valid && g == 0) ? rbf(du32) : 0.f, l32 = (valid && g == 0) ? du32 - h32 : 0.f;
                *reinterpret_cast<u32x2*>(du_img + du_off(trow, (32 + 4 * g) * 2)) = pack4(h32, 0.f, 0.f, 0.f);
                *reinterpret_cast<u32x2*>(du_lo + du_off(trow, (32 + 4 * g) * 2)) = pack4(l32, 0.f, 0.f, 0.f);
            }
            *reinterpret_cast<u32x2*>(du_img + du_off(trow, (48 + 4 * g) * 2)) = u32x2{0u, 0u};
            *reinterpret_cast<u32x2*>(du_lo + du_off(trow, (48 + 4 * g) * 2)) = u32x2{0u, 0u};
            *reinterpret_cast<bf16x8*>(q_img + q_off(trow, g * 16)) = qf;      // lane holds channels 8g..8g+7 of token li
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // wave-private images written
        // dS[c'][c] += sum_n dU[n][c'] relu(q)[n][c]: tokens on k via transposed reads (natural k order both sides)
        bf16x8 af[3], al[3], bfr[2];
        const uint32_t r0 = 8 * g + q4, r1 = r0 + 4;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const uint32_t off = (t * 16 + 4 * p4) * 2;
            af[t] = cat4(lds_read_tr4(du_img, du_off(r0, off)), lds_read_tr4(du_img, du_off(r1, off)));
            al[t] = cat4(lds_read_tr4(du_lo, du_off(r0, off)), lds_read_tr4(du_lo, du_off(r1, off)));
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const uint32_t off = (ct * 16 + 4 * p4) * 2;
            bfr[ct] = cat4(lds_read_tr4(q_img, q_off(r0, off)), lds_read_tr4(q_img, q_off(r1, off)));
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                ds[t][ct] = mfma16(af[t], bfr[ct], ds[t][ct]);
                ds[t][ct] = mfma16(al[t], bfr[ct], ds[t][ct]);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // workgroup partial: sum the 4 waves in LDS (fixed order), write slab [33][32]
    __syncthreads();                            // every wave is done with its images: their bytes become the reduction slab
    float* red = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) *reinterpret_cast<f32x4*>(red + ((wave * 6 + t * 2 + ct) * 64 + lane) * 4) = ds[t][ct];
    __syncthreads();
    if (wave == 0) {
        float* slab = dS_part + (((int64_t)b * H + h) * gridDim.x + cx) * SS;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(red + ((w * 6 + t * 2 + ct) * 64 + lane) * 4);
                    s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cp = t * 16 + 4 * g + r;
                    if (cp < 33) slab[cp * C + ct * 16 + li] = s[r];
                }
            }
    }
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

// ------------------------------------------------------------------------------------------ backward apply (k, v side)
__global__ __launch_bounds__(256) void la_bwd_kv_kernel(int N, int H, const bf16_t* qkv, int ld, int k_off, int v_off,
                                                        const float* dS_all, bf16_t* dqkv, int ld_dq) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int cx, h, b;
    xcd_contiguous3(cx, h, b);                  // chunks of a head, then the next head, on one XCD
    const int g = lane >> 4, li = lane & 15;
    const float* dS = dS_all + ((int64_t)b * H + h) * SS;
    bf16x8 rhi[2], rlo[2], chi[2], clo[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) state_frag_rows(dS, t * 16, 32, lane, rhi[t], rlo[t]);       // dv: rows c' < 32, k = c
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) state_frag_cols<false>(dS, ct * 16, lane, chi[ct], clo[ct]);   // dk: cols c, k = c' < 32
    float s32[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) s32[ct][r] = dS[32 * C + ct * 16 + 4 * g + r];
    const bf16_t* kb = qkv + (int64_t)b * N * ld + h * C + k_off;
    const bf16_t* vb = qkv + (int64_t)b * N * ld + h * C + v_off;
    bf16_t* dkb = dqkv + (int64_t)b * N * ld_dq + h * C + k_off;
    bf16_t* dvb = dqkv + (int64_t)b * N * ld_dq + h * C + v_off;
    bf16x8 kfr[4], vfr[4];
    u32x2 k4r[4][2];
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {          // every load of the wave's 64 tokens first (see tok4_raw)
        const int n = cx * TB + wave * 64 + grp * 16 + li;
        kfr[grp] = tok_frag(kb, ld, n, N, lane);
        vfr[grp] = tok_frag(vb, ld, n, N, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) k4r[grp][t] = tok4_raw(kb, ld, n, N, t * 16 + 4 * g);
    }
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
        const int n = cx * TB + wave * 64 + grp * 16 + li;
        const bf16x8 kf = relu8(kfr[grp]);
        const bf16x8 vf = vfr[grp];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // dv^T[c'][n] = sum_c dS[c'][c] relu(k)[n][c]
            f32x4 dv = mfma16(rhi[t], kf, f32x4{0.f, 0.f, 0.f, 0.f});
            dv = mfma16(rlo[t], kf, dv);
            // dk^T[c][n] = sum_{c' < 32} dS[c'][c] v[n][c'] + dS[32][c]
            f32x4 dk = mfma16(chi[t], vf, f32x4{0.f, 0.f, 0.f, 0.f});
            dk = mfma16(clo[t], vf, dk);
            float k4[4];
            unpack4(k4r[grp][t], k4);
            if (n < N) {
                *reinterpret_cast<u32x2*>(dvb + (int64_t)n * ld_dq + t * 16 + 4 * g) = pack4(dv[0], dv[1], dv[2], dv[3]);
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = k4[r] > 0.f ? dk[r] + s32[t][r] : 0.f;
                *reinterpret_cast<u32x2*>(dkb + (int64_t)n * ld_dq + t * 16 + 4 * g) = pack4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

}  // namespace

extern "C" {

uint64_t yat_linear_attn_workspace_bytes(int B, int N, int H) {
    const uint64_t nchunks = (N + TB - 1) / TB;
    return (uint64_t)B * H * SS * sizeof(float) * (2 + nchunks);
}

static int la_check(int B, int N, int H, int ld, int k_off, int v_off) {
    if (B <= 0 || N <= 0 || H <= 0 || (ld & 7) || (k_off & 7) || (v_off & 7)) return YAT_EINVAL;
    if ((uint64_t)B * N * ld * 2 > 0x7fffffffull) return YAT_EINVAL;
    return YAT_OK;
}

int yat_linear_attn_fwd(int B, int N, int H, const void* qkv, int ld, int k_off, int v_off, void* out, int ld_out,
                        void* workspace, yat_stream_t stream) {
    if (la_check(B, N, H, ld, k_off, v_off) || (ld_out & 7) || !qkv || !out || !workspace) return YAT_EINVAL;
    float* S = (float*)workspace;
    hipLaunchKernelGGL(la_state_kernel, dim3(H, B), dim3(256), 0, (hipStream_t)stream, N, (const bf16_t*)qkv, ld, k_off, v_off,
                       (uint64_t)B * N * ld * 2, S);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(la_fwd_kernel, dim3((N + TB - 1) / TB, H, B), dim3(256), 0, (hipStream_t)stream, N, H,
                       (const bf16_t*)qkv, ld, (const float*)S, (bf16_t*)out, ld_out);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

int yat_linear_attn_bwd(int B, int N, int H, const void* qkv, int ld, int k_off, int v_off, const void* dout, int ld_dout,
                        void* dqkv, int ld_dqkv, const float* state, void* workspace, yat_stream_t stream) {
    if (la_check(B, N, H, ld, k_off, v_off) || (ld_dout & 7) || (ld_dqkv & 7) || !qkv || !dout || !dqkv || !workspace)
        return YAT_EINVAL;
    const int nchunks = (N + TB - 1) / TB;
    float* S = (float*)workspace;
    float* dS = S + (int64_t)B * H * SS;
    float* part = dS + (int64_t)B * H * SS;
    hipStream_t st = (hipStream_t)stream;
    if (state) {
        S = const_cast<float*>(state);          // the forward's state (first B*H*33*32 floats of its workspace), kept by the caller
    } else {
        hipLaunchKernelGGL(la_state_kernel, dim3(H, B), dim3(256), 0, st, N, (const bf16_t*)qkv, ld, k_off, v_off,
                           (uint64_t)B * N * ld * 2, S);
        YAT_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(la_bwd_q_kernel, dim3(nchunks, H, B), dim3(256), 0, st, N, H, (const bf16_t*)qkv, ld,
                       (const bf16_t*)dout, ld_dout, (const float*)S, (bf16_t*)dqkv, ld_dqkv, part);
    YAT_CHECK_LAUNCH();
    const int64_t tot = (int64_t)B * H * SS;
    hipLaunchKernelGGL(la_reduce_slabs_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, B * H, nchunks,
                       (const float*)part, dS);
    YAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(la_bwd_kv_kernel, dim3(nchunks, H, B), dim3(256), 0, st, N, H, (const bf16_t*)qkv, ld, k_off, v_off,
                       (const float*)dS, (bf16_t*)dqkv, ld_dqkv);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

}  // extern "C"
