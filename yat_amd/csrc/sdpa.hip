// Masked softmax cross-attention (SANA attn2) forward + backward on MFMA for gfx950.
//
// Restates F.scaled_dot_product_attention(q, k, v, attn_mask=bias) as reached through diffusers
// AttnProcessor2_0 at /root/reference/utils/patch_sana_attention_layers.py:98-104, with the
// additive key bias built at /root/reference/utils/patched_sana_transformer.py:275-277
// ((1 - mask) * -10000, no -inf).  Shapes at SANA-1.6B: 20 heads x 112, N = 1024 queries,
// T = 512 padded keys of which only kv_len[b] are real -> key tiles past kv_len are skipped
// (exact: their probabilities are exp(-9984 + ..) == 0 in fp32).
//
// Flash-style, 64-lane waves, MFMA 16x16x32 bf16, head dim padded to 128 in LDS by the LDS-DMA
// range check.  Every product is issued with swapped operands so the softmax row (query) index
// sits on the lane (lane & 15): row max / sum need two shuffles, the probability accumulators are
// directly the B operand of the next MFMA (no LDS round trip; the k order of that MFMA is the
// accumulator's own order, and the other operand is fetched in the same order by
// ds_read_b64_tr_b16 from a row-major tile).
//   forward : workgroup = 64 queries (4 waves x 16), loop over 64-key tiles, online softmax.
//   backward: dq kernel (same decomposition, recomputes P, also emits delta = rowsum(dO*O));
//             dkv kernel (workgroup = 64 keys, loops over 64-query tiles); no atomics.
#include <cstdlib>
#include <type_traits>
#include "common.hpp"
#include "../../include/yat_hip.h"

// Diagnostic build -DYAT_SDPA_STAMPS (scripts/sdpa_stamps.py): the waves of one workgroup sum, per loop segment, the s_memtime
// ticks they spent there (backward kernels); the product build contains none of this.
#ifdef YAT_SDPA_STAMPS
__device__ unsigned int yat_sdpa_stamp_buf[8 * 16 + 8];
// per workgroup: 100 MHz time at kernel entry, loop entry, loop exit, kernel exit; HW_ID; XCC_ID (wave 0 writes)
__device__ unsigned int yat_sdpa_wg_times[4096 * 6];
#define SD_WG_T(slot)                                                                                              \
    do {                                                                                                            \
        const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                        \
        if (wg_ < 4096 && threadIdx.x == 0) {                                                                       \
            yat_sdpa_wg_times[wg_ * 6 + slot] = (uint32_t)__builtin_amdgcn_s_memrealtime();                         \
            if (slot == 0) {                                                                                        \
                yat_sdpa_wg_times[wg_ * 6 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                         \
                yat_sdpa_wg_times[wg_ * 6 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                        \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)
#define SD_STAMP(slot)                                                  \
    do {                                                                \
        __builtin_amdgcn_sched_barrier(0);                              \
        const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime();   \
        st_sum[slot] += now_ - st_prev;                                 \
        st_prev = now_;                                                 \
        __builtin_amdgcn_sched_barrier(0);                              \
    } while (0)
#define SD_STAMP_BEGIN() uint32_t st_sum[16] = {}; uint32_t st_prev = (uint32_t)__builtin_amdgcn_s_memtime(); int st_n = 0; \
    const uint32_t st_t0 = st_prev; const uint32_t st_r0 = (uint32_t)__builtin_amdgcn_s_memrealtime()
#define SD_STAMP_END(cond)                                                                  \
    do {                                                                                    \
        if ((cond) && lane == 0) {                                                          \
            for (int i_ = 0; i_ < 16; ++i_) yat_sdpa_stamp_buf[wave * 16 + i_] = st_sum[i_];  \
            yat_sdpa_stamp_buf[128] = (unsigned)st_n;                                        \
            if (wave == 0) { yat_sdpa_stamp_buf[129] = (uint32_t)__builtin_amdgcn_s_memtime() - st_t0;       \
                             yat_sdpa_stamp_buf[130] = (uint32_t)__builtin_amdgcn_s_memrealtime() - st_r0; } \
        }                                                                                   \
    } while (0)
#else
#define SD_WG_T(slot) do {} while (0)
#define SD_STAMP(slot) do {} while (0)
#define SD_STAMP_BEGIN() do {} while (0)
#define SD_STAMP_END(cond) do {} while (0)
#endif

// Wave priority by loop phase (s_setprio): raised while a wave issues its MFMA phases, dropped for its softmax.  Two workgroups
// share a CU and their waves a SIMD at unrelated points of the same loop; with the priority the wave that has matrix work
// gets the issue slots first and the partner's exponentials fill the gaps (and the s_setprio keeps the compiler from mixing
// the phases).  Measured at N = T = 4096 (git history: scripts/gpu_attn_libs.sh; profiles/r04_l_*): dK/dV kernel 1.43 -> 1.36 ms (dh 72) and
// 1.88 -> 1.80 ms (dh 64); forward 1.02 -> 0.98 ms at dh 64, level at dh 72; the dQ kernel 1 % slower with it -> not there.
// The opposite assignment (softmax high) costs the forward 3..7 %.
// A tie-break between the two waves of a SIMD (matrix phases at level 2 or 3 by the parity of the wave slot, HW_ID bit 0, so that
// waves reaching their MFMA phases together do not take turns) was measured in round 5 and is slower: forward 795 -> 827 us (dh 72),
// 974 -> 1043 (dh 64), dK/dV 1360 -> 1397; the waves are not in step to begin with (profiles/r04_l_attention_phase_stamps.txt:
// the partner wave adds 14 % to a wave's tile time).  profiles/r05_w_attention_priority_tiebreak.txt.
#ifndef YAT_SDPA_PRIO_DKV
#define YAT_SDPA_PRIO_DKV 1
#endif
#ifndef YAT_SDPA_PRIO_DQ
#define YAT_SDPA_PRIO_DQ 0
#endif
#ifndef YAT_SDPA_PRIO_FWD
#define YAT_SDPA_PRIO_FWD 1
#endif
#define SD_PRIO(on, level) do { if (on) __builtin_amdgcn_s_setprio(level); } while (0)

namespace {

constexpr int TILE = 16384;  // one [64][128] bf16 image
constexpr int DKV_STAGE = 2 * TILE + 512;   // one query-tile stage of the dK/dV kernel: Q, dO (one image each), lse, delta

enum { IMG_ROW = 0, IMG_TR = 1 };

// Staging a [64 rows][128 cols] bf16 tile image by LDS-DMA (rows past the limit / cols past dh read as zero).  What a lane
// contributes to a tile -- (row within the tile, 16-byte chunk) of each of the 16 / NW pieces its wave issues -- never changes,
// so its byte offsets relative to the tile's first row are loop-invariant registers (``TileSrc``); the tile's position rides
// in the buffer RESOURCE, rebuilt per tile on the scalar unit: base = first row of the tile (at the head's columns), size =
// the rows left below the limit, so the hardware range check zero-fills the rows past it.  (The first version computed a
// 64-bit row index, a compare and a select per piece and lane for every tile: ~55 vector instructions per 64-key tile in
// loops that are bound by vector issue, not by the matrix pipe.)
template <int NW = 4>
struct TileSrc { uint32_t voff[16 / NW]; };
template <int IMG, int NW = 4>
__device__ __forceinline__ TileSrc<NW> tile_src(int ld, int dh, int wave, int lane) {
    TileSrc<NW> t;
#pragma unroll
    for (int j = 0; j < 16 / NW; ++j) {
        const int piece = j * NW + wave;
        const int r = piece * 4 + (lane >> 4);
        const int slot = lane & 15;
        const int chunk = IMG == IMG_ROW ? (slot ^ (r & 15)) : (slot ^ ((r & 7) << 1));
        t.voff[j] = chunk * 8 < dh ? (uint32_t)((r * ld + chunk * 8) * 2) : YAT_OOB;
    }
    return t;
}
// rows [row0, row_limit) of a matrix with row stride ld, seen from column col0: offset (r * ld + c) * 2 is in range iff
// r < row_limit - row0 (for every c < ld; columns past the head are switched off per lane in tile_src)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const bf16_t* base, int64_t row0, int64_t row_limit, int ld,
                                                            int col0) {
    const int64_t rows = row_limit - row0;
    return make_rsrc(base + row0 * ld + col0, rows > 0 ? (uint64_t)rows * ld * 2 : 0);
}
template <int NW = 4>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds, const TileSrc<NW>& src, int wave) {
#pragma unroll
    for (int j = 0; j < 16 / NW; ++j) lds_dma16(rsrc, (YAT_LDS void*)(lds + (j * NW + wave) * 1024), src.voff[j]);
}
// operand fragment, natural k order, from a ROW image: idx = row0 + (lane&15), k = ks*32 + 8*(lane>>4) + j
__device__ __forceinline__ bf16x8 frag_row(const char* lds, int row0, int ks, int lane) {
    const uint32_t r = row0 + (lane & 15);
    const uint32_t c = (ks * 4 + (lane >> 4)) ^ (r & 15);
    return lds_read8(lds, r * 256 + c * 16);
}
// same fragment from a TR-swizzled image (chunk ^ ((r&7)<<1)): the eight rows a ds_read_b128 serves together still land in
// eight distinct 16-byte bank groups, so one K image can feed both Q K^T (this) and dS K (frag_tr_acc)
__device__ __forceinline__ bf16x8 frag_row_tr(const char* lds, int row0, int ks, int lane) {
    const uint32_t r = row0 + (lane & 15);
    const uint32_t c = (ks * 4 + (lane >> 4)) ^ ((r & 7) << 1);
    return lds_read8(lds, r * 256 + c * 16);
}
// operand fragment in ACCUMULATOR k order from a TR image: idx = col0 + (lane&15);
// k slot (g, j): row = krow0 + 4g + j (j < 4), krow0 + 16 + 4g + (j - 4) (j >= 4)
__device__ __forceinline__ bf16x8 frag_tr_acc(const char* lds, int krow0, int col0, int lane) {
    const uint32_t g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const uint32_t col = col0 + 4 * p;
    const uint32_t r0 = krow0 + 4 * g + q, r1 = r0 + 16;
    const uint32_t c0 = (col >> 3) ^ ((r0 & 7) << 1), c1 = (col >> 3) ^ ((r1 & 7) << 1);
    return cat4(lds_read_tr4(lds, r0 * 256 + c0 * 16 + (p & 1) * 8), lds_read_tr4(lds, r1 * 256 + c1 * 16 + (p & 1) * 8));
}
// operand fragment straight from global memory (row-operand layout), zero padded
__device__ __forceinline__ bf16x8 frag_global(const bf16_t* base, int64_t row, int64_t row_limit, int ld, int col0, int dh,
                                              int ks, int lane) {
    const int d = ks * 32 + 8 * (lane >> 4);
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.0f;
    if (row < row_limit && d < dh) z = *reinterpret_cast<const bf16x8*>(base + row * ld + col0 + d);
    return z;
}
// two accumulator tiles (keys/queries 16*(2s) and 16*(2s+1)) -> one bf16 operand fragment
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x4& a, const f32x4& b) {
    bf16x8 r;
    r[0] = (__bf16)a[0]; r[1] = (__bf16)a[1]; r[2] = (__bf16)a[2]; r[3] = (__bf16)a[3];
    r[4] = (__bf16)b[0]; r[5] = (__bf16)b[1]; r[6] = (__bf16)b[2]; r[7] = (__bf16)b[3];
    return r;
}
__device__ __forceinline__ float group_max(float v) {  // across the 4 lane groups that share lane&15
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

constexpr float LOG2E = 1.4426950408889634f;
// exp(x - m) as one FMA + v_exp_f32: exp2(x * log2e - m * log2e); `nm2` = -m * log2e is per row
__device__ __forceinline__ float exp_sub(float x, float nm2) { return __builtin_amdgcn_exp2f(__builtin_fmaf(x, LOG2E, nm2)); }
// keys past T in the last key tile: their scores must vanish from the softmax.  The DMA range check zero-fills their
// bias slot; one wave rewrites those slots to -1e30 (finite: exp2 of it is 0) so the per-score loop needs no bounds test.
__device__ __forceinline__ void mask_tail_bias(char* bias_slot, int k0, int T, int wave, int lane) {
    if (k0 + 64 > T) {                           // uniform
        if (wave == 0 && k0 + lane >= T) reinterpret_cast<float*>(bias_slot)[lane] = -1e30f;
        __syncthreads();
    }
}

struct SdpaP {
    int N, T, H, dh; float scale;
    const bf16_t* q; int ldq; const bf16_t* k; const bf16_t* v; int ldkv;
    const float* bias; const int* kv_len;
    bf16_t* out; int ldo; float* lse;
    // backward
    const bf16_t* dout; int lddo; float* delta; bf16_t* dq; int lddq; bf16_t* dk; bf16_t* dv; int lddkv;
    uint64_t q_bytes, kv_bytes, do_bytes;
    uint64_t stat_bytes;             // bytes of the lse / delta arrays (B*H*N*4)
    uint64_t bias_bytes;             // bytes of the key-bias array (B*T*4)
    const int* work; int n_work;     // dK/dV kernel: compact list of (batch, key tile) pairs, or null for the dense grid
    int xcd_remap;                   // workgroup index -> (tile, head, image) so that an XCD owns contiguous (image, head) runs
    // Packed keys (yat_sdpa_*_packed): image b's K / V (and dK / dV) rows are [kv_off[b], kv_off[b] + kv_len[b]) of one
    // matrix without padding rows; key_bias / kv_len keep their [B, T] / [B] layout.  Null: rows [b T, b T + T).
    const int* kv_off;
};
// first K / V row of image b and the row limit of its keys (rows at or past it read as zero and are never written)
__device__ __forceinline__ int64_t kv_row0(const SdpaP& p, int b) { return p.kv_off ? (int64_t)p.kv_off[b] : (int64_t)b * p.T; }
__device__ __forceinline__ int64_t kv_row_limit(const SdpaP& p, int64_t r0, int klim) { return r0 + (p.kv_off ? klim : p.T); }

// ------------------------------------------------------------------------------------------ forward
// One stage = K (ROW image) + V (TR image) + the 64 key-bias floats.  Two stages in LDS: tile t+1 (and its bias, by
// 4-byte LDS-DMA -- an ordinary global load inside the loop would drain the DMAs with its vmcnt(0)) is in flight while
// tile t is consumed, one barrier per tile.  With ~3 live key tiles per image the loop is latency, not MFMA, bound.
//
// Long key loops (self-attention over thousands of keys) are bound by VECTOR ISSUE, not by the matrix pipe: per 64-key
// tile and wave (dh 72, three 16-query sub-tiles) 66 MFMAs = 1056 pipe cycles stood against ~2100 issue cycles (MFMA 8,
// v_exp 8, everything else 4 each: MI355X_MICROARCH.md "vector-instruction ISSUE cost") -- 51 exps and ~290 other vector
// instructions.  The exps are the minimum; what the template switches remove is the rest:
//   NOBIAS  no key bias and every key attends (self-attention): the scale rides in the exp's own FMA --
//           exp2(s * (scale log2e) - m * (scale log2e)) with the running maximum kept on the raw scores -- so the
//           per-score scale/bias FMA, the bias tile and its LDS reads go; keys past T in the last tile are masked there only.
//   ONES    the row sums come out of the P V product: column dh of the V image (a padding column: dh < 16 DT) holds 1.0 --
//           written once, the DMA lanes that would zero-fill it are switched off -- so o[.][column dh] accumulates
//           sum_k bf16(P) with every rescale applied, and the per-score adds and their cross-lane reduction go.
//   (both)  LAZY RESCALE: the output accumulators are multiplied by exp(m_old - m_new) only when some row's tile maximum
//           exceeds the running reference by more than 2^8 (then every row moves to its true maximum); otherwise the
//           reference stays, exp2 arguments stay <= 8 and P <= 256 -- exact in fp32 / bf16, the final O / l ratio is the
//           same number.  After the first tile this is rare, and the 60 multiplies + 3 exps per tile go with it.
//   The cross-lane maxima use v_permlane16/32_swap (two VALU instructions per step) instead of ds_bpermute.
constexpr int FWD_STAGE = 2 * TILE + 256;
constexpr float LAZY_LOG2 = 8.0f;                   // rescale threshold, in units of log2 (P stays <= 2^8)

__device__ __forceinline__ float vmaxf(float a, float b) {      // v_max_f32 without the canonicalising self-max
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// reductions across the 4 lane groups that share lane & 15, on the VALU: swap odd / even 16-lane rows, then 32-lane halves
__device__ __forceinline__ float group_max_swap(float v) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    u2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = vmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    u2 c = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmaxf(__uint_as_float(c[0]), __uint_as_float(c[1]));
}
__device__ __forceinline__ float group_sum_swap(float v) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    u2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    u2 c = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(c[0]) + __uint_as_float(c[1]);
}
#define YAT_SKIP 0xffffffffu                        // TileSrc offset of a lane that does not take part in the DMA (ONES)

template <int KS, int DT, int QS, bool NOBIAS, bool ONES>
__global__ __launch_bounds__(256) void sdpa_fwd_kernel(SdpaP p) {
    // QS = 16-query sub-tiles per wave: the workgroup covers 64*QS queries.  With QS = 2 every K / V fragment read from LDS
    // feeds two MFMAs and the per-tile costs (9 LDS-DMA issues per wave, the barrier) are shared by twice the work -- the
    // long-sequence variant (self-attention over thousands of keys); QS = 1 keeps more workgroups for short key loops.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    SD_WG_T(0);
    const int g = lane >> 4, li = lane & 15;
    // the query blocks of one (image, head) read the same K / V: give them to one XCD (one L2) instead of all eight
    int bx, h, b;
    if (p.xcd_remap == 2) xcd_contiguous3_zfast(bx, h, b);
    else if (p.xcd_remap) xcd_contiguous3(bx, h, b);
    else { bx = blockIdx.x; h = blockIdx.y; b = blockIdx.z; }
    const int q0 = bx * (64 * QS) + wave * (16 * QS);
    const int kvl = (!NOBIAS && p.kv_len) ? p.kv_len[b] : 0;
    const int klim = kvl > 0 ? kvl : p.T;
    const int col0 = h * p.dh;
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.bias, NOBIAS ? 0 : p.bias_bytes);

    const int64_t kvr0 = kv_row0(p, b), kvrl = kv_row_limit(p, kvr0, klim);
    const TileSrc<> src_k = tile_src<IMG_ROW>(p.ldkv, p.dh, wave, lane);
    TileSrc<> src_v = tile_src<IMG_TR>(p.ldkv, p.dh, wave, lane);
    if constexpr (ONES) {
        // column dh of the V image = 1.0 in every row of both stages, never touched by the DMA again
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = j * 4 + wave, r = piece * 4 + (lane >> 4);
            const int chunk = (lane & 15) ^ ((r & 7) << 1);
            if (chunk * 8 == p.dh) {
                src_v.voff[j] = YAT_SKIP;
#pragma unroll
                for (int st = 0; st < 2; ++st)
                    *reinterpret_cast<u32x4*>(smem + st * FWD_STAGE + TILE + piece * 1024 + lane * 16) = u32x4{0x3F80u, 0u, 0u, 0u};
            }
        }
    }
    auto stage = [&](int k0, char* base) {
        stage_tile(tile_rsrc(p.k, kvr0 + k0, kvrl, p.ldkv, col0), base, src_k, wave);
        const __amdgpu_buffer_rsrc_t rv = tile_rsrc(p.v, kvr0 + k0, kvrl, p.ldkv, col0);
        if constexpr (ONES) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (src_v.voff[j] != YAT_SKIP) lds_dma16(rv, (YAT_LDS void*)(base + TILE + (j * 4 + wave) * 1024), src_v.voff[j]);
        } else {
            stage_tile(rv, base + TILE, src_v, wave);
        }
        if (!NOBIAS && wave == 0) {
            const int key = k0 + lane;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (YAT_LDS void*)(base + 2 * TILE), 4,
                                                     key < p.T ? (uint32_t)(((int64_t)b * p.T + key) * 4) : YAT_OOB, 0, 0, 0);
        }
    };
    stage(0, smem);

    bf16x8 qf[QS][KS];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[qs][ks] = frag_global(p.q, (int64_t)b * p.N + q0 + qs * 16 + li, (int64_t)b * p.N + p.N, p.ldq, col0, p.dh, ks, lane);

    f32x4 o[QS][DT];
    float m[QS], l[QS];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qs][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        m[qs] = -1e30f;
        l[qs] = 0.f;
    }
    // exp2 argument = x * ce - m * ce: x = raw score (NOBIAS, ce = scale log2e) or scaled + biased score (ce = log2e)
    const float ce = NOBIAS ? p.scale * LOG2E : LOG2E;
    const float lazy = LAZY_LOG2 / ce;              // the threshold in the units of m

    int it = 0;
    SD_WG_T(1);
    SD_STAMP_BEGIN();
    for (int k0 = 0; k0 < klim; k0 += 64, ++it) {
        char* cur = smem + (it & 1) * FWD_STAGE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // tile `it` landed for every wave; stage (it+1)&1 is free
        SD_STAMP(0);
        if (k0 + 64 < klim) stage(k0 + 64, smem + ((it + 1) & 1) * FWD_STAGE);
        SD_STAMP(1);
        if constexpr (!NOBIAS) mask_tail_bias(cur + 2 * TILE, k0, p.T, wave, lane);
        const char* Ks = cur;
        const char* Vs = cur + TILE;
        const float* bias_s = reinterpret_cast<const float*>(cur + 2 * TILE);

        SD_PRIO(YAT_SDPA_PRIO_FWD, 1);
        f32x4 s[QS][4];
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) s[qs][nj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kfrag = frag_row(Ks, nj * 16, ks, lane);
#pragma unroll
                for (int qs = 0; qs < QS; ++qs) s[qs][nj] = mfma16(kfrag, qf[qs][ks], s[qs][nj]);
            }
        }
        if constexpr (NOBIAS) {
            if (k0 + 64 > p.T) {                               // uniform: keys past T in the last tile vanish from the softmax
#pragma unroll
                for (int nj = 0; nj < 4; ++nj)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k0 + nj * 16 + 4 * g + r >= p.T)
#pragma unroll
                            for (int qs = 0; qs < QS; ++qs) s[qs][nj][r] = -1e30f;
            }
        } else {
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + nj * 16 + 4 * g);
#pragma unroll
                for (int qs = 0; qs < QS; ++qs)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[qs][nj][r] = __builtin_fmaf(s[qs][nj][r], p.scale, bv[r]);
            }
        }
        SD_STAMP(2);
        SD_PRIO(YAT_SDPA_PRIO_FWD, 0);
        // tile maxima; does any row of this wave outrun its reference by more than the threshold?
        float mx[QS];
        bool grow = false;
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) {
            float t = -1e30f;
#pragma unroll
            for (int nj = 0; nj < 4; ++nj)
#pragma unroll
                for (int r = 0; r < 4; ++r) t = fmaxf(t, s[qs][nj][r]);
            mx[qs] = group_max_swap(t);
            grow |= mx[qs] > m[qs] + lazy;
        }
        if (__builtin_amdgcn_ballot_w64(grow) != 0) {          // uniform; after the first tile: rare
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) {
                const float mn = fmaxf(m[qs], mx[qs]);
                const float alpha = __builtin_amdgcn_exp2f((m[qs] - mn) * ce);
                m[qs] = mn;
                l[qs] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[qs][dt][r] *= alpha;
            }
        }
        bf16x8 pf0[QS], pf1[QS];
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) {
            const float nm2 = -m[qs] * ce;
            float rs = 0.f;
#pragma unroll
            for (int nj = 0; nj < 4; ++nj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qs][nj][r], ce, nm2));
                    s[qs][nj][r] = e;
                    if constexpr (!ONES) rs += e;
                }
            if constexpr (!ONES) l[qs] += group_sum_swap(rs);
            pf0[qs] = acc_to_frag(s[qs][0], s[qs][1]);
            pf1[qs] = acc_to_frag(s[qs][2], s[qs][3]);
        }
        SD_STAMP(3);
        SD_PRIO(YAT_SDPA_PRIO_FWD, 1);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16x8 v0 = frag_tr_acc(Vs, 0, dt * 16, lane), v1 = frag_tr_acc(Vs, 32, dt * 16, lane);
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) {
                o[qs][dt] = mfma16(v0, pf0[qs], o[qs][dt]);
                o[qs][dt] = mfma16(v1, pf1[qs], o[qs][dt]);
            }
        }
        SD_STAMP(4);
#ifdef YAT_SDPA_STAMPS
        ++st_n;
#endif
    }
    SD_STAMP_END(bx == 3 && h == 1 && b == 0);
    SD_WG_T(2);
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        if constexpr (ONES) {
            // the row sum sits in the accumulator of output column dh: lane group (dh % 16) / 4, component dh % 4
            const int dc = p.dh - (DT - 1) * 16;
            const f32x4 t = o[qs][DT - 1];
            const float mine = (dc & 3) == 0 ? t[0] : (dc & 3) == 1 ? t[1] : (dc & 3) == 2 ? t[2] : t[3];
            l[qs] = __shfl(mine, (dc >> 2) * 16 + li, 64);
        }
        const int qi = q0 + qs * 16 + li;
        if (qi < p.N) {
            const float inv = 1.0f / l[qs];
            bf16_t* op = p.out + ((int64_t)b * p.N + qi) * p.ldo + col0;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = dt * 16 + 4 * g;
                if (d < p.dh)
                    *reinterpret_cast<u32x2*>(op + d) =
                        pack4(o[qs][dt][0] * inv, o[qs][dt][1] * inv, o[qs][dt][2] * inv, o[qs][dt][3] * inv);
            }
            // lse in the units of the scaled (+ biased) scores, as the backward kernels read it
            if (g == 0 && p.lse) p.lse[((int64_t)b * p.H + h) * p.N + qi] = m[qs] * (NOBIAS ? p.scale : 1.0f) + __logf(l[qs]);
        }
    }
    SD_WG_T(3);
}

// (A software-pipelined forward for the long no-bias key loops -- the Q K^T product of tile t+1 issued beside the exponentials
// of tile t, K and V in separate double buffers, the loop body twice so that every LDS offset is an immediate -- was built and
// measured in round 3 and was not faster than the kernel above with 256-query workgroups; DESIGN.md section 10.  It lived here
// behind a macro until the end of round 3 (git history: 6014b8f .. 68ea669); removed rather than kept as dead code.)

// ------------------------------------------------------------------------------------------ backward: dQ
// Stage = K (TR-swizzled image: read row-wise for S = Q K^T and transposed for dQ = dS K), V (ROW image), key bias.
// Two stages, one barrier per key tile, as in the forward.
constexpr int DQ_STAGE = 2 * TILE + 256;
template <int KS, int DT, int QS, bool NOBIAS, int NW = 4>
__global__ __launch_bounds__(64 * NW) void sdpa_bwd_dq_kernel(SdpaP p) {
    // NW waves: the workgroup covers 16 * NW * QS queries and shares each key tile's LDS-DMA among them
    // NOBIAS (no key bias, every key attends): P = exp2(s * (scale log2e) - lse log2e) in one FMA + exp, no bias tile.  Keys
    // past T in the last tile need no mask here: their K rows are zero-filled, so whatever dS they get multiplies zeros.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, li = lane & 15;
    int bx, h, b;
    if (p.xcd_remap == 2) xcd_contiguous3_zfast(bx, h, b);
    else if (p.xcd_remap) xcd_contiguous3(bx, h, b);
    else { bx = blockIdx.x; h = blockIdx.y; b = blockIdx.z; }
    const int q0 = bx * (16 * NW * QS) + wave * (16 * QS);               // QS 16-query sub-tiles per wave, as in the forward
    const int kvl = (!NOBIAS && p.kv_len) ? p.kv_len[b] : 0;
    const int klim = kvl > 0 ? kvl : p.T;
    const int col0 = h * p.dh;
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.bias, NOBIAS ? 0 : p.bias_bytes);
    const int64_t qlim = (int64_t)b * p.N + p.N;
    const float ce = p.scale * LOG2E;

    const int64_t kvr0 = kv_row0(p, b), kvrl = kv_row_limit(p, kvr0, klim);
    const TileSrc<NW> src_k = tile_src<IMG_TR, NW>(p.ldkv, p.dh, wave, lane), src_v = tile_src<IMG_ROW, NW>(p.ldkv, p.dh, wave, lane);
    auto stage = [&](int k0, char* base) {
        stage_tile<NW>(tile_rsrc(p.k, kvr0 + k0, kvrl, p.ldkv, col0), base, src_k, wave);
        stage_tile<NW>(tile_rsrc(p.v, kvr0 + k0, kvrl, p.ldkv, col0), base + TILE, src_v, wave);
        if (!NOBIAS && wave == 0) {
            const int key = k0 + lane;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (YAT_LDS void*)(base + 2 * TILE), 4,
                                                     key < p.T ? (uint32_t)(((int64_t)b * p.T + key) * 4) : YAT_OOB, 0, 0, 0);
        }
    };
    stage(0, smem);

    bf16x8 qf[QS][KS], dof[QS][KS];
    float dl[QS], nlse2[QS];                     // nlse2 = -lse * log2e (queries past N: -1e30 -> P = 0)
    f32x4 ndl[QS];                               // -delta in every component: the dP accumulators START there (dP - delta
    f32x4 acc[QS][DT];                           // without a subtraction per score: the MFMA's C operand, no copy)
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        const int qi = q0 + qs * 16 + li;
        const int64_t qrow = (int64_t)b * p.N + qi;
        float d_ = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[qs][ks] = frag_global(p.q, qrow, qlim, p.ldq, col0, p.dh, ks, lane);
            dof[qs][ks] = frag_global(p.dout, qrow, qlim, p.lddo, col0, p.dh, ks, lane);
            const bf16x8 of = frag_global(p.out, qrow, qlim, p.ldo, col0, p.dh, ks, lane);
#pragma unroll
            for (int e = 0; e < 8; ++e) d_ += (float)dof[qs][ks][e] * (float)of[e];
        }
        dl[qs] = group_sum(d_);
        ndl[qs] = f32x4{-dl[qs], -dl[qs], -dl[qs], -dl[qs]};
        nlse2[qs] = qi < p.N ? -p.lse[((int64_t)b * p.H + h) * p.N + qi] * LOG2E : -1e30f;
        if (qi < p.N && g == 0) p.delta[((int64_t)b * p.H + h) * p.N + qi] = dl[qs];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[qs][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    int it = 0;
    SD_STAMP_BEGIN();
    for (int k0 = 0; k0 < klim; k0 += 64, ++it) {
        char* cur = smem + (it & 1) * DQ_STAGE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        SD_STAMP(0);
        if (k0 + 64 < klim) stage(k0 + 64, smem + ((it + 1) & 1) * DQ_STAGE);
        SD_STAMP(1);
        SD_PRIO(YAT_SDPA_PRIO_DQ, 1);
        if constexpr (!NOBIAS) mask_tail_bias(cur + 2 * TILE, k0, p.T, wave, lane);
        const char* Kt = cur;
        const char* Vs = cur + TILE;
        const float* bias_s = reinterpret_cast<const float*>(cur + 2 * TILE);

        f32x4 s[QS][4], dp[QS][4];
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) {
                s[qs][nj] = f32x4{0.f, 0.f, 0.f, 0.f};
                dp[qs][nj] = ndl[qs];
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kfr = frag_row_tr(Kt, nj * 16, ks, lane), vfr = frag_row(Vs, nj * 16, ks, lane);
#pragma unroll
                for (int qs = 0; qs < QS; ++qs) {
                    s[qs][nj] = mfma16(kfr, qf[qs][ks], s[qs][nj]);
                    dp[qs][nj] = mfma16(vfr, dof[qs][ks], dp[qs][nj]);
                }
            }
        }
        SD_STAMP(2);
        SD_PRIO(YAT_SDPA_PRIO_DQ, 0);
        bf16x8 f0[QS], f1[QS];
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) {
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) {
                f32x4 bv;
                if constexpr (!NOBIAS) bv = *reinterpret_cast<const f32x4*>(bias_s + nj * 16 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = NOBIAS ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[qs][nj][r], ce, nlse2[qs]))
                                            : exp_sub(__builtin_fmaf(s[qs][nj][r], p.scale, bv[r]), nlse2[qs]);
                    s[qs][nj][r] = pr * dp[qs][nj][r];              // dS (w.r.t. the scaled logits); dp = dP - delta
                }
            }
            f0[qs] = acc_to_frag(s[qs][0], s[qs][1]);
            f1[qs] = acc_to_frag(s[qs][2], s[qs][3]);
        }
        SD_STAMP(3);
        SD_PRIO(YAT_SDPA_PRIO_DQ, 1);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16x8 k0f = frag_tr_acc(Kt, 0, dt * 16, lane), k1f = frag_tr_acc(Kt, 32, dt * 16, lane);
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) {
                acc[qs][dt] = mfma16(k0f, f0[qs], acc[qs][dt]);
                acc[qs][dt] = mfma16(k1f, f1[qs], acc[qs][dt]);
            }
        }
        SD_STAMP(4);
        SD_PRIO(YAT_SDPA_PRIO_DQ, 0);
#ifdef YAT_SDPA_STAMPS
        ++st_n;
#endif
    }
    SD_STAMP_END(bx == 3 && h == 1 && b == 0);
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        const int qi = q0 + qs * 16 + li;
        if (qi < p.N) {
            bf16_t* dp_ = p.dq + ((int64_t)b * p.N + qi) * p.lddq + col0;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = dt * 16 + 4 * g;
                if (d < p.dh)
                    *reinterpret_cast<u32x2*>(dp_ + d) = pack4(acc[qs][dt][0] * p.scale, acc[qs][dt][1] * p.scale,
                                                               acc[qs][dt][2] * p.scale, acc[qs][dt][3] * p.scale);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
// where the S accumulators of queries [q, q+4) start: -lse / scale (raw-score units)
__device__ __forceinline__ f32x4 score_start(const float* lse_s, int q, float inv_scale) {
    const f32x4 l = *reinterpret_cast<const f32x4*>(lse_s + q);
    return f32x4{-l[0] * inv_scale, -l[1] * inv_scale, -l[2] * inv_scale, -l[3] * inv_scale};
}
// (two waves per SIMD asked for where the kernel fits them -- head dims up to 96; the allocator otherwise lands the dh-72 dense
//  instantiation at 260 registers, one wave per SIMD: 1.41 -> 1.96 ms when that happened in round 3)
template <int KS, int DT, int KB, int NW, bool NOBIAS>
__global__ __launch_bounds__(64 * NW, (KS <= 3 && NW == 4) ? 2 : 1) void sdpa_bwd_dkv_kernel(SdpaP p) {
    // NOBIAS: as in the dQ kernel (the scale folded into the exp's FMA); keys past T need no mask -- their dK / dV rows are
    // not stored and no other key's result depends on them.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // stage layout: Q | dO (TR-swizzled images, read row-wise for S / dP and transposed for dK / dV) | lse[64] | delta[64]
    // (staging a ROW and a TR image of each, as before, made the loop LDS-DMA bound: 64 KB per query tile per CU)
    // KB = 16-key sub-tiles per wave, NW = waves: the workgroup owns KT = 16*KB*NW keys.  NW = 8 is the long-sequence
    // variant (dense grid only): every Q / dO tile staged in LDS serves 128 keys, so the per-wave LDS-DMA issue cost (the
    // bound of this loop at NW = 4: ~18 issues per 44 MFMAs) halves while the per-wave register footprint stays put.
    constexpr int KT = 16 * KB * NW;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, li = lane & 15;
    // Work decomposition: launching early-exit workgroups of this LDS-heavy kernel is NOT free (measured: ~0.24 us
    // per idle workgroup, 230 us for the 70 % idle tiles of a T=512 / kv_len~160 batch), so the host hands a compact
    // (batch, key tile) list built from the embedding lengths it already knows; the dense grid remains as a fallback.
    int bx = blockIdx.x, h = blockIdx.y, bz = blockIdx.z;
    // key tiles of one (image, head) share Q / dO in one L2.  With the compact work list (grid = (work items, heads)) every
    // unit is ONE key tile -- equal work -- so contiguous runs are balanced whatever the lengths; units x fastest = the key
    // tiles of one image, one head, side by side (round 6: the list path ran unmapped, the three tiles of an (image, head)
    // on three XCDs, each fetching its own copy of Q and dO).  Dense grid: mode 2 = images fastest among the pairs (ragged).
    if (p.xcd_remap == 2 && !p.work) xcd_contiguous3_zfast(bx, h, bz);
    else if (p.xcd_remap) xcd_contiguous3(bx, h, bz);
    const int b = p.work ? p.work[2 * bx] : bz;
    const int tile = p.work ? p.work[2 * bx + 1] : bx;
    const int k0 = tile * KT + wave * (16 * KB);
    const int kvl = (!NOBIAS && p.kv_len) ? p.kv_len[b] : 0;
    const int klim = kvl > 0 ? kvl : p.T;
    const int ntiles = (klim + KT - 1) / KT;
    const float ce = p.scale * LOG2E;
    const int col0 = h * p.dh;

    if (tile >= ntiles) return;                 // dense-grid fallback only: masked tile, zeros written by its owner below
    // exact zero gradients for the fully masked key tiles of this (b, h): tile t owns tiles t + ntiles, t + 2 ntiles, ...
    // (packed keys: there are no such rows)
    for (int tz = tile + ntiles; !p.kv_off && tz * KT < p.T; tz += ntiles) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int kz = tz * KT + wave * (16 * KB) + kb * 16 + li;
            if (kz < p.T) {
                bf16_t* zk = p.dk + ((int64_t)b * p.T + kz) * p.lddkv + col0;
                bf16_t* zv = p.dv + ((int64_t)b * p.T + kz) * p.lddkv + col0;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const int d = dt * 16 + 4 * g;
                    if (d < p.dh) {
                        *reinterpret_cast<u32x2*>(zk + d) = u32x2{0u, 0u};
                        *reinterpret_cast<u32x2*>(zv + d) = u32x2{0u, 0u};
                    }
                }
            }
        }
    }
    const TileSrc<NW> src_q = tile_src<IMG_TR, NW>(p.ldq, p.dh, wave, lane), src_do = tile_src<IMG_TR, NW>(p.lddo, p.dh, wave, lane);
    const uint64_t stat_bytes = p.stat_bytes;
    const __amdgpu_buffer_rsrc_t rlse = make_rsrc(p.lse, stat_bytes), rdel = make_rsrc(p.delta, stat_bytes);
    const int64_t kvr0 = kv_row0(p, b), klimrow = kv_row_limit(p, kvr0, klim);
    // The per-score arithmetic is P = exp2(ce * a), dS' = P * d with the accumulators STARTED at what used to be subtracted:
    //   a = (bias - lse) / scale + q.k      (the S chain's C operand; ce * a = (scale q.k + bias - lse) log2e)
    //   d = delta + dO.(-V) = -(dP - delta) (the dP chain's C operand = the delta quad straight from LDS, V negated once)
    // so dS' = -dS, undone by the -scale of the dK store: two multiplies and the exponential per score instead of two FMAs, a
    // subtraction and a multiply, and no lse / delta operand live in the softmax.
    const float inv_scale = 1.0f / p.scale;
    bf16x8 kf[KB][KS], vf[KB][KS];
    float kb_[KB];
    bool kvalid[KB];
    f32x4 adk[KB][DT], adv[KB][DT];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int key = k0 + kb * 16 + li;
        const int64_t krow = kvr0 + key;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[kb][ks] = frag_global(p.k, krow, klimrow, p.ldkv, col0, p.dh, ks, lane);
            vf[kb][ks] = frag_global(p.v, krow, klimrow, p.ldkv, col0, p.dh, ks, lane);
            {                                                               // -V (sign bits; exact): see below
                u32x4 t = __builtin_bit_cast(u32x4, vf[kb][ks]);
                t ^= u32x4{0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
                vf[kb][ks] = __builtin_bit_cast(bf16x8, t);
            }
        }
        kvalid[kb] = krow < klimrow;                // padded layout: key < T; packed: key < kv_len (the next row is another image's)
        // key bias in the units of the RAW scores (the accumulator's); keys past T: P = exp2(-huge) = 0
        kb_[kb] = NOBIAS ? 0.f : (key < p.T ? p.bias[(int64_t)b * p.T + key] * inv_scale : -1e30f);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { adk[kb][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; adv[kb][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }

    // double-buffered query tiles: tile qt+1 is DMA'd while tile qt is consumed
    auto stage_q = [&](int q0, char* base) {
        const int64_t r0 = (int64_t)b * p.N + q0, rl = (int64_t)b * p.N + p.N;
        stage_tile<NW>(tile_rsrc(p.q, r0, rl, p.ldq, col0), base, src_q, wave);
        stage_tile<NW>(tile_rsrc(p.dout, r0, rl, p.lddo, col0), base + TILE, src_do, wave);
        // lse / delta rows by 4-byte LDS-DMA as well: an ordinary VGPR load here would make the compiler wait
        // vmcnt(0) for it -- draining the 16 tile DMAs just issued and undoing the double buffering.  Rows past N read
        // as 0 (range check); their Q and dO rows are zero too, so P stays finite and dS = P * (0 - 0) = 0.
        if (wave < 2) {
            const int qi = q0 + lane;
            const int64_t si = ((int64_t)b * p.H + h) * p.N + qi;
            const uint32_t voff = qi < p.N ? (uint32_t)(si * 4) : YAT_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wave == 0 ? rlse : rdel, (YAT_LDS void*)(base + 2 * TILE + wave * 256), 4,
                                                     voff, 0, 0, 0);
        }
    };
    stage_q(0, smem);
    SD_STAMP_BEGIN();
    for (int q0 = 0, it = 0; q0 < p.N; q0 += 64, ++it) {
        char* cur = smem + (it & 1) * DKV_STAGE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();      // tile `it` landed everywhere; every wave is done with the other stage
        SD_STAMP(0);
        if (q0 + 64 < p.N) stage_q(q0 + 64, smem + ((it + 1) & 1) * DKV_STAGE);
        SD_STAMP(1);
        SD_PRIO(YAT_SDPA_PRIO_DKV, 1);
        const char* Qt = cur;
        const char* Ot = cur + TILE;
        const float* lse_s = reinterpret_cast<const float*>(cur + 2 * TILE);
        const float* del_s = lse_s + 64;

        if constexpr (KB == 1) {
            f32x4 s[KB][4], dp[KB][4];
#pragma unroll
            for (int nq = 0; nq < 4; ++nq) {
                const f32x4 s0 = score_start(lse_s, nq * 16 + 4 * g, inv_scale);
                const f32x4 d0 = *reinterpret_cast<const f32x4*>(del_s + nq * 16 + 4 * g);
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    s[kb][nq] = NOBIAS ? s0 : s0 + kb_[kb];
                    dp[kb][nq] = d0;
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 qfr = frag_row_tr(Qt, nq * 16, ks, lane), ofr = frag_row_tr(Ot, nq * 16, ks, lane);
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) {
                        s[kb][nq] = mfma16(qfr, kf[kb][ks], s[kb][nq]);     // [q = 16nq+4g+r][key = li]
                        dp[kb][nq] = mfma16(ofr, vf[kb][ks], dp[kb][nq]);
                    }
                }
            }
            bf16x8 pf0[KB], pf1[KB], sf0[KB], sf1[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                for (int nq = 0; nq < 4; ++nq)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pr = __builtin_amdgcn_exp2f(s[kb][nq][r] * ce);
                        s[kb][nq][r] = pr;                                      // P
                        dp[kb][nq][r] = pr * dp[kb][nq][r];                     // -dS
                    }
                pf0[kb] = acc_to_frag(s[kb][0], s[kb][1]); pf1[kb] = acc_to_frag(s[kb][2], s[kb][3]);
                sf0[kb] = acc_to_frag(dp[kb][0], dp[kb][1]); sf1[kb] = acc_to_frag(dp[kb][2], dp[kb][3]);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 o0 = frag_tr_acc(Ot, 0, dt * 16, lane), o1 = frag_tr_acc(Ot, 32, dt * 16, lane);
                const bf16x8 q0f = frag_tr_acc(Qt, 0, dt * 16, lane), q1f = frag_tr_acc(Qt, 32, dt * 16, lane);
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    adv[kb][dt] = mfma16(o0, pf0[kb], adv[kb][dt]);
                    adv[kb][dt] = mfma16(o1, pf1[kb], adv[kb][dt]);
                    adk[kb][dt] = mfma16(q0f, sf0[kb], adk[kb][dt]);
                    adk[kb][dt] = mfma16(q1f, sf1[kb], adk[kb][dt]);
                }
            }
    
        } else {
            // two halves of 32 queries: S / dP, the softmax and the P / dS fragments of a half are dead before the next one starts
            // (half the live score registers: 32 keys per wave then fit two waves per SIMD; at 16 keys per wave -- the
            // branch above -- all four 16-query tiles stay in flight, which measured 3-15 % faster there)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 s[KB][2], dp[KB][2];
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    const int nq = 2 * half + hq;
                    const f32x4 s0 = score_start(lse_s, nq * 16 + 4 * g, inv_scale);
                    const f32x4 d0 = *reinterpret_cast<const f32x4*>(del_s + nq * 16 + 4 * g);
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) {
                        s[kb][hq] = NOBIAS ? s0 : s0 + kb_[kb];
                        dp[kb][hq] = d0;
                    }
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const bf16x8 qfr = frag_row_tr(Qt, nq * 16, ks, lane), ofr = frag_row_tr(Ot, nq * 16, ks, lane);
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb) {
                            s[kb][hq] = mfma16(qfr, kf[kb][ks], s[kb][hq]);     // [q = 16nq+4g+r][key = li]
                            dp[kb][hq] = mfma16(ofr, vf[kb][ks], dp[kb][hq]);
                        }
                    }
                }
                SD_STAMP(2 + 3 * half);
                SD_PRIO(YAT_SDPA_PRIO_DKV, 0);
                bf16x8 pf[KB], sf[KB];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // the same arithmetic as the 16-key branch: dense and work-list launches agree bit for bit
                            const float pr = __builtin_amdgcn_exp2f(s[kb][hq][r] * ce);
                            s[kb][hq][r] = pr;                                  // P
                            dp[kb][hq][r] = pr * dp[kb][hq][r];                 // -dS
                        }
                    pf[kb] = acc_to_frag(s[kb][0], s[kb][1]);
                    sf[kb] = acc_to_frag(dp[kb][0], dp[kb][1]);
                }
                SD_STAMP(3 + 3 * half);
                SD_PRIO(YAT_SDPA_PRIO_DKV, 1);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 of = frag_tr_acc(Ot, 32 * half, dt * 16, lane), qf_ = frag_tr_acc(Qt, 32 * half, dt * 16, lane);
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) {
                        adv[kb][dt] = mfma16(of, pf[kb], adv[kb][dt]);
                        adk[kb][dt] = mfma16(qf_, sf[kb], adk[kb][dt]);
                    }
                }
                SD_STAMP(4 + 3 * half);
            }
        }
#ifdef YAT_SDPA_STAMPS
        ++st_n;
#endif
    }
    SD_STAMP_END(tile == 3 && h == 1 && b == 0);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        if (!kvalid[kb]) continue;
        const int key = k0 + kb * 16 + li;
        bf16_t* dkp = p.dk + (kvr0 + key) * p.lddkv + col0;
        bf16_t* dvp = p.dv + (kvr0 + key) * p.lddkv + col0;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = dt * 16 + 4 * g;
            if (d < p.dh) {
                *reinterpret_cast<u32x2*>(dvp + d) = pack4(adv[kb][dt][0], adv[kb][dt][1], adv[kb][dt][2], adv[kb][dt][3]);
                const float ns = -p.scale;                   // the accumulators hold -dS^T Q
                *reinterpret_cast<u32x2*>(dkp + d) = pack4(adk[kb][dt][0] * ns, adk[kb][dt][1] * ns, adk[kb][dt][2] * ns,
                                                           adk[kb][dt][3] * ns);
            }
        }
    }
}

#ifdef YAT_SDPA_ONE_WG            // diagnostic: LDS padded so that one workgroup fits a CU (a wave alone on its SIMD)
constexpr int FWD_LDS = 100 * 1024, DQ_LDS = 100 * 1024, DKV_LDS = 100 * 1024;
#else
constexpr int FWD_LDS = 2 * FWD_STAGE, DQ_LDS = 2 * DQ_STAGE, DKV_LDS = 2 * DKV_STAGE;
#endif

// The LDS images stay 128 columns wide (columns past dh are zero-filled by the DMA range check); what the head dim decides
// is how many of the 32-wide k-steps (KS) and 16-wide output tiles (DT) carry data.  Instantiations: dh <= 32 (SANA's
// softmax variant of attn1), <= 64, <= 80 (PixArt-Sigma: 72), <= 112 (SANA cross-attention), <= 128.
template <int KS, int DT, int QS, bool NOBIAS, bool ONES>
int launch_fwd_qs(const SdpaP& p, int B, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)sdpa_fwd_kernel<KS, DT, QS, NOBIAS, ONES>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                FWD_LDS) != hipSuccess)
            return YAT_EINVAL;
        attr_set = true;
    }
    hipLaunchKernelGGL((sdpa_fwd_kernel<KS, DT, QS, NOBIAS, ONES>), dim3((p.N + 64 * QS - 1) / (64 * QS), p.H, B), dim3(256),
                       FWD_LDS, stream, p);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}
template <int KS, int DT, bool NOBIAS, bool ONES>
int launch_fwd_x(const SdpaP& p, int B, int wide, hipStream_t stream) {
    // the ONES form reads the row sums from output tile DT - 1: the ones column (column dh of the V image) must lie in it
    if constexpr (ONES) {
        if (!((DT - 1) * 16 <= p.dh && p.dh < DT * 16)) return YAT_EINVAL;
    }
    if constexpr (NOBIAS && ONES && KS <= 3) {
        if (wide == 3) return launch_fwd_qs<KS, DT, 4, NOBIAS, ONES>(p, B, stream);      // 256-query workgroups
    }
    if constexpr (KS <= 3) {
        if (wide == 2) return launch_fwd_qs<KS, DT, 3, NOBIAS, ONES>(p, B, stream);      // 192-query workgroups
    }
    return wide ? launch_fwd_qs<KS, DT, 2, NOBIAS, ONES>(p, B, stream) : launch_fwd_qs<KS, DT, 1, NOBIAS, ONES>(p, B, stream);
}
template <int KS, int DT>
int launch_fwd(const SdpaP& p, int B, int wide, hipStream_t stream) {
    return p.bias ? launch_fwd_x<KS, DT, false, false>(p, B, wide, stream) : launch_fwd_x<KS, DT, true, false>(p, B, wide, stream);
}
// no key bias (self-attention): the head dims of the models get the ONES form -- the row sums from a padding column of V,
// which for a head dim that fills its 16-wide output tiles exactly costs one more tile (dh 64: two MFMAs more per 16 queries
// and key tile, against 16 adds and a cross-lane reduction less, in a loop bound by vector issue)
int launch_fwd_nobias(const SdpaP& p, int B, int wide, hipStream_t stream) {
    const int dh = p.dh;
    if (dh == 32) return launch_fwd_x<1, 3, true, true>(p, B, wide, stream);
    if (dh == 64) return launch_fwd_x<2, 5, true, true>(p, B, wide, stream);
    if (dh > 64 && dh < 80) return launch_fwd_x<3, 5, true, true>(p, B, wide, stream);
    if (dh == 112) return launch_fwd_x<4, 8, true, true>(p, B, wide, stream);
    // (dh 104 would need <4, 7>: its ones column sits in output tile 6, not 7 -- not a head dim of any model here)
    return -100;            // no ONES instantiation: the caller falls back to the head-dim classes
}
#ifndef YAT_SDPA_DQ_NW
#define YAT_SDPA_DQ_NW 4
#endif
template <int KS, int DT, int QS>
int launch_dq_qs(const SdpaP& p, int B, hipStream_t stream) {
    constexpr int NW = QS >= 2 ? YAT_SDPA_DQ_NW : 4;
    const void* fn = p.bias ? (const void*)sdpa_bwd_dq_kernel<KS, DT, QS, false, NW> : (const void*)sdpa_bwd_dq_kernel<KS, DT, QS, true, NW>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[p.bias ? 0 : 1]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, DQ_LDS) != hipSuccess) return YAT_EINVAL;
        attr_set[p.bias ? 0 : 1] = true;
    }
    const dim3 grid((p.N + 16 * NW * QS - 1) / (16 * NW * QS), p.H, B);
    if (p.bias) hipLaunchKernelGGL((sdpa_bwd_dq_kernel<KS, DT, QS, false, NW>), grid, dim3(64 * NW), DQ_LDS, stream, p);
    else hipLaunchKernelGGL((sdpa_bwd_dq_kernel<KS, DT, QS, true, NW>), grid, dim3(64 * NW), DQ_LDS, stream, p);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}
template <int KS, int DT>
int launch_dq(const SdpaP& p, int B, int wide, hipStream_t stream) {
    if constexpr (KS <= 2) {
        if (wide == 2) return launch_dq_qs<KS, DT, 3>(p, B, stream);          // 192-query workgroups (dh <= 64)
    }
    return wide ? launch_dq_qs<KS, DT, 2>(p, B, stream) : launch_dq_qs<KS, DT, 1>(p, B, stream);
}
template <int KS, int DT, int KB, int NW>
int launch_dkv_kb(const SdpaP& p, int B, hipStream_t stream) {
    const void* fn = p.bias ? (const void*)sdpa_bwd_dkv_kernel<KS, DT, KB, NW, false> : (const void*)sdpa_bwd_dkv_kernel<KS, DT, KB, NW, true>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[p.bias ? 0 : 1]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS) != hipSuccess) return YAT_EINVAL;
        attr_set[p.bias ? 0 : 1] = true;
    }
    constexpr int KT = 16 * KB * NW;
    const dim3 grid = p.work ? dim3(p.n_work, p.H, 1) : dim3((p.T + KT - 1) / KT, p.H, B);
    if (p.bias) hipLaunchKernelGGL((sdpa_bwd_dkv_kernel<KS, DT, KB, NW, false>), grid, dim3(64 * NW), DKV_LDS, stream, p);
    else hipLaunchKernelGGL((sdpa_bwd_dkv_kernel<KS, DT, KB, NW, true>), grid, dim3(64 * NW), DKV_LDS, stream, p);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}
template <int KS, int DT>
int launch_dkv(const SdpaP& p, int B, int wide, hipStream_t stream) {
    if (wide == 2) return launch_dkv_kb<KS, DT, 2, 4>(p, B, stream);      // 32 keys per wave: measured slower, see below
    return wide ? launch_dkv_kb<KS, DT, 1, 8>(p, B, stream) : launch_dkv_kb<KS, DT, 1, 4>(p, B, stream);
}
#define YAT_SDPA_DISPATCH(fn, dh, ...)                      \
    ((dh) <= 32    ? fn<1, 2>(__VA_ARGS__)                  \
     : (dh) <= 64  ? fn<2, 4>(__VA_ARGS__)                  \
     : (dh) <= 80  ? fn<3, 5>(__VA_ARGS__)                  \
     : (dh) <= 112 ? fn<4, 7>(__VA_ARGS__)                  \
                   : fn<4, 8>(__VA_ARGS__))

// kv_rows: rows of the K / V matrices (B * T in the padded layout, the packed row count otherwise)
int check_common(int B, int N, int T, int H, int dh, int ldq, int ldkv, int64_t kv_rows) {
    if (B <= 0 || N <= 0 || T <= 0 || H <= 0 || dh <= 0 || dh > 128 || (dh & 7) || (ldq & 7) || (ldkv & 7) || kv_rows <= 0)
        return YAT_EINVAL;
    if ((uint64_t)B * N * ldq * 2 > 0x7fffffffull || (uint64_t)kv_rows * ldkv * 2 > 0x7fffffffull) return YAT_EINVAL;
    return YAT_OK;
}

int sdpa_fwd_impl(int B, int N, int T, int H, int dh, float scale, const void* q, int ldq, const void* k, const void* v,
                  int ldkv, const int* kv_off, int64_t kv_rows, const float* key_bias, const int* kv_len, void* out, int ldo,
                  float* lse, yat_stream_t stream) {
    if (check_common(B, N, T, H, dh, ldq, ldkv, kv_rows) || (ldo & 3) || !q || !k || !v || !out) return YAT_EINVAL;
    if (kv_off && !kv_len) return YAT_EINVAL;          // packed keys: every image says how many rows it owns (all >= 1)
    if (!key_bias && (kv_len || kv_off)) return YAT_EINVAL;      // no bias = plain attention over all T keys of every image
    SdpaP p{};
    p.N = N; p.T = T; p.H = H; p.dh = dh; p.scale = scale;
    p.q = (const bf16_t*)q; p.ldq = ldq; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.ldkv = ldkv;
    p.bias = key_bias; p.kv_len = kv_len; p.out = (bf16_t*)out; p.ldo = ldo; p.lse = lse; p.kv_off = kv_off;
    p.q_bytes = (uint64_t)B * N * ldq * 2; p.kv_bytes = (uint64_t)kv_rows * ldkv * 2; p.bias_bytes = (uint64_t)B * T * 4;
    // XCD-contiguous order only for long uniform key loops (self-attention): with ragged kv_len the images with long
    // captions would pile up on one XCD (measured: T = 300 cross-attention 81 -> 104 us; N = T = 4096 1175 -> 1117 us)
    // (round 6: mode 2 -- pairs ordered images-fastest, so that an XCD's run holds whole (image, head) pairs, one L2 copy of
    // their K / V instead of eight, and every image about equally often -- measured on the SANA cross-attention, B = 8, ragged
    // 20..300 keys: forward 44.9 -> 46.0 us, dQ 59.0 -> 63.9 us: the K / V copies were never what these kernels wait for.
    // Not the default; reachable in tuning builds, YAT_SDPA_XCD=2.  profiles/r06_b_cross_attention_xcd_order.txt)
    static const int xcd_env = YAT_TUNE_INT("YAT_SDPA_XCD", -1);
    p.xcd_remap = xcd_env >= 0 ? xcd_env : (T >= 1024);
    // 128-query workgroups once there are enough of them to fill the chip twice over (PixArt-Sigma: N = 4096)
    static const int wide_env = YAT_TUNE_INT("YAT_SDPA_WIDE", -1);
    // ... and 192-query ones (three sub-tiles per wave: K / V fragments feed three MFMAs, 236 registers) while the head dim
    // leaves room for them at two waves per SIMD (dh <= 80): 1.26 -> 1.17 ms at N = T = 4096, dh 72
    int wide = (int64_t)((N + 127) / 128) * H * B >= 1024;
    if (dh <= 80 && (int64_t)((N + 191) / 192) * H * B >= 1024) wide = 2;
    static const int wide4 = YAT_TUNE_INT("YAT_SDPA_WIDE4", 1);
    if (wide4 && !key_bias && dh <= 80 && (int64_t)((N + 255) / 256) * H * B >= 1024) wide = 3;    // (no-bias ONES classes only)
    if (wide_env >= 0) wide = wide_env;
    if (!key_bias) {
        const int rc = launch_fwd_nobias(p, B, wide, (hipStream_t)stream);
        if (rc != -100) return rc;
    }
    return YAT_SDPA_DISPATCH(launch_fwd, dh, p, B, wide, (hipStream_t)stream);
}

int sdpa_bwd_impl(int B, int N, int T, int H, int dh, float scale, const void* q, int ldq, const void* k, const void* v,
                  int ldkv, const int* kv_off, int64_t kv_rows, const float* key_bias, const int* kv_len, const void* out,
                  int ldo, const void* dout, int lddo, const float* lse, float* delta, void* dq, int lddq, void* dk, void* dv,
                  int lddkv, const int* work_list, int n_work, int parts, yat_stream_t stream) {
    if (check_common(B, N, T, H, dh, ldq, ldkv, kv_rows) || (ldo & 7) || (lddo & 7) || (lddq & 3) || (lddkv & 3) || !q || !k ||
        !v || !out || !dout || !lse || !delta || !dq || !dk || !dv)
        return YAT_EINVAL;
    if (!key_bias && (kv_len || kv_off || work_list)) return YAT_EINVAL;
    if ((uint64_t)B * N * lddo * 2 > 0x7fffffffull) return YAT_EINVAL;
    if (kv_off && !kv_len) return YAT_EINVAL;
    SdpaP p{};
    p.N = N; p.T = T; p.H = H; p.dh = dh; p.scale = scale;
    p.q = (const bf16_t*)q; p.ldq = ldq; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.ldkv = ldkv;
    p.bias = key_bias; p.kv_len = kv_len; p.out = (bf16_t*)out; p.ldo = ldo; p.lse = (float*)lse; p.kv_off = kv_off;
    p.dout = (const bf16_t*)dout; p.lddo = lddo; p.delta = delta; p.dq = (bf16_t*)dq; p.lddq = lddq;
    p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.lddkv = lddkv;
    p.q_bytes = (uint64_t)B * N * ldq * 2; p.kv_bytes = (uint64_t)kv_rows * ldkv * 2; p.do_bytes = (uint64_t)B * N * lddo * 2;
    p.bias_bytes = (uint64_t)B * T * 4;
    p.stat_bytes = (uint64_t)B * H * N * 4;
    static const int xcd_env = YAT_TUNE_INT("YAT_SDPA_XCD", -1);
    p.xcd_remap = xcd_env >= 0 ? xcd_env : (T >= 1024);
    if (parts < 1 || parts > 3) return YAT_EINVAL;
    static const int wide_env = YAT_TUNE_INT("YAT_SDPA_WIDE", -1);
    if (parts & 1) {                               // dQ, and delta = rowsum(dO * O) which the dK/dV part reads
        // (three sub-tiles per wave with the key tile walked in halves fit -- 254 registers at dh 72 -- but measured only
        //  -2.4 % at N = T = 4096 and +5 % on the T = 300 cross-attention: not instantiated)
        int wide = wide_env >= 0 ? wide_env : ((int64_t)((N + 127) / 128) * H * B >= 1024);
        static const int dq3 = YAT_TUNE_INT("YAT_SDPA_DQ3", 1);      // dh 64, L = 4429: 1524 -> 1421 us (profiles/r03_f)
        if (dq3 && wide_env < 0 && dh <= 64 && (int64_t)((N + 191) / 192) * H * B >= 1024) wide = 2;
        const int rc = YAT_SDPA_DISPATCH(launch_dq, dh, p, B, wide, (hipStream_t)stream);
        if (rc != YAT_OK) return rc;
    }
    if (!(parts & 2)) return YAT_OK;
    if (work_list && n_work > 0) {
        p.work = work_list; p.n_work = n_work;
        // the compact list's units are single key tiles (equal work): an XCD-contiguous order is balanced whatever the lengths and
        // puts the key tiles of one (image, head) -- which read the same Q / dO -- on one L2: 67.3 -> 56.8 us on the ragged
        // SANA batch (round 6, profiles/r06_b_cross_attention_xcd_order.txt; the list path ran unmapped before)
        static const int xcd_work = YAT_TUNE_INT("YAT_SDPA_XCD_WORK", 1);
        p.xcd_remap = xcd_work;
    }
    // 128-key workgroups: dense grid only (the host's compact work list counts 64-key tiles)
    // dK/dV: 128-key workgroups of 4 waves x 32 keys (code 2) on the dense grid (the host's compact work list counts 64-key
    // tiles) while the head dim leaves the registers for two waves per SIMD (dh <= 80: 222 VGPRs once the query tile is
    // walked in two 32-row halves): every Q / dO fragment read from LDS feeds two MFMAs -- 1.83 -> 1.42 ms at N = T = 4096,
    // dh 72.  (Same shape with all four 16-query tiles in flight: 330 registers, one wave per SIMD, 3.19 vs 2.37 ms; code 1 =
    // 8 waves x 16 keys: 1.97 vs 1.80 ms.  Both stay reachable through YAT_SDPA_WIDE_KV.)
    static const int wide_kv_env = YAT_TUNE_INT("YAT_SDPA_WIDE_KV", -1);
    int wide_kv = (dh <= 80 && (int64_t)((T + 127) / 128) * H * B >= 1024) ? 2 : 0;
    if (wide_kv_env >= 0) wide_kv = wide_kv_env;
    if (p.work) wide_kv = 0;
    return YAT_SDPA_DISPATCH(launch_dkv, dh, p, B, wide_kv, (hipStream_t)stream);
}

}  // namespace

extern "C" {

int yat_sdpa_fwd(int B, int N, int T, int H, int dh, float scale, const void* q, int ldq, const void* k, const void* v,
                 int ldkv, const float* key_bias, const int* kv_len, void* out, int ldo, float* lse, yat_stream_t stream) {
    return sdpa_fwd_impl(B, N, T, H, dh, scale, q, ldq, k, v, ldkv, nullptr, (int64_t)B * T, key_bias, kv_len, out, ldo, lse,
                         stream);
}
int yat_sdpa_fwd_packed(int B, int N, int T, int H, int dh, float scale, const void* q, int ldq, const void* k, const void* v,
                        int ldkv, const int* kv_row_offsets, int kv_rows, const float* key_bias, const int* kv_len, void* out,
                        int ldo, float* lse, yat_stream_t stream) {
    if (!kv_row_offsets) return YAT_EINVAL;
    return sdpa_fwd_impl(B, N, T, H, dh, scale, q, ldq, k, v, ldkv, kv_row_offsets, kv_rows, key_bias, kv_len, out, ldo, lse,
                         stream);
}
int yat_sdpa_bwd(int B, int N, int T, int H, int dh, float scale, const void* q, int ldq, const void* k, const void* v,
                 int ldkv, const float* key_bias, const int* kv_len, const void* out, int ldo, const void* dout, int lddo,
                 const float* lse, float* delta, void* dq, int lddq, void* dk, void* dv, int lddkv, const int* work_list,
                 int n_work, int parts, yat_stream_t stream) {
    return sdpa_bwd_impl(B, N, T, H, dh, scale, q, ldq, k, v, ldkv, nullptr, (int64_t)B * T, key_bias, kv_len, out, ldo, dout,
                         lddo, lse, delta, dq, lddq, dk, dv, lddkv, work_list, n_work, parts, stream);
}
int yat_sdpa_bwd_packed(int B, int N, int T, int H, int dh, float scale, const void* q, int ldq, const void* k, const void* v,
                        int ldkv, const int* kv_row_offsets, int kv_rows, const float* key_bias, const int* kv_len,
                        const void* out, int ldo, const void* dout, int lddo, const float* lse, float* delta, void* dq, int lddq,
                        void* dk, void* dv, int lddkv, const int* work_list, int n_work, int parts, yat_stream_t stream) {
    if (!kv_row_offsets) return YAT_EINVAL;
    return sdpa_bwd_impl(B, N, T, H, dh, scale, q, ldq, k, v, ldkv, kv_row_offsets, kv_rows, key_bias, kv_len, out, ldo, dout,
                         lddo, lse, delta, dq, lddq, dk, dv, lddkv, work_list, n_work, parts, stream);
}

}  // extern "C"

#ifdef YAT_SDPA_STAMPS
extern "C" int yat_debug_sdpa_wg_times(unsigned int* host_dst) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(yat_sdpa_wg_times), sizeof(unsigned int) * 4096 * 6);
}
extern "C" int yat_debug_sdpa_stamps(unsigned int* host_dst) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(yat_sdpa_stamp_buf), sizeof(unsigned int) * 136);
}
#endif
