// bf16 MFMA GEMM family for gfx950 with fused epilogues.
//
//   C[M,N] = epilogue( A_op[M,K] * B_op[K,N] ),   fp32 accumulate, bf16 in/out
//
// Three operand layouts cover forward, dgrad and wgrad of every Linear / 1x1-conv in the SANA
// block (reference call sites: utils/patch_sana_attention_layers.py:94-113 via diffusers
// Attention/GLUMBConv; autograd backward at common/trainer.py:344):
//   NT  (a_t=0,b_t=0)  A[M,K] k-contiguous, B[N,K] k-contiguous   y  = x W^T
//   NN  (a_t=0,b_t=1)  A[M,K] k-contiguous, B[K,N] n-contiguous   dx = dy W
//   TN  (a_t=1,b_t=1)  A[K,M] m-contiguous, B[K,N] n-contiguous   dW = dy^T x
//
// Design (MI355X-first): 128x128x64 tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32
// accumulators.  Operand tiles are moved HBM->LDS by LDS-DMA (buffer_load ... lds, 16 B/lane,
// hardware range check supplies the zero padding of ragged M/N/K tails), double-buffered, one
// barrier per K-tile.  LDS images are lane-linear (DMA constraint) with the bank-conflict
// swizzle applied to the *source* address and undone on the read:
//   k-contiguous operand: [128 rows][64 k] 128-B rows, ds_read_b128, chunk ^= (row>>1)&7
//   k-strided operand   : [64 k][128 cols] 256-B rows, ds_read_b64_tr_b16 (hardware
//                         transpose), 32-B block ^= (k&3)|((k>>3)&1)<<2
// MFMA operands are swapped (D = B_frag x A_frag) so a lane owns 4 consecutive n of one row m:
// the epilogue loads/stores 8 B per lane.  Workgroups are remapped so each XCD (private L2)
// works on a contiguous band of tiles.
#include <math.h>
#include "common.hpp"
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include "gemm_common.hpp"
#include "../../include/yat_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM * BK + BN * BK) * 2;  // 32 KiB
constexpr int LDS_BYTES = 2 * STAGE_BYTES;            // 64 KiB -> 2 workgroups / CU


__device__ __forceinline__ uint32_t trswz(uint32_t krow) { return ((krow & 3) | (((krow >> 3) & 1) << 2)) << 1; }

// stage one operand tile (16 KiB) into LDS: 4 LDS-DMA instructions per wave
template <bool KSTRIDED>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds, int ld, int idx0, int idx_max,
                                           int k0, int K, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = j * 4 + wave;  // 1-KiB piece index 0..15
        uint32_t voff;
        if (!KSTRIDED) {
            const int r = piece * 8 + (lane >> 3);           // tile row (m or n index)
            const int c = swz128(r, lane & 7);               // source chunk for this LDS slot
            const int gi = idx0 + r, gk = k0 + c * 8;
            voff = (gi < idx_max && gk < K) ? (uint32_t)(((int64_t)gi * ld + gk) * 2) : YAT_OOB;
        } else {
            const int r = piece * 4 + (lane >> 4);           // tile k-row
            const int c = (lane & 15) ^ trswz(r);            // source chunk (8 cols)
            const int gk = k0 + r, gi = idx0 + c * 8;
            voff = (gk < K && gi < idx_max) ? (uint32_t)(((int64_t)gk * ld + gi) * 2) : YAT_OOB;
        }
        lds_dma16(rsrc, (YAT_LDS void*)(lds + piece * 1024), voff);
    }
}

// fragment for 16 consecutive output indices starting at idx0 (tile-local), k-substep kk (32 wide)
template <bool KSTRIDED>
__device__ __forceinline__ bf16x8 load_frag(const char* lds, int idx0, int kk, int lane) {
    if (!KSTRIDED) {
        const uint32_t r = idx0 + (lane & 15);
        const uint32_t c = swz128(r, kk * 4 + (lane >> 4));
        return lds_read8(lds, r * 128 + c * 16);
    } else {
        const uint32_t g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const uint32_t col = idx0 + 4 * p;
        const uint32_t r0 = kk * 32 + 8 * g + q, r1 = r0 + 4;
        const uint32_t c0 = (col >> 3) ^ trswz(r0), c1 = (col >> 3) ^ trswz(r1);
        bf16x4 lo = lds_read_tr4(lds, r0 * 256 + c0 * 16 + (p & 1) * 8);
        bf16x4 hi = lds_read_tr4(lds, r1 * 256 + c1 * 16 + (p & 1) * 8);
        return cat4(lo, hi);
    }
}

template <bool A_T, bool B_T, bool PRE = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware remap (bijective): blocks b and b+8 share an XCD -> give each XCD a contiguous band
    const int nwg = p.nbm * p.nbn;
    int id;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    // grouped ordering: 8 M-tiles x all N-tiles per group keeps A/B panels L2-resident
    const int GROUP = 8;
    const int per_group = GROUP * p.nbn;
    const int gid = id / per_group, first_m = gid * GROUP;
    const int gsz = min(p.nbm - first_m, GROUP);
    const int tm = first_m + (id % per_group) % gsz;
    const int tn = (id % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = (p.K + BK - 1) / BK;
    stage_tile<A_T>(ra, smem, p.lda, m0, p.M, 0, p.K, wave, lane);
    stage_tile<B_T>(rb, smem + BM * BK * 2, p.ldb, n0, p.N, 0, p.K, wave, lane);

    for (int t = 0; t < nt; ++t) {
        char* cur = smem + (t & 1) * STAGE_BYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // tile t landed for every wave; every wave is done reading the other buffer
        if (t + 1 < nt) {
            char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
            stage_tile<A_T>(ra, nxt, p.lda, m0, p.M, (t + 1) * BK, p.K, wave, lane);
            stage_tile<B_T>(rb, nxt + BM * BK * 2, p.ldb, n0, p.N, (t + 1) * BK, p.K, wave, lane);
        }
        const char* la = cur;
        const char* lb = cur + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = load_frag<A_T>(la, wm * 64 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = load_frag<B_T>(lb, wn * 64 + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(bfr[j], af[i], acc[i][j]);  // D[n][m]
        }
    }

    // ---- epilogue: lane owns row m = ..+(lane&15), cols n = ..+4*(lane>>4) + 0..3 ----
    const int rpb = p.rows_per_batch > 0 ? p.rows_per_batch : p.M;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= p.M) continue;
        const int b = m / rpb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
            if (n >= p.N) continue;
            gemm_epilogue_store<PRE>(p, acc[i][j], m, n, b);
        }
    }
}

}  // namespace

namespace {

// Tile-shape policy.  Modelled per-CU rates (TFLOP/s per CU, measured on MI355X): the 128x128 kernel
// runs 2 workgroups per CU, the 256-row kernels one.  Cost = rounds of workgroups x work per round.
double est_time_128(int M, int N, int K) {
    const double tiles = (double)((M + 127) / 128) * ((N + 127) / 128);
    const double rounds = ceil(tiles / 512.0);
    // per-round fixed cost: the kernel alone runs 32768 x 2240 x 320 in 85 us (9 rounds of 6.2 us of MFMA work) and
    // 32768 x 11200 x 320 in 401 us (44 rounds) -- ~3 us a round that the rate does not cover, and what made the policy
    // keep short-K products on this kernel that the 256 x 320 tile runs 7..16 % faster (profiles/r04_k_*)
    static const double fixed = YAT_TUNE_F64("YAT_GEMM_FIXED_128", 3e-6);
    return rounds * (2.0 * (2.0 * 128 * 128 * (double)K) / 3.4e12 + fixed);
}
// ``streams``: the caller's per-call hint (policy word of yat_gemm_bf16_ex): independent GEMM streams sharing the chip
double est_time_256(int M, int N, int K, int BNv, int ksplit, int streams) {
    const double tiles = (double)((M + 255) / 256) * ((N + BNv - 1) / BNv) * ksplit;
    // A GEMM of the training step seldom has the chip to itself (weight gradients on the side stream beside the dgrad chain, two
    // forward chains), and a whole-round model then picks wrongly: it splits K (slab traffic + a reduce kernel) or takes the
    // narrower tile to fill a last round that the neighbouring stream would have filled anyway.  The host says how many
    // independent GEMM streams it keeps in flight (per call, in the policy word of yat_gemm_bf16_ex).  One: whole rounds, the choice that is fastest
    // for the launch alone.  More: the launch is charged its CU-time (tiles / 256 rounds) but at least 0.375 of the chip, so K
    // is still split until ~96 workgroups exist (a 25-tile K = 32768 weight gradient left unsplit is cheap in CU-time and
    // 0.75 ms long: the side stream becomes the critical path).  Measured in the step on one box, every bench
    // (git history: scripts/gpu_policy_sweep.sh; today scripts/gpu_ab.sh): whole rounds -> this: SANA 88.3 -> 86.0 ms, PixArt 235 -> 232, LoKr B=32 354 -> 352;
    // floor 0.25 loses PixArt (260), no floor loses LoKr (431) and PixArt (315); planning for a 128-CU share with whole
    // rounds of THAT (the obvious model) keeps half the gain (87.4).  YAT_GEMM_ROUND_W / _FLOOR: the sweep's knobs.
    static const double round_w = YAT_TUNE_F64("YAT_GEMM_ROUND_W", -1.0);
    double rounds;
    if (round_w >= 0.0) {
        static const double round_floor = YAT_TUNE_F64("YAT_GEMM_ROUND_FLOOR", 0.375);
        rounds = round_w * ceil(tiles / 256.0) + (1.0 - round_w) * fmax(tiles / 256.0, round_floor);
    } else if (streams <= 1) {
        rounds = ceil(tiles / 256.0);                        // alone on the chip: whole rounds
    } else {
        rounds = fmax(tiles / 256.0, 0.375);                 // sharing it: CU-time, but spread over >= 96 CUs
    }
    double t = rounds * ((2.0 * 256 * BNv * (double)K / ksplit) / 5.0e12 + 8e-6);   // + per-workgroup fixed cost
    if (ksplit > 1) t += (ksplit + 0.5) * (double)M * N * 4.0 / 4.0e12 + 3e-6;      // slab write + reduce pass
    return t;
}

}  // namespace

// first published layout of yat_gemm_epilogue: struct_size .. rows_per_batch
static constexpr uint32_t YAT_GEMM_EPILOGUE_V1_BYTES = offsetof(yat_gemm_epilogue, glu_u);
static_assert(YAT_GEMM_EPILOGUE_V1_BYTES == 64 && sizeof(yat_gemm_epilogue) % 8 == 0, "yat_gemm_epilogue layout");

// argument checks + descriptor shared by the single and the grouped entry points
static int fill_gemm_p(int a_t, int b_t, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                       int ldc, const yat_gemm_epilogue* ep, GemmP& p) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return YAT_EINVAL;
    if ((N & 3) || (lda & 7) || (ldb & 7) || (ldc & 3)) return YAT_EINVAL;
    if (!a_t && (K & 7)) return YAT_EINVAL;            // 16-B chunks along k
    if (a_t && (M & 7)) return YAT_EINVAL;             // 16-B chunks along m
    if (b_t && (N & 7)) return YAT_EINVAL;
    if (!b_t && (K & 7)) return YAT_EINVAL;
    p = GemmP{};
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = (bf16_t*)C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    yat_gemm_epilogue e_{};
    if (ep) {
        // versioned struct: copy exactly what the caller owns, zero-fill options it does not know (never read past it)
        const uint32_t sz = ep->struct_size;
        if (sz < YAT_GEMM_EPILOGUE_V1_BYTES || sz > sizeof(e_) || (sz & 7)) return YAT_EINVAL;
        memcpy(&e_, ep, sz);
        ep = &e_;
        p.bias = (const bf16_t*)ep->bias; p.gate = (const bf16_t*)ep->gate; p.res = (const bf16_t*)ep->residual;
        p.aux = (bf16_t*)ep->aux_out; p.ldr = ep->ld_residual ? ep->ld_residual : ldc;
        p.ldaux = ep->ld_aux ? ep->ld_aux : ldc; p.gate_ld = ep->ld_gate; p.rows_per_batch = ep->rows_per_batch;
        p.act = ep->activation;
        if (p.act < 0 || p.act > 2) return YAT_EINVAL;
        if (p.gate && (p.gate_ld & 3)) return YAT_EINVAL;
        p.pre_add = (const bf16_t*)ep->pre_add; p.ld_pre = ep->ld_pre_add;
        if (p.pre_add && ((p.ld_pre & 3) || p.ld_pre < N)) return YAT_EINVAL;
        p.glu_u = (const bf16_t*)ep->glu_u; p.ld_glu = ep->ld_glu_u;
        if (p.glu_u && (p.bias || p.gate || p.res || p.aux || p.act || (p.ld_glu & 3) || p.ld_glu < 2 * N || ldc < 2 * N))
            return YAT_EINVAL;
        p.dact_z = (const bf16_t*)ep->dact_z; p.ld_z = ep->ld_dact_z;
        if (p.dact_z && (p.bias || p.gate || p.res || p.aux || p.glu_u || p.pre_add || !p.act || (p.ld_z & 3) || p.ld_z < N))
            return YAT_EINVAL;
        p.rowsum = (bf16_t*)ep->a_rowsum_out; p.rowsum_acc = ep->a_rowsum_accumulate;
        if (p.rowsum && (!a_t || !b_t || p.bias || p.gate || p.aux || p.act || p.glu_u || p.pre_add || p.dact_z))
            return YAT_EINVAL;                                   // weight-gradient layout, plain (or accumulating) epilogue
        p.A2 = (const bf16_t*)ep->a2; p.B2 = (const bf16_t*)ep->b2; p.K2 = ep->k2; p.a2_group = ep->a2_group_n;
        if (p.A2 || p.B2) {                                      // second operand pair: forward layout, whole tiles of K2
            if (!p.A2 || !p.B2 || a_t || b_t || p.K2 <= 0 || (p.K2 & 63) || p.glu_u || p.pre_add || p.dact_z || p.rowsum)
                return YAT_EINVAL;
            if (p.a2_group < 0 || (p.a2_group > 0 && (p.a2_group % 320) && (p.a2_group % 256))) return YAT_EINVAL;
            const int blocks = p.a2_group > 0 ? (N + p.a2_group - 1) / p.a2_group : 1;
            if ((int64_t)blocks * p.K2 > lda || p.K2 > ldb) return YAT_EINVAL;     // the rows of A2 / B2 have A's / B's stride
            p.a2_bytes = ((uint64_t)(M - 1) * lda + (uint64_t)blocks * p.K2) * 2;
            p.b2_bytes = ((uint64_t)(N - 1) * ldb + (uint64_t)p.K2) * 2;
            if (p.a2_bytes > 0x7fffffffull || p.b2_bytes > 0x7fffffffull) return YAT_EINVAL;
        }
    }
    p.a_bytes = (uint64_t)(a_t ? K : M) * lda * 2;
    p.b_bytes = (uint64_t)(b_t ? K : N) * ldb * 2;
    if (p.a_bytes > 0x7fffffffull || p.b_bytes > 0x7fffffffull) return YAT_EINVAL;
    p.ksplit = 1;
    return YAT_OK;
}

extern "C" uint64_t yat_gemm_epilogue_size(void) { return sizeof(yat_gemm_epilogue); }

int yat_gemm256_grouped_launch(int a_t, int b_t, int ngroups, const GemmP* probs, hipStream_t stream);

extern "C" int yat_gemm_grouped_bf16(int a_t, int b_t, int count, const yat_gemm_problem* problems, yat_stream_t stream) {
    if (count < 1 || count > 8 || !problems) return YAT_EINVAL;
    GemmP ps[8];
    for (int i = 0; i < count; ++i) {
        const yat_gemm_problem& q = problems[i];
        const int rc = fill_gemm_p(a_t, b_t, q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.epilogue, ps[i]);
        if (rc) return rc;
        if (ps[i].rowsum && !(a_t && b_t)) return YAT_EINVAL;
    }
    return yat_gemm256_grouped_launch(a_t, b_t, count, ps, (hipStream_t)stream);
}

extern "C" int yat_gemm_bf16_ex(int a_t, int b_t, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                                void* C, int ldc, const yat_gemm_epilogue* ep, int variant, void* workspace,
                                uint64_t workspace_bytes, yat_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    // policy word: tile variant + 100 * forced K split (tests / tuning) + 10000 * independent GEMM streams the caller keeps
    // in flight (0 / 1: this launch has the chip to itself) -- per call, so the library holds no policy state
    int ksplit = 1, streams = 1, group = 0;
    if (variant < 0) return YAT_EINVAL;
    if (variant >= 1000000) { group = variant / 1000000; variant %= 1000000; }      // tile-order row group (tuning; 0 = policy)
    if (group > 64) return YAT_EINVAL;
    if (variant >= 10000) { streams = variant / 10000; variant %= 10000; }
    if (streams > 8) return YAT_EINVAL;
    {
        static const int forced = YAT_TUNE_INT("YAT_GEMM_CONCURRENCY", 0);
        if (forced >= 1 && forced <= 8) streams = forced;
    }
    if (variant >= 100) { ksplit = variant / 100; variant %= 100; }
    if (variant != 0 && variant != 1 && variant != 4 && variant != 5) return YAT_EINVAL;
    if (ksplit < 1 || ksplit > 32) return YAT_EINVAL;
    GemmP p;
    {
        const int rc = fill_gemm_p(a_t, b_t, M, N, K, A, lda, B, ldb, C, ldc, ep, p);
        if (rc) return rc;
    }
    const bool wide_ok = !(N & 7) && !(ldc & 7) && !(p.res && (p.ldr & 7)) && !(p.aux && (p.ldaux & 7)) &&
                         !(p.gate && (p.gate_ld & 7)) && !(p.glu_u && (p.ld_glu & 7)) && !(p.pre_add && (p.ld_pre & 7));
    if (p.pre_add && (a_t || b_t || p.glu_u)) return YAT_EINVAL;       // adapter addend: forward layout only
    if (p.dact_z) {                             // activation-backward epilogue: 256-row kernel, dgrad layout only
        if (ksplit != 1 || variant == 1 || a_t || !b_t) return YAT_EINVAL;
        if (variant == 0) variant = est_time_256(M, N, K, 320, 1, streams) < est_time_256(M, N, K, 256, 1, streams) ? 5 : 4;
    }
    if (p.glu_u) {                              // GLU-backward epilogue lives in the 256-row kernel only
        if (ksplit != 1 || variant == 1 || a_t || !b_t || !wide_ok) return YAT_EINVAL;
        if (variant == 0) variant = est_time_256(M, N, K, 320, 1, streams) < est_time_256(M, N, K, 256, 1, streams) ? 5 : 4;
    }
    if (p.A2) {                                 // second operand pair: 256-row kernel only, the K loop runs over K + K2
        if (ksplit != 1 || variant == 1 || !wide_ok) return YAT_EINVAL;
        const bool ok5 = p.a2_group == 0 || p.a2_group % 320 == 0, ok4 = p.a2_group == 0 || p.a2_group % 256 == 0;
        if ((variant == 5 && !ok5) || (variant == 4 && !ok4)) return YAT_EINVAL;
        if (variant == 0)
            variant = !ok4 ? 5 : !ok5 ? 4
                    : est_time_256(M, N, K + p.K2, 320, 1, streams) < est_time_256(M, N, K + p.K2, 256, 1, streams) ? 5 : 4;
    }
    if (p.rowsum) {                             // bias gradient fused into the weight gradient: 256 x 256 tile (the 320-wide
        if (ksplit != 1 || (variant != 0 && variant != 4)) return YAT_EINVAL;   // one has no registers for the extra
        variant = 4;                            // accumulators), whole K in one workgroup
    }
    static const int max_ksplit = YAT_TUNE_INT("YAT_GEMM_MAX_KSPLIT", 32);
    static const bool pow2_only = YAT_TUNE_INT("YAT_GEMM_KSPLIT_POW2", 0) != 0;
    if (variant == 0) {
        variant = 1;
        // (N >= 256: one 256/320-wide column tile is fine when K is long enough to split -- the [out, in_m*r] weight
        //  gradients of the factored LoKr path are 2240 x 320 x 32768: 9 tiles, split 28 ways)
        static const int min_n256 = YAT_TUNE_INT("YAT_GEMM_MIN_N256", 256);
        const bool big = M >= 1024 && (N >= 512 || (N >= min_n256 && K >= 2048) || (N >= 256 && K >= 8192)) && K >= 256;
        // Skinny outputs with a long reduction -- the embedders' M = B rows (time_embed.linear dgrad: 8 x 2240 x 13440), the
        // patch-embedding / output-head weight gradients (32 channels x 2240 x 8192 tokens): on the 128 x 128 kernel they are
        // 18 workgroups walking 128..210 k-tiles each (150..250 us with the chip idle, at the head of the forward and the tail
        // of the backward); split along K on the 256-row kernel (mostly zero-filled tile, but the launch is bound by reading
        // the weight) they are ~7 tiles x 16..26 slices.  Only with a split: unsplit, the small kernel is the better one.
        static const bool skinny_on = YAT_TUNE_INT("YAT_GEMM_SKINNY", 1) != 0;
        const bool skinny = skinny_on && !big && K >= 2048 && M >= 8 && N >= 32;
        if (big || skinny) {
            double best = est_time_128(M, N, K);
            for (int v = 4; v <= 5; ++v)
                for (int s = skinny ? 2 : 1; s <= max_ksplit; s = pow2_only ? s * 2 : s + 1) {
                    // any factor, not only powers of two: what matters is tiles x s against the 256 CUs (75 tiles x 3 = 225
                    // fills one round; x 2 leaves 106 CUs idle, x 4 spills into a second round)
                    if (s > 1 && p.pre_add) continue;
                    if (s > 1 && (!workspace || !wide_ok || (uint64_t)s * M * N * 4 > workspace_bytes || K / s < 512)) continue;
                    const double t = est_time_256(M, N, K, v == 4 ? 256 : 320, s, streams);
                    if (t < best) { best = t; variant = v; ksplit = s; }
                }
        }
    }
    if (variant == 4 || variant == 5) {
        if (ksplit > 1 && (!workspace || !wide_ok || (uint64_t)ksplit * M * N * 4 > workspace_bytes)) return YAT_EINVAL;
        p.ksplit = ksplit;
        p.partial = (float*)workspace;
        // Tile order: 4 rows of 256-row tiles share their B panels inside one XCD's contiguous run.  Swept per shape
        // (scripts/gemm_group_sweep.py, profiles/r03_d_gemm_group_sweep.txt): the L2-miss traffic moves by up to 2.7 x with the
        // group size (profiles/r03_d_gemm_traffic_vs_tile_order.txt) and the time by < 1 % -- the re-reads are served by the
        // Infinity Cache -- except the long-K input gradient (8192 x 2240 x 11200: 7 MB B panels), 3 - 6 % faster with 8.
        p.group = group ? group : ((!a_t && b_t && K >= 8192) ? 8 : 0);
        const int rc = yat_gemm256_launch(a_t, b_t, variant, p, stream);
        if (rc || ksplit == 1) return rc;
        return yat_gemm_splitk_reduce(p, stream);
    }
    if (ksplit != 1) return YAT_EINVAL;
    p.nbm = (M + BM - 1) / BM; p.nbn = (N + BN - 1) / BN;
    static bool attr_set = false;   // idempotent one-time launch attribute (64 KiB dynamic LDS)
    if (!attr_set) {
        const void* kernels[] = {(const void*)gemm_bf16_kernel<false, false>, (const void*)gemm_bf16_kernel<false, true>,
                                 (const void*)gemm_bf16_kernel<true, true>, (const void*)gemm_bf16_kernel<true, false>,
                                 (const void*)gemm_bf16_kernel<false, false, true>};
        for (const void* k : kernels) {
            const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            if (e != hipSuccess) return (int)e;          // a HIP error (e.g. no device), not an argument error
        }
        attr_set = true;
    }
    dim3 grid(p.nbm * p.nbn), block(256);
    if (p.pre_add) hipLaunchKernelGGL((gemm_bf16_kernel<false, false, true>), grid, block, LDS_BYTES, stream, p);
    else if (!a_t && !b_t) hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, LDS_BYTES, stream, p);
    else if (!a_t && b_t) hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, LDS_BYTES, stream, p);
    else if (a_t && b_t) hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, block, LDS_BYTES, stream, p);
    YAT_CHECK_LAUNCH();
    return YAT_OK;
}

extern "C" int yat_gemm_bf16(int a_t, int b_t, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                             void* C, int ldc, const yat_gemm_epilogue* ep, yat_stream_t stream) {
    return yat_gemm_bf16_ex(a_t, b_t, M, N, K, A, lda, B, ldb, C, ldc, ep, 0, nullptr, 0, stream);
}
