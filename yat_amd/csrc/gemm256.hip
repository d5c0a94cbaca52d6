// gemm256: 256 x (256|320) x 64 tile bf16 MFMA GEMM for gfx950 with two staggered wave groups.
//
// Why a second kernel: the 128x128 kernel (gemm.hip) needs ~150 GB/s of L2->LDS traffic per CU at
// full MFMA rate (38 TB/s chip-wide, above the ~34 TB/s aggregate L2 bandwidth); a 256-wide tile
// halves the bytes per flop.  A 256x256 tile fills the LDS of a CU (2 x 64..72 KiB), so there is
// one workgroup per CU and nothing from a neighbouring workgroup to hide its barriers; the overlap
// has to come from inside the workgroup:
//
//   8 waves = 2 groups (g = wave>>2, rows [128g,128g+128)) x 4 column slices.  Waves w and w+4 land
//   on the same SIMD.  Every K-tile is split into two 32-deep sub-steps, each a LOAD segment (12+
//   ds_reads of the fragments, LDS-DMA issue for the next tile) followed by a COMPUTE segment (32..40
//   MFMA 16x16x32 on registers only), separated by raw s_barriers.  Group 1 executes one extra
//   barrier before its loop (group 0 one after), so on each SIMD one wave computes while the other
//   loads: the matrix pipe sees back-to-back MFMA segments.
//
// LDS-DMA protocol (the only ordering for a DMA'd tile is the issuing wave's vmcnt + a barrier the
// reader has passed).  Barrier bookkeeping: the n-th barrier of every wave is one rendezvous; group 0's
// iteration t spans rendezvous 4t+1..4t+4, group 1's 4t+2..4t+5; tile u (stage u&1) is first read after
// rendezvous 4u and last read before 4u+4.  DMA pieces are issued INSIDE compute segments, between
// groups of four MFMAs (measured: with the issue in a load segment that segment outlasts the partner's
// compute segment and the matrix pipe idles ~30 %): group 0 issues tile t+1 in COMPUTE(t,ks0) and waits
// before its 4th barrier of iteration t; group 1 issues tile t+2 in COMPUTE(t,ks1) and waits for tile
// t+1 before its 3rd barrier of iteration t -- both waits precede rendezvous 4t+4.
//
// DEEP schedule (round 4: layouts with a k-strided B operand, the input- and weight-gradient GEMMs; round 5: every layout -- a
// k-contiguous B operand gets a half-major LDS image of the same piece shape, see B_KH below).  The two-stage ring above
// gives a piece 2..4 of the 4 segments of a K-tile to land: enough for the Infinity Cache, short for HBM under load (operands
// from HBM cost the same launches +10..+25 % at the sustained clock, scripts/gemm_sustained_probe.py).  The LDS images are
// unchanged, but a k-strided image is refilled per 32-deep HALF (k-rows 0..31 / 32..63 are contiguous: a four-slot ring of
// halves in the same two stages) as soon as group 1 has read it, and the rows of a k-contiguous A image -- private to one
// wave group -- are refilled by that group right after its own last read:
//   u = 2 t + kk numbers the 32-deep sub-steps; group 0's LOAD(u) ends with rendezvous 2u+1, its COMPUTE(u) with 2u+2;
//   group 1's with 2u+2 / 2u+3.  Half u is last read before rendezvous 2u+2 and half v first read after rendezvous 2v.
//   group 0 issues its share of half u+3 in COMPUTE(u) (after 2u+1 > 2(u-1)+2) and waits for it at the end of COMPUTE(u+2)
//           (before 2u+6) with two younger batches still in flight:  s_waitcnt vmcnt(n(u+1) + n(u+2));
//   group 1 issues its share of half u+4 in COMPUTE(u) (after 2u+2) and waits at the end of LOAD(u+3) (before 2u+8), again
//           past two younger batches;
//   k-contiguous A (input gradient): group g issues its own 128 rows of tile t+2 in COMPUTE(t, ks1), first in the batch, and
//           waits before the rendezvous that precedes its LOAD(t+2, ks0).
// A piece now has 4..6 segments (1.0..1.5 K-tiles) to land instead of 2..4.  Same pieces, same LDS bytes, same fragment
// reads, same MFMA order: results are bit-identical to the two-stage schedule (-DYAT_GEMM_DEEP=0).
//
// Operand layouts, swizzles, swapped-operand MFMA and epilogue are those of gemm.hip.
#include "common.hpp"
#include "gemm_common.hpp"
#include <type_traits>

namespace {

constexpr int BM = 256, BK = 64;

template <int NT> struct Geo {
    static constexpr int BN = 64 * NT;                 // 4 column slices x NT tiles of 16
    static constexpr int A_BYTES = BM * BK * 2;        // 32 KiB
    static constexpr int B_BYTES = BN * BK * 2;        // 32 / 40 KiB
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int LDS = 2 * STAGE;              // 128 / 144 KiB
    static constexpr int PA = A_BYTES / 1024 / 8;      // DMA pieces per wave per tile (A): 4
    static constexpr int PB = B_BYTES / 1024 / 8;      // (B): 4 / 5
};

// Swizzle of the k-strided image (XOR on the 16-B chunk index of k-row `krow`; bit 0 untouched so a 32-B block of
// ds_read_b64_tr_b16 stays together).  The 8 rows one 32-lane half reads ({0..3, 8..11} + 4n) must land on 8 distinct
// 32-B blocks of the 256-B bank row:
//  * row stride = 0 mod 256 B (256 columns): every row starts on the same bank -> XOR 3 bits (closed over 16 chunks);
//  * row stride = 128 mod 256 B (320 columns, 40 chunks per row): odd rows are already offset by 128 B, so XOR only
//    2 bits (closed over groups of 8 chunks -- a 4-bit XOR would leave the 40-chunk row).
template <int COLS>
__device__ __forceinline__ uint32_t trswz2(uint32_t krow) {
    if ((COLS * 2) % 256 == 0) return ((krow & 3) | (((krow >> 3) & 1) << 2)) << 1;
    return (((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 1;
}

// loop-invariant part of one DMA piece: byte offset of this lane's 16-B chunk at k-tile 0 (or OOB).  The source chunk index
// inside the 64-wide k-tile that the ragged-K check of a row-mode piece needs is the same for every piece of a wave
// (pieces are 8 rows apart per wave step of 8 or 4, and the swizzle looks at (row >> 1) & 7): piece_kchunk().
struct Piece { uint32_t voff; };
__device__ __forceinline__ uint32_t piece_kchunk(int wave, int lane) { return swz128(wave * 8 + (lane >> 3), lane & 7); }

template <bool KSTRIDED, int ROWS_OR_COLS>
__device__ __forceinline__ Piece make_piece(int piece, int lane, int ld, int idx0, int idx_max) {
    Piece pc;
    if (!KSTRIDED) {
        const int r = piece * 8 + (lane >> 3);
        const int c = swz128(r, lane & 7);
        const int gi = idx0 + r;
        pc.voff = gi < idx_max ? (uint32_t)(((int64_t)gi * ld + c * 8) * 2) : YAT_OOB;
    } else {
        constexpr int ROWB = ROWS_OR_COLS * 2;                 // bytes per k-row of the LDS image
        const uint32_t o = piece * 1024 + lane * 16;
        const uint32_t r = o / ROWB, slot = (o % ROWB) >> 4;
        const uint32_t c = slot ^ trswz2<ROWS_OR_COLS>(r);
        const int gi = idx0 + c * 8;
        pc.voff = gi < idx_max ? (uint32_t)(((int64_t)r * ld + gi) * 2) : YAT_OOB;   // k-rows past K: buffer range check
    }
    return pc;
}

template <bool KSTRIDED, int COLS>
__device__ __forceinline__ bf16x8 frag256(const char* lds, int idx0, int kk, int lane) {
    if (!KSTRIDED) {
        const uint32_t r = idx0 + (lane & 15);
        const uint32_t c = swz128(r, kk * 4 + (lane >> 4));
        return lds_read8(lds, r * 128 + c * 16);
    } else {
        constexpr uint32_t ROWB = COLS * 2;
        const uint32_t g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const uint32_t col = idx0 + 4 * p;
        const uint32_t r0 = kk * 32 + 8 * g + q, r1 = r0 + 4;
        const uint32_t c0 = (col >> 3) ^ trswz2<COLS>(r0), c1 = (col >> 3) ^ trswz2<COLS>(r1);
        return cat4(lds_read_tr4(lds, r0 * ROWB + c0 * 16 + (p & 1) * 8), lds_read_tr4(lds, r1 * ROWB + c1 * 16 + (p & 1) * 8));
    }
}

// The k-strided fragment addresses, arranged so a LOAD segment computes almost nothing.  A lane reads rows r0 = 32 kk + 8 g + q
// and r0 + 4; trswz2 is the same for r0, r0 + 4 and r0 + 32 (it looks at bits 0, 1 and 3 of the row only), so ONE address per
// 16-column fragment serves both rows and both sub-steps through the instruction's immediate offset.
//  * 256-column image (operand A always; B of the 256 x 256 tile): the chunk index of fragment f is (first/8 + 2 f + p/2) ^ swz
//    with `first` a multiple of 64 columns, so bits 1..3 of first/8 are free and the XOR lands on 2 f alone:
//    address(f) = address(0) ^ (f << 5) -- one per-lane base, one v_xor per fragment.
//  * 320-column image (B of the 256 x 320 tile): first = 80 wc is not aligned like that; one per-lane address per fragment.
template <int COLS>
__device__ __forceinline__ uint32_t tr_lane_addr(uint32_t col0, int lane) {
    const uint32_t g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const uint32_t r0 = 8 * g + q;
    const uint32_t c = ((col0 + 4 * p) >> 3) ^ trswz2<COLS>(r0);
    return r0 * (COLS * 2) + c * 16 + (p & 1) * 8;
}
// The transposing read is issued as inline assembly on purpose.  Through the builtin the compiler cannot tell that the read
// does not alias a pending LDS-DMA write and puts an `s_waitcnt vmcnt(0)` in front of the first transposing read of EVERY load
// segment: the wave then waits for the tile it has only just started fetching -- the prefetch distance of the whole design
// collapses to zero for the k-strided layouts (stamps: that load segment 840..900 ticks instead of 410..490).  The ordering
// a DMA'd tile needs is in the protocol already (the issuing wave's own vmcnt wait + the rendezvous before the first read);
// the reads' own completion is the explicit `s_waitcnt lgkmcnt(0)` that ends every load segment, before any MFMA uses them.
template <int OFF>
__device__ __forceinline__ bf16x4 lds_read_tr4_asm(uint32_t lds_addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "i"(OFF));
    return v;
}
template <int COLS>
__device__ __forceinline__ bf16x8 frag_tr(uint32_t lds_addr, int kk) {
    constexpr int R4 = 4 * COLS * 2, K1 = 32 * COLS * 2;
#ifdef YAT_GEMM_TR_BUILTIN          // A/B only: the builtin form with the compiler's conservative vmcnt(0)
    YAT_LDS const char* a = (YAT_LDS const char*)(uintptr_t)lds_addr;
    return cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((YAT_LDS bf16x4*)(a + kk * K1)),
                __builtin_amdgcn_ds_read_tr16_b64_v4bf16((YAT_LDS bf16x4*)(a + kk * K1 + R4)));
#endif
    if (kk == 0) return cat4(lds_read_tr4_asm<0>(lds_addr), lds_read_tr4_asm<R4>(lds_addr));
    return cat4(lds_read_tr4_asm<K1>(lds_addr), lds_read_tr4_asm<K1 + R4>(lds_addr));
}

#define YAT_PHASE_BARRIER()                  \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)

// Diagnostic builds only (scripts/build_variant.py ... -DYAT_ABL_*; results are then WRONG, only the time is read): where
// does the main loop's time go -- waiting for the LDS-DMA to land, issuing it, or the four rendezvous per K-tile?
#ifdef YAT_ABL_NO_BARRIER
#define YAT_LOOP_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define YAT_LOOP_BARRIER() YAT_PHASE_BARRIER()
#endif
#ifdef YAT_ABL_NO_LDSREAD
#define YAT_ABL_SKIP_READS 1
#else
#define YAT_ABL_SKIP_READS 0
#endif

// Diagnostic build -DYAT_GEMM_STAMPS (scripts/gemm_stamps.py): every wave of workgroup 0 sums, per K-loop iteration slot, the
// s_memtime ticks it spent in  [4 kk + 0] LOAD (fragment reads issued and landed, incl. its DMA wait)  [+1] the rendezvous
// after it  [+2] COMPUTE (MFMA + DMA issue)  [+3] the rendezvous after it (incl. group 0's DMA wait).  The sums go to a
// buffer of their own that nothing else reads; the product build contains none of this.
#ifdef YAT_GEMM_STAMPS
__device__ unsigned int yat_gemm_stamp_buf[8 * 8 + 8];
#define YAT_STAMP(slot)                                                       \
    do {                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                    \
        const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime();         \
        st_sum[slot] += now_ - st_prev;                                       \
        st_prev = now_;                                                       \
        __builtin_amdgcn_sched_barrier(0);                                    \
    } while (0)
#else
#define YAT_STAMP(slot) do {} while (0)
#endif

// workgroups are dealt round-robin to the 8 XCDs: give every XCD one contiguous run of `nwg` work items
__device__ __forceinline__ int xcd_contiguous(int nwg) {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// One workgroup's whole job: `id` = work item inside problem p (tile x K slice, before the row-group swizzle).
template <bool A_T, bool B_T, int NT, int EPI = 0>      // EPI: 0 standard, 1 GLU backward, 2 standard + pre_add, 3 act backward,
                                                         //      4 standard + row sums of A (wgrad + bias gradient),
                                                         //      5 standard, second operand pair behind K (forward layout)
__device__ __forceinline__ void gemm256_body(const GemmP p, int id) {
    using G = Geo<NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef YAT_GEMM_STAMPS
    const uint32_t st_kernel_begin = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, wc = wave & 3;

    const int ksl = id % p.ksplit;             // K slice of this workgroup (slices of one tile are neighbours -> same XCD)
    id /= p.ksplit;
#ifndef YAT_GEMM_GROUP
#define YAT_GEMM_GROUP 4
#endif
    const int GROUP = p.group > 0 ? p.group : YAT_GEMM_GROUP;
    const int per_group = GROUP * p.nbn;
    const int gid = id / per_group, first_m = gid * GROUP;
    const int gsz = min(p.nbm - first_m, GROUP);
    const int tm = first_m + (id % per_group) % gsz;
    const int tn = (id % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * G::BN;

    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);
    // EPI 5, the second operand pair (C = epilogue(A B^T + A2 B2^T): a PEFT adapter's factored term T P^T folded into the base
    // Linear): A2 / B2 have the row strides of A / B, so the K2 / 64 k-tiles behind the last tile of K are the same per-lane
    // offsets against two other buffer resources -- nothing else in the loop knows.  Column tiles of one A2 block: a2_group.
    const int a2_col = (EPI == 5 && p.a2_group > 0) ? (n0 / p.a2_group) * p.K2 : 0;
    const __amdgpu_buffer_rsrc_t ra2 = make_rsrc(EPI == 5 ? p.A2 + a2_col : p.A, EPI == 5 ? p.a2_bytes - (uint64_t)a2_col * 2 : 0);
    const __amdgpu_buffer_rsrc_t rb2 = make_rsrc(EPI == 5 ? p.B2 : p.B, EPI == 5 ? p.b2_bytes : 0);

    // this wave's DMA pieces (loop invariant): A pieces wave + 8j, B pieces wave + 8j
    Piece pa[G::PA], pb[G::PB];
#pragma unroll
    for (int j = 0; j < G::PA; ++j) pa[j] = make_piece<A_T, BM>(wave + 8 * j, lane, p.lda, m0, p.M);
#pragma unroll
    for (int j = 0; j < G::PB; ++j) pb[j] = make_piece<B_T, G::BN>(wave + 8 * j, lane, p.ldb, n0, p.N);
    const uint32_t kchunk = piece_kchunk(wave, lane);
    const uint32_t a_kstep = A_T ? (uint32_t)(BK * p.lda * 2) : (uint32_t)(BK * 2);
    const uint32_t b_kstep = B_T ? (uint32_t)(BK * p.ldb * 2) : (uint32_t)(BK * 2);
    const int nt_main = (p.K + BK - 1) / BK;
    const int nt_all = nt_main + (EPI == 5 ? p.K2 / BK : 0);
    const int kt0 = (int)(((int64_t)nt_all * ksl) / p.ksplit);                 // this slice: k-tiles [kt0, kt0 + nt)
    const int nt = (int)(((int64_t)nt_all * (ksl + 1)) / p.ksplit) - kt0;
    const bool ragged = (p.K & (BK - 1)) != 0;

    // A DMA piece costs the issuing wave its instruction AND whatever computes its operands.  The k-tile advance therefore
    // rides in the instruction's scalar offset (one s_mul per segment, no VALU per piece) and the per-lane offsets stay the
    // loop-invariant registers of make_piece.  The scalar offset is not part of the buffer range check, so the two cases that
    // lean on per-lane checks -- the ragged last k-tile (row-mode chunks past K, k-strided rows past K) -- take the CHECKED
    // form (per-lane add + compare) instead; which form a segment uses is one uniform branch per segment, not per piece.
    // (K rotation -- workgroups that share a B panel starting their loops at different depths, so that only one of them at
    // a time waits for a line's HBM miss -- was built and measured in round 5: the cold / hot gap of the forward GEMMs fell
    // from 9 .. 19 % to 3 .. 13 %, but giving up the lockstep L2 sharing cost more than that: hot + 10 %, step 76.5 -> 80.4 ms.
    // profiles/r05_b_gemm_k_rotation_rejected_*.)
    auto ktile = [&](int tl) { return kt0 + tl; };                              // iteration tl of this slice -> global K-tile
    auto is_tail = [&](int tl) { return ragged && ktile(tl) == nt_main - 1; };
    // leading k-tiles of this slice that are all full ones: up to the ragged tile of K if the slice holds it (it is the slice's
    // last tile unless a second operand pair follows it)
    const int tail_at = ragged ? nt_main - 1 - kt0 : nt;
    const int nfull = (tail_at >= 0 && tail_at < nt) ? tail_at : nt;
    auto piece = [&](auto checked, int tl, char* stage, int j) {
        constexpr bool CHECKED = decltype(checked)::value;
        const int t = ktile(tl);
        const bool opa = j < G::PA;
        const int jj = opa ? j : j - G::PA;
        if (opa) {
            YAT_LDS void* dst = (YAT_LDS void*)(stage + (wave + 8 * jj) * 1024);
            if (CHECKED) {
                const uint32_t kvalid = (uint32_t)(p.K - t * BK) >> 3;     // valid 16-B chunks in this k-tile (row-mode)
                uint32_t v = pa[jj].voff + (uint32_t)t * a_kstep;
                if (!A_T && kchunk >= kvalid) v = YAT_OOB;
                lds_dma16(ra, dst, v);
            } else {
                lds_dma16s(ra, dst, pa[jj].voff, (uint32_t)t * a_kstep);
            }
        } else {
            YAT_LDS void* dst = (YAT_LDS void*)(stage + G::A_BYTES + (wave + 8 * jj) * 1024);
            if (CHECKED) {
                const uint32_t kvalid = (uint32_t)(p.K - t * BK) >> 3;
                uint32_t v = pb[jj].voff + (uint32_t)t * b_kstep;
                if (!B_T && kchunk >= kvalid) v = YAT_OOB;
                lds_dma16(rb, dst, v);
            } else {
                lds_dma16s(rb, dst, pb[jj].voff, (uint32_t)t * b_kstep);
            }
        }
    };
    constexpr int NPIECE = G::PA + G::PB;        // 8 or 9 pieces per wave per tile
    auto issue_range = [&](int tl, char* stage, int j0, int j1) {       // pieces [j0, j1) of tile tl, form chosen once
        if (is_tail(tl)) {
#pragma unroll
            for (int j = 0; j < NPIECE; ++j)
                if (j >= j0 && j < j1) piece(std::true_type{}, tl, stage, j);
        } else {
#pragma unroll
            for (int j = 0; j < NPIECE; ++j)
                if (j >= j0 && j < j1) piece(std::false_type{}, tl, stage, j);
        }
    };
    auto issue = [&](int tl, char* stage) { issue_range(tl, stage, 0, NPIECE); };

    f32x4 acc[8][NT];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 af[8], bfr[NT];
    // EPI 4: the bias gradient of the Linear whose weight gradient this is = row sums of A (= dy^T) over K, from one extra MFMA
    // per A fragment against a fragment of ones -- in the waves that own column slice 0 of the FIRST column tile only (every
    // column tile sees the same A panel).  All 16 result columns of such a tile are equal; column 0 is stored.
    const bool rs_wave = EPI == 4 && p.rowsum != nullptr && tn == 0 && wc == 0;
    f32x4 racc[EPI == 4 ? 8 : 1];
    if (EPI == 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) racc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#ifdef YAT_GEMM_STAMPS
    uint32_t st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t st_prev = 0;
    int st_slot = 0;
#endif
    // per-lane LDS addresses of the k-strided fragments (tr_lane_addr above): one for A, one (256-column) or NT for B
    const uint32_t lds0 = (uint32_t)(uintptr_t)(YAT_LDS char*)smem;
    uint32_t a_tr[1] = {0}, b_tr[NT == 4 ? 1 : NT] = {0};
    if (A_T) a_tr[0] = tr_lane_addr<BM>(0, lane);
    if (B_T && NT == 4) b_tr[0] = tr_lane_addr<G::BN>(wc * 64, lane);
    else if (B_T) {
#pragma unroll
        for (int j = 0; j < NT; ++j) b_tr[j] = tr_lane_addr<G::BN>(wc * 16 * NT + j * 16, lane);
    }
    // COMPUTE segment: 8 x NT MFMAs on registers; when dma_tile >= 0 this wave's DMA pieces for that tile are
    // issued between groups of 4 MFMAs (the matrix pipe keeps draining queued MFMAs while the wave issues a DMA),
    // so the LOAD segments carry only the fragment reads and stay shorter than the partner's COMPUTE segment.
#ifndef YAT_GEMM_DIC_ALL
#define YAT_GEMM_DIC_ALL 0          // 1: measured +4..6 % on the forward shapes alone, but the step got 2.4 ms SLOWER
#endif
    constexpr bool DIC_ = A_T || B_T || YAT_GEMM_DIC_ALL;
    // One piece every GAP MFMAs from the start of the segment (2 or 4: no measurable difference once the transposing reads
    // stopped waiting for the DMA, see frag_tr).
#ifndef YAT_GEMM_GAP
#define YAT_GEMM_GAP 2
#endif
    constexpr int GAP = YAT_GEMM_GAP;      // 256 x 192 tile: 24 MFMAs carry 7 pieces -> one every 3
    static_assert((8 * NT) / GAP >= NPIECE, "not enough MFMA slots for the DMA pieces of a tile");
    // MODE 0: no DMA in this segment; 1: scalar-offset pieces, unconditionally (two scalar instructions + the DMA each);
    // 2: decided per piece at run time (tile missing / ragged last k-tile) -- only in the last iterations of a K loop
    // The rendezvous that ends the segment is executed EARLY MFMAs before its end: the partner group, parked at the end of
    // its LOAD segment, is released while this wave's last MFMAs are still in the pipe, so the barrier's turnaround overlaps
    // them instead of idling the matrix pipe.  Legal because a COMPUTE segment touches registers only (every LDS read
    // retired before the barrier that ends the LOAD segment) and the DMA waits move with the barrier.
#ifndef YAT_GEMM_EARLY
#define YAT_GEMM_EARLY 0
#endif
    constexpr int EARLY = YAT_GEMM_EARLY;
    auto compute = [&](auto mode_c, int dma_tile, bool wait_dma) {
        constexpr int MODE = decltype(mode_c)::value;
        char* dst = smem + (dma_tile & 1) * G::STAGE;
        const bool checked = MODE == 2 && dma_tile >= 0 && is_tail(dma_tile);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[i][j] = mfma16(bfr[j], af[i], acc[i][j]);   // D[n][m]
                const int idx = i * NT + j;
                if (MODE != 0 && idx % GAP == GAP - 1 && idx / GAP < NPIECE) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (MODE == 1) piece(std::false_type{}, dma_tile, dst, idx / GAP);
                    else if (dma_tile >= 0) {
                        if (checked) piece(std::true_type{}, dma_tile, dst, idx / GAP);
                        else piece(std::false_type{}, dma_tile, dst, idx / GAP);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (EPI == 4 && idx == 8 * NT - 1 && rs_wave) {
                    bf16x8 ones;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) racc[q] = mfma16(ones, af[q], racc[q]);
                }
                if (idx == 8 * NT - 1 - EARLY) {
                    YAT_STAMP(st_slot);
#ifndef YAT_ABL_NO_VMWAIT
                    if (wait_dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile t+1 (group 0's pieces)
#endif
                    YAT_LOOP_BARRIER();
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // Where the DMA issue goes is a measured choice (gemm micro-benchmark, SANA shapes): with both operands
    // k-contiguous the load segment is short (12-13 ds_read_b128) and absorbs the issue for free, while DMA between
    // MFMAs costs 15-20 %; with a k-strided operand (twice the LDS read instructions + swizzle arithmetic) the load
    // segment is the long pole and the issue belongs in the compute segment (+12..20 %).
    constexpr bool DIC = DIC_;                 // DMA In Compute segment (group 0; group 1 too unless stated otherwise)
#ifndef YAT_GEMM_NT_G1C
#define YAT_GEMM_NT_G1C 0
#endif
    constexpr bool DIC_G1 = DIC || YAT_GEMM_NT_G1C;   // group 1 alone may prefetch tile t+2 from COMPUTE(t,ks1)

#ifndef YAT_GEMM_DEEP
#define YAT_GEMM_DEEP 1
#endif
#ifndef YAT_GEMM_PIN_WAIT
#define YAT_GEMM_PIN_WAIT 1
#endif
#ifndef YAT_GEMM_DEEP_KH
#define YAT_GEMM_DEEP_KH 1          // the DEEP schedule for a k-contiguous B operand too (round 5: half-major B image, below)
#endif
#ifdef YAT_GEMM_STAMPS
    constexpr bool DEEP = false;                           // (the stamp slots describe the two-stage schedule)
#else
    constexpr bool DEEP = YAT_GEMM_DEEP && (B_T || YAT_GEMM_DEEP_KH);
#endif
    static_assert(EPI != 5 || DEEP, "the second operand pair is wired into the DEEP loop's DMA only");
    // B image of a DEEP kernel with a k-contiguous B operand (the forward GEMMs: B = W[N, K]).  The row image of the two-stage
    // schedule keeps both 32-deep halves of a column in one 128-B row, so no half can be refilled before the other has been
    // read.  Here the image is HALF-MAJOR: [half h][column n][4 chunks of 16 B] -- 64-B rows, half h at h * 64 * BN bytes --
    // which makes it piece for piece the shape of the k-strided image (1 KiB = 16 columns of one half; BN / 16 pieces per
    // half), so the DEEP loop, its batches and its counted waits carry over unchanged.  Chunk c of column n sits at slot
    // c ^ kh_perm((n >> 2) & 3): the 16 lanes one ds_read_b128 services together (MI355X_MICROARCH.md, LDS) read columns
    // {0..3, 12..15} of one chunk and {4..11} of its neighbour and land on 16 distinct 16-B slots of the 256-B bank row.
    constexpr bool B_KH = DEEP && !B_T;
    auto kh_perm = [](uint32_t x) { return (0x78u >> (2 * x)) & 3u; };          // 0, 2, 3, 1
    const uint32_t kh_lane = ((uint32_t)(lane & 15) * 64) + ((((uint32_t)lane >> 4) ^ kh_perm((lane & 15) >> 2)) << 4);

    // the fragment reads of sub-step kk of the tile in stage t & 1 (the LDS traffic of a LOAD segment)
    auto load_frags = [&](auto grp_c, int t, int kk) {
        constexpr int GRP = decltype(grp_c)::value;
        const char* cur = smem + (t & 1) * G::STAGE;
        const uint32_t st = lds0 + (t & 1) * G::STAGE;
        if (A_T) {
            const uint32_t a0 = a_tr[0] + st + GRP * 256;            // group's rows = columns 128 GRP.. of the image
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = frag_tr<BM>(a0 ^ (i << 5), kk);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = frag256<false, BM>(cur, GRP * 128 + i * 16, kk, lane);
        }
        if (B_T && NT == 4) {
            const uint32_t b0 = b_tr[0] + st + G::A_BYTES;
#pragma unroll
            for (int j = 0; j < NT; ++j) bfr[j] = frag_tr<G::BN>(b0 ^ (j << 5), kk);
        } else if (B_T) {
#pragma unroll
            for (int j = 0; j < NT; ++j) bfr[j] = frag_tr<G::BN>(b_tr[j] + st + G::A_BYTES, kk);
        } else if (B_KH) {
            const char* b0 = cur + G::A_BYTES + kk * (G::B_BYTES / 2) + wc * (16 * NT * 64) + kh_lane;
#pragma unroll
            for (int j = 0; j < NT; ++j) bfr[j] = lds_read8(b0, j * 1024);
        } else {
#pragma unroll
            for (int j = 0; j < NT; ++j) bfr[j] = frag256<false, G::BN>(cur + G::A_BYTES, wc * 16 * NT + j * 16, kk, lane);
        }
    };

#ifdef YAT_GEMM_STAMPS
    uint32_t st_loop_begin = 0;
#endif
    if constexpr (DEEP) {
      auto deep_loop = [&](auto grp_c) {
        constexpr int GRP = decltype(grp_c)::value;
        constexpr int NB = (NT == 5 && GRP == 1) ? 3 : 2;    // B pieces per half and wave (320 columns: 20 = 4 x 2 + 4 x 3)
        constexpr int NAH = A_T ? 2 : 0;                      // k-strided A: pieces per half and wave
        constexpr int NAO = A_T ? 0 : 4;                      // k-contiguous A: pieces of the group's own 128 rows per tile and wave
        // ---- this wave's pieces.  A piece index fixes both the 1 KiB of the LDS image it fills and (make_piece) the global
        // bytes that belong there, so any assignment of pieces to waves fills the same image.  The pieces of one wave are
        // chosen a whole number of swizzle periods apart (16 or 32 k-rows of a k-strided image, 32 rows of a k-contiguous
        // one): their per-lane offsets then differ by a wave-uniform number of rows, so ONE per-lane offset per operand (two
        // for the odd piece of group 1 in the 320-column image) serves them all and the row delta rides, with the K-tile
        // advance, in the instruction's scalar offset -- 2..3 address registers where the two-stage schedule keeps 9.
        //   k-strided 256-column image (A; B of the 256-wide tile): piece = wave + 8 i + 16 h   -> k-row delta 16 i + 32 h
        //   k-strided 320-column image: 20 pieces per half = classes {q, q + 10}; group 0 wave wc: class wc; group 1 wave wc:
        //                               class 4 + wc and one piece of classes 8 / 9 (8, 18, 9, 19)       -> deltas 16 i + 32 h
        //   k-contiguous A, own rows:   piece = 16 GRP + wc + 4 j                                       -> own offsets (below)
        const int pa0 = A_T ? wave : 16 * GRP + wc;
        const int pb0 = NT == 4 ? wave : (GRP ? 4 + wc : wc);
        const int pbx = 8 + (wc >> 1) + 10 * (wc & 1);                     // (320 columns, group 1: 8, 18, 9, 19)
        const uint32_t va0 = make_piece<A_T, BM>(pa0, lane, p.lda, m0, p.M).voff;
        // (k-contiguous A: the pieces of a wave are 32 ROWS of M apart, and a scalar-offset delta is outside the buffer range
        // check -- past the last row of a ragged M it would read beyond the operand.  They keep their own per-lane offsets,
        // each with make_piece's row check.)
        uint32_t vao[A_T ? 1 : 4];
        if (!A_T) {
#pragma unroll
            for (int j = 0; j < 4; ++j) vao[j] = j == 0 ? va0 : make_piece<false, BM>(pa0 + 4 * j, lane, p.lda, m0, p.M).voff;
        }
        const uint32_t vb0 = B_KH ? 0 : make_piece<true, G::BN>(pb0, lane, p.ldb, n0, p.N).voff;
        const uint32_t vbx = (NB == 3 && !B_KH) ? make_piece<true, G::BN>(pbx, lane, p.ldb, n0, p.N).voff : 0;
        // half-major k-contiguous B image (B_KH): piece q of half h = the 64-B half rows of columns 16 q .. 16 q + 15.  The
        // pieces of a wave are 128 or 160 COLUMNS of N apart, and a scalar-offset delta is outside the buffer range check (past
        // the last column of a ragged N it would read beyond the operand): each keeps its own per-lane offset with its own column
        // check; only the half (+ 64 B inside a valid row) and the K-tile advance ride in the scalar offset.
        constexpr int PH = G::BN / 16;                                                      // pieces per half
        const uint32_t kchunk_b = ((uint32_t)lane & 3) ^ kh_perm(((uint32_t)lane >> 4) & 3);  // source chunk of a B_KH lane
        auto kh_q = [&](int i) { return NT == 4 ? pb0 + 8 * i : (i < 2 ? pb0 + 10 * i : pbx); };
        uint32_t vbk[B_KH ? NB : 1];
        if (B_KH) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int col = n0 + 16 * kh_q(i) + (lane >> 2);
                vbk[i] = col < p.N ? (uint32_t)(((int64_t)col * p.ldb + kchunk_b * 8) * 2) : YAT_OOB;
            }
        }
        const uint32_t a_row = (uint32_t)p.lda * 2, b_row = (uint32_t)p.ldb * 2;           // bytes per k-row (k-strided)
        const uint32_t kchunk_a = swz128((uint32_t)pa0 * 8 + (lane >> 3), lane & 7);       // (k-contiguous A: the same for all 4)

        // one piece: operand / image piece / source offset (per-lane + uniform) / target tile (local index tl)
        auto dma = [&](auto checked, bool is_a, int pi, uint32_t voff, uint32_t delta, int tl, int kh_half = 0) {
            constexpr bool CHECKED = decltype(checked)::value;
            const int tg = ktile(tl);
            const bool second = EPI == 5 && tg >= nt_main;                  // uniform: a tile of the second operand pair
            const int t = second ? tg - nt_main : tg;
            YAT_LDS void* dst = (YAT_LDS void*)(smem + (tl & 1) * G::STAGE + (is_a ? 0 : G::A_BYTES) + pi * 1024);
            const uint32_t soff = (uint32_t)t * (is_a ? a_kstep : b_kstep) + delta;
            if (CHECKED) {                                                  // (only the ragged tile of K: never `second`)
                uint32_t v = voff + soff;
                if (is_a && !A_T && kchunk_a >= ((uint32_t)(p.K - t * BK) >> 3)) v = YAT_OOB;
                if (!is_a && B_KH && kchunk_b + 4 * kh_half >= ((uint32_t)(p.K - t * BK) >> 3)) v = YAT_OOB;
                lds_dma16(is_a ? ra : rb, dst, v);
            } else if (second) {
                lds_dma16s(is_a ? ra2 : rb2, dst, voff, soff);
            } else {
                lds_dma16s(is_a ? ra : rb, dst, voff, soff);
            }
        };
        // piece i of half h of the k-strided images / own-row piece j of the k-contiguous A image
        auto dma_ah = [&](auto checked, int h, int i, int tl) {
            dma(checked, true, pa0 + 8 * i + 16 * h, va0, (uint32_t)(16 * i + 32 * h) * a_row, tl);
        };
        auto dma_ao = [&](auto checked, int j, int tl) { dma(checked, true, pa0 + 4 * j, vao[A_T ? 0 : j], 0u, tl); };
        auto dma_bh = [&](auto checked, int h, int i, int tl) {
            if (B_KH) dma(checked, false, kh_q(i) + PH * h, vbk[B_KH ? i : 0], 64u * h, tl, h);
            else if (NT == 4) dma(checked, false, pb0 + 8 * i + 16 * h, vb0, (uint32_t)(16 * i + 32 * h) * b_row, tl);
            else if (i < 2) dma(checked, false, pb0 + 10 * i + 20 * h, vb0, (uint32_t)(16 * i + 32 * h) * b_row, tl);
            else dma(checked, false, pbx + 20 * h, vbx, (uint32_t)(32 * h) * b_row, tl);
        };
        // batch(u), u = 2 t + kk: piece j of [own A rows of tile t + 2 (kk = 1, k-contiguous A)] [A half] [B half] where the
        // half is u + 3 + GRP: tile t + (c >> 1), half c & 1 with c = kk + 3 + GRP
        auto batch_count = [](int kk) constexpr { return (NAO && kk == 1 ? NAO : 0) + NAH + NB; };
        auto batch_piece = [&](auto kk_c, auto fast_c, int t, int j) {
            constexpr int KK = decltype(kk_c)::value;
            constexpr bool FAST = decltype(fast_c)::value;
            constexpr int C = KK + 3 + GRP, HV = C & 1, NOWN = (NAO && KK == 1) ? NAO : 0;
            const bool own = j < NOWN;
            const int tl = own ? t + 2 : t + (C >> 1);
            auto go = [&](auto checked) {
                if (own) dma_ao(checked, j, tl);
                else if (j < NOWN + NAH) dma_ah(checked, HV, j - NOWN, tl);
                else dma_bh(checked, HV, j - NOWN - NAH, tl);
            };
            if (FAST) go(std::false_type{});
            else if (tl < nt) {
                if (is_tail(tl)) go(std::true_type{});
                else go(std::false_type{});
            }
        };
        // counted waits (FAST iterations: every batch of the window was issued in full; otherwise vmcnt(0))
        constexpr int PER_TILE = NAO + 2 * NAH + 2 * NB;                   // pieces per wave and K-tile = two consecutive batches
        auto wait_g0 = [&](auto fast_c) {                                  // end of COMPUTE(u): batch(u - 2) landed
            if (decltype(fast_c)::value) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        auto wait_g1 = [&](auto kk_c, auto fast_c) {                       // end of LOAD(u): batch(u - 3) landed, and the own
            constexpr int KK = decltype(kk_c)::value;                      // A rows of batch(u - 2) when u is odd
            constexpr int N = NAO ? (KK == 1 ? 2 * NB : 2 * NB + NAO) : PER_TILE;
            if (decltype(fast_c)::value) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        auto compute_deep = [&](auto kk_c, auto fast_c, int t) {
            constexpr int KK = decltype(kk_c)::value;
            constexpr int NP = batch_count(KK);
            static_assert((8 * NT) / GAP >= NP, "not enough MFMA slots for the DMA pieces of a batch");
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = mfma16(bfr[j], af[i], acc[i][j]);   // D[n][m]
                    const int idx = i * NT + j;
                    if (idx % GAP == GAP - 1 && idx / GAP < NP) {
                        __builtin_amdgcn_sched_barrier(0);
                        batch_piece(kk_c, fast_c, t, idx / GAP);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (EPI == 4 && idx == 8 * NT - 1 && rs_wave) {
                        bf16x8 ones;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
                        for (int q = 0; q < 8; ++q) racc[q] = mfma16(ones, af[q], racc[q]);
                    }
                    if (idx == 8 * NT - 1) {
                        // (the counted wait is an asm the scheduler may hoist over the register-only MFMAs -- it did, to right
                        // behind the segment's last piece, giving up a segment of landing time: pinned behind them)
                        if (GRP == 0) {
#if YAT_GEMM_PIN_WAIT
                            __builtin_amdgcn_sched_barrier(0);
#endif
                            wait_g0(fast_c);
                        }
                        YAT_PHASE_BARRIER();
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
        };
        auto iteration_deep = [&](auto fast_c, int t) {
            // ---- sub-step 0
            load_frags(grp_c, t, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (GRP == 1) wait_g1(std::integral_constant<int, 0>{}, fast_c);
            YAT_PHASE_BARRIER();
            compute_deep(std::integral_constant<int, 0>{}, fast_c, t);
            // ---- sub-step 1
            load_frags(grp_c, t, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (GRP == 1) wait_g1(std::integral_constant<int, 1>{}, fast_c);
            YAT_PHASE_BARRIER();
            compute_deep(std::integral_constant<int, 1>{}, fast_c, t);
        };

        // prologue: tile 0 in full (every wave its own pieces), then the batches a running loop would have issued for tile 1:
        // group 0 batch(-1); group 1 batch(-2) and batch(-1) -- the counted waits of the first iterations count on them
        {
            auto all_of_tile0 = [&](auto checked) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (A_T) dma_ah(checked, j >> 1, j & 1, 0);
                    else dma_ao(checked, j, 0);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < NB; ++i) dma_bh(checked, h, i, 0);
            };
            if (is_tail(0)) all_of_tile0(std::true_type{});
            else all_of_tile0(std::false_type{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        YAT_PHASE_BARRIER();                       // rendezvous 0: tile 0 visible to everyone
        if (GRP == 1) {
#pragma unroll
            for (int j = 0; j < batch_count(0); ++j) batch_piece(std::integral_constant<int, 0>{}, std::false_type{}, -1, j);
        }
#pragma unroll
        for (int j = 0; j < batch_count(1); ++j) batch_piece(std::integral_constant<int, 1>{}, std::false_type{}, -1, j);
        if (GRP == 1) YAT_PHASE_BARRIER();         // stagger: group 1 runs one segment behind group 0

        const int nfast = max(0, nfull - 2);
        int t = 0;
        for (; t < nfast; ++t) iteration_deep(std::true_type{}, t);
        for (; t < nt; ++t) iteration_deep(std::false_type{}, t);
      };
      if (grp == 0) deep_loop(std::integral_constant<int, 0>{});
      else deep_loop(std::integral_constant<int, 1>{});
    } else {

    issue(0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    YAT_PHASE_BARRIER();                       // rendezvous 0: tile 0 visible to everyone
    if (grp == 1) {
        // (DIC) group 1 issues tile t+2 inside its COMPUTE(t, ks1), so its share of tile 1 goes out here
        if (DIC_G1 && nt > 1) issue(1, smem + G::STAGE);
        YAT_PHASE_BARRIER();                   // stagger: group 1 runs one segment behind group 0
    }

    // DMA protocol (n-th barrier of every wave is one rendezvous; group 0's iteration t spans 4t+1..4t+4, group 1's
    // 4t+2..4t+5).  Tile u lives in stage u&1 and is first read after rendezvous 4u.
    //   group 0 issues its pieces of tile t+1 in COMPUTE(t,ks0)  [after 4t+1 > 4t: stage free] and waits for them
    //           before its 4th barrier of iteration t (= 4t+4);
    //   group 1 issues its pieces of tile t+2 in COMPUTE(t,ks1)  [after 4t+4: every read of tile t retired] and waits
    //           for tile t+1's pieces before its 3rd barrier of iteration t (= 4t+4).
    //   (!DIC) every wave issues tile t+1 at the top of LOAD(t,ks0) [after its 4th barrier of iteration t-1 >= 4t]
    //           and waits before its 3rd barrier of iteration t (<= 4t+4).
    // The loop body exists per wave group (GRP a constant: which segment carries the DMA is then decided at compile time)
    // and as a FAST form -- the tile this iteration fetches exists and is a full one, so its pieces are unconditional,
    // scalar-offset DMAs with no branch and no VALU around them -- plus the general form for the last iterations.  Both
    // groups execute the same number of barriers per iteration in either form.
    auto iteration = [&](auto grp_c, auto fast_c, int t) {
        constexpr int GRP = decltype(grp_c)::value;
        constexpr bool dic = GRP == 0 ? DIC : DIC_G1;      // does THIS group issue its DMA from a compute segment?
        constexpr bool FAST = decltype(fast_c)::value;
        const char* cur = smem + (t & 1) * G::STAGE;
        // (!DIC) pieces of tile t+1 issued at the top of LOAD(t,ks0); group 0 may keep the last NPIECE - H0 of them for the top
        // of its LOAD(t,ks1) -- it then waits after COMPUTE(t,ks1) like a DIC kernel's group 0 (YAT_GEMM_NT_SPLIT)
#ifndef YAT_GEMM_NT_SPLIT
#define YAT_GEMM_NT_SPLIT 99
#endif
        const bool g0split = !dic && GRP == 0 && YAT_GEMM_NT_SPLIT < NPIECE;
        const int H0 = g0split ? YAT_GEMM_NT_SPLIT : NPIECE;
        char* nxt = smem + ((t + 1) & 1) * G::STAGE;
        // (!dic) YAT_GEMM_NT_READS_FIRST=1 puts the pieces AFTER the fragment reads of LOAD(t,ks0) (the reads then do not queue
        // behind nine pieces): +3..4 % on the forward shapes alone, 84.4 -> 84.6 ms in the step (both orders) -- off.
#ifndef YAT_GEMM_NT_READS_FIRST
#define YAT_GEMM_NT_READS_FIRST 0
#endif
        auto nt_issue = [&]() {
#ifndef YAT_ABL_NO_DMA
            if (!dic) {
                if (FAST) {
#pragma unroll
                    for (int j = 0; j < NPIECE; ++j)
                        if (j < H0) piece(std::false_type{}, t + 1, nxt, j);
                } else if (t + 1 < nt) {
                    issue_range(t + 1, nxt, 0, H0);
                }
            }
#endif
        };
        if (!YAT_GEMM_NT_READS_FIRST) nt_issue();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // ---- LOAD segment
#ifndef YAT_ABL_NO_DMA
            if (g0split && kk == 1) {
                if (FAST) {
#pragma unroll
                    for (int j = 0; j < NPIECE; ++j)
                        if (j >= H0) piece(std::false_type{}, t + 1, nxt, j);
                } else if (t + 1 < nt) {
                    issue_range(t + 1, nxt, H0, NPIECE);
                }
            }
#endif
#ifdef YAT_GEMM_LOADPRIO
            __builtin_amdgcn_s_setprio(YAT_GEMM_LOADPRIO);
#endif
            if (!YAT_ABL_SKIP_READS || t == 0) load_frags(grp_c, t, kk);
            if (YAT_GEMM_NT_READS_FIRST && kk == 0) {
                __builtin_amdgcn_sched_barrier(0);
                nt_issue();
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef YAT_ABL_NO_VMWAIT
            if (kk == 1 && ((!dic && !g0split) || GRP == 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile t+1 landed
#endif
            YAT_STAMP(4 * kk + 0);
            YAT_LOOP_BARRIER();
            YAT_STAMP(4 * kk + 1);
            // ---- COMPUTE segment: group 0 carries its DMA (tile t+1) in ks0, group 1 (tile t+2) in ks1
#ifdef YAT_ABL_NO_DMA
            compute(std::integral_constant<int, 0>{}, 0, false);
#else
#ifdef YAT_GEMM_STAMPS
            st_slot = 4 * kk + 2;
#endif
            const int tile = t + 1 + GRP;
            const bool wd = (dic || g0split) && kk == 1 && GRP == 0;                               // group 0 waits for its pieces of tile t+1
            if (dic && kk == GRP) {
                if (FAST) compute(std::integral_constant<int, 1>{}, tile, wd);
                else compute(std::integral_constant<int, 2>{}, tile < nt ? tile : -1, wd);
            } else {
                compute(std::integral_constant<int, 0>{}, 0, wd);
            }
#endif
            YAT_STAMP(4 * kk + 3);
        }
    };
    auto k_loop = [&](auto grp_c) {
        // iterations whose fetched tile (t+1; t+2 for group 1 of a DIC kernel) exists and is not the ragged last one
        const int ahead = (grp_c.value == 0 ? DIC : DIC_G1) ? 1 + grp_c.value : 1;
        const int nfast = max(0, nfull - ahead);
        int t = 0;
        for (; t < nfast; ++t) iteration(grp_c, std::true_type{}, t);
        for (; t < nt; ++t) iteration(grp_c, std::false_type{}, t);
    };
#ifdef YAT_GEMM_STAMPS
    st_prev = (uint32_t)__builtin_amdgcn_s_memtime();
    st_loop_begin = st_prev;
#endif
    if (grp == 0) k_loop(std::integral_constant<int, 0>{});
    else k_loop(std::integral_constant<int, 1>{});
    }   // (!DEEP)
#ifdef YAT_GEMM_STAMPS
    const uint32_t st_loop_end = (uint32_t)__builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) yat_gemm_stamp_buf[wave * 8 + i] = st_sum[i];
        yat_gemm_stamp_buf[64] = (unsigned)nt;
        if (wave == 0) { yat_gemm_stamp_buf[65] = st_loop_begin - st_kernel_begin; yat_gemm_stamp_buf[66] = st_loop_end - st_loop_begin; }
    }
#endif
    if (grp == 0) YAT_PHASE_BARRIER();         // pair group 1's last barrier

    if (EPI == 4 && rs_wave && p.ksplit == 1 && (lane >> 4) == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + grp * 128 + i * 16 + lane;
            if (m < p.M) {
                float v = racc[i][0];
                if (p.rowsum_acc) v = rbf(v) + bf2f(p.rowsum[m]);
                p.rowsum[m] = f2bf(v);
            }
        }
    }
    // ---- epilogue.  The MFMA layout gives a lane 4 columns of 16 different rows (8-B accesses in 32-B runs).  When
    // everything is 16-B aligned the accumulators go through the (now free) LDS instead: each wave transposes 32 rows
    // at a time in a private padded fp32 slab and reads back 8 consecutive columns per lane, so bias / gate / residual
    // loads and C / aux stores are 16 B per lane over whole 128..160-B row segments.
    const int rpb = p.rows_per_batch > 0 ? p.rows_per_batch : p.M;
    const bool wide = !(p.N & 7) && !(p.ldc & 7) && !(p.res && (p.ldr & 7)) && !(p.aux && (p.ldaux & 7)) &&
                      !(p.gate && (p.gate_ld & 7)) && !(p.glu_u && (p.ld_glu & 7)) && !(p.pre_add && (p.ld_pre & 7)) &&
                      !(p.dact_z && (p.ld_z & 7));
    const bool split = p.ksplit > 1;           // host guarantees `wide` alignment when splitting
    // (Round 5 built the 8-column units by a LANE EXCHANGE instead -- v_permlane16_swap of two neighbouring result tiles leaves
    // every lane with 8 consecutive columns of one row, no LDS slab, no lgkmcnt waits; bit-identical -- and it was 3 .. 10 %
    // SLOWER per launch (step 76.5 -> 79.7 ms): a store instruction then covers 16 rows x 64 B instead of 6.4 rows x 160 B.
    // The epilogue is bound by the row segments its stores touch, not by the LDS round trip.  profiles/r05_g_*.)
    if (wide || split) {
        constexpr int WCOLS = 16 * NT;             // columns per wave
        constexpr int LDW = WCOLS + 4;             // padded fp32 row (conflict-free ds_write_b128)
        constexpr int CPR = WCOLS / 8;             // 8-column chunks per row
        float* slab = reinterpret_cast<float*>(smem) + wave * (32 * LDW);
        const int g4 = lane >> 4, li = lane & 15;
        // The epilogue's own global INPUTS (residual, pre_add, GLU u, activation input z) are loaded one pass of 32 rows
        // ahead of their use, into the registers the pass before has just emptied: read where they are consumed, each of the
        // four passes of a tile paid an HBM round trip behind its slab round trip (the gated-residual and GLU-backward GEMMs
        // ran 714 .. 990 TFLOP/s in the step).  Same loads, same arithmetic, same stores: bit-identical.
#ifndef YAT_GEMM_EPI_PREFETCH
#define YAT_GEMM_EPI_PREFETCH 1
#endif
        constexpr int UPL = (32 * CPR + 63) / 64;          // 8-column units per lane and pass
        constexpr bool HAS_IN = YAT_GEMM_EPI_PREFETCH && (EPI == 1 || EPI == 2 || EPI == 3 || EPI == 0 || EPI == 4 || EPI == 5);
        const bool pre_on = HAS_IN && !split && (EPI == 1 || EPI == 3 || EPI == 2 || p.res != nullptr);
        // one input stream (residual | z): a whole pass ahead, double-buffered; two streams (GLU u_a + u_g, residual + pre_add):
        // the registers do not hold two passes of both -- loaded at the top of their own pass, before its slab round trip
        constexpr bool AHEAD = EPI != 1 && EPI != 2;
        EpiIn ein[AHEAD ? 2 : 1][UPL];
        auto unit = [&](int pass, int k, int& row, int& ch, int& m, int& n) {
            const int u = lane + 64 * k;
            row = u / CPR;
            ch = u % CPR;
            m = m0 + grp * 128 + pass * 32 + row;
            n = n0 + wc * WCOLS + ch * 8;
            return u < 32 * CPR && m < p.M && n < p.N;
        };
        auto prefetch = [&](int pass, EpiIn (&in)[UPL]) {
#pragma unroll
            for (int k = 0; k < UPL; ++k) {
                int row, ch, m, n;
                if (!unit(pass, k, row, ch, m, n)) continue;
                if (EPI == 1) {
                    const bf16_t* up = p.glu_u + (int64_t)m * p.ld_glu + n;
                    in[k].a = *reinterpret_cast<const u32x4*>(up);
                    in[k].b = *reinterpret_cast<const u32x4*>(up + p.N);
                } else if (EPI == 3) {
                    in[k].a = *reinterpret_cast<const u32x4*>(p.dact_z + (int64_t)m * p.ld_z + n);
                } else {
                    if (p.res) in[k].a = *reinterpret_cast<const u32x4*>(p.res + (int64_t)m * p.ldr + n);
                    if (EPI == 2) in[k].b = *reinterpret_cast<const u32x4*>(p.pre_add + (int64_t)m * p.ld_pre + n);
                }
            }
        };
        if (pre_on && AHEAD) prefetch(0, ein[0]);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            if (pre_on && !AHEAD) prefetch(pass, ein[0]);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    *reinterpret_cast<f32x4*>(slab + (ii * 16 + li) * LDW + j * 16 + 4 * g4) = acc[pass * 2 + ii][j];
            if (pre_on && AHEAD && pass + 1 < 4) prefetch(pass + 1, ein[(pass + 1) & 1]);       // (into the registers this pass freed)
            // wave-private slab: the wave's own LDS writes are ordered before its reads by lgkmcnt; no barrier needed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < UPL; ++k) {
                int row, ch, m, n;
                if (unit(pass, k, row, ch, m, n)) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + row * LDW + ch * 8);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + row * LDW + ch * 8 + 4);
                    if (split) {               // fp32 partial slab [ksl][m][n]; epilogue runs in the reduce kernel
                        float* dst = p.partial + ((int64_t)ksl * p.M + m) * p.N + n;
                        *reinterpret_cast<f32x4*>(dst) = lo;
                        *reinterpret_cast<f32x4*>(dst + 4) = hi;
                    } else {
                        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        const EpiIn* pin = pre_on ? &ein[AHEAD ? (pass & 1) : 0][k] : nullptr;
                        if (EPI == 1) glu_bwd_store<8>(p, v, m, n, pin);
                        else if (EPI == 3) act_bwd_store<8>(p, v, m, n, pin);
                        else gemm_epilogue_store8<EPI == 2, A_T && B_T>(p, v, m, n, m / rpb, pin);     // (weight-gradient layout: non-temporal result)
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads retired before the next pass overwrites the slab
        }
#ifdef YAT_GEMM_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (blockIdx.x == 0 && lane == 0 && wave == 0) yat_gemm_stamp_buf[67] = (uint32_t)__builtin_amdgcn_s_memtime() - st_loop_end;
#endif
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + grp * 128 + i * 16 + (lane & 15);
        if (m >= p.M) continue;
        const int b = m / rpb;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + wc * 16 * NT + j * 16 + 4 * (lane >> 4);
            if (n >= p.N) continue;
            if (EPI == 1 || EPI == 3) {
                const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (EPI == 1) glu_bwd_store<4>(p, v, m, n);
                else act_bwd_store<4>(p, v, m, n);
            } else {
                gemm_epilogue_store<EPI == 2>(p, acc[i][j], m, n, b);
            }
        }
    }
}

// One tile per workgroup.  Measured in round 5 and not kept (profiles/r05_d_*): a PERSISTENT form (256 workgroups walking the
// tile list, same tile -> XCD assignment, bit-identical) is 1.7 .. 2.3 % faster alone on launches of five and more rounds,
// 4.6 % slower on two-round ones and 0.5 % slower in the step, where its grid holds every CU against the other stream; the
// K sweep it came with puts the fixed cost of a tile round at 6.6 us against 1.75 us per K-tile -- the time is in the loop,
// not between tiles.  Delaying the first round's workgroups so that the epilogues leave lockstep changes nothing (the
// epilogue is not HBM-bound: cold outputs cost nothing).
template <bool A_T, bool B_T, int NT, int EPI = 0>
__global__ __launch_bounds__(512, 1) void gemm256_kernel(GemmP p) {
    gemm256_body<A_T, B_T, NT, EPI>(p, xcd_contiguous(p.nbm * p.nbn * p.ksplit));
}

// Grouped launch: up to YAT_MAX_GROUP independent problems of one layout in ONE grid (problem g owns work items
// [first[g], first[g+1])).  A transformer block's seven weight-gradient GEMMs launched one by one leave 5..45 % of the
// CUs idle in their last round (81..396 tiles on 256 CUs) or pay for split-K slabs; together they are ~1240 full-K
// tiles = 4.85 rounds.  The descriptors travel in the kernel argument segment (no device allocation, no copy).
constexpr int YAT_MAX_GROUP = 8;
struct GroupedP {
    int ngroups;
    int first[YAT_MAX_GROUP + 1];
    GemmP g[YAT_MAX_GROUP];
};

template <bool A_T, bool B_T, int NT>
__global__ __launch_bounds__(512, 1) void gemm256_grouped_kernel(GroupedP gp) {
    const int id = xcd_contiguous(gp.first[gp.ngroups]);
    int g = 0;
    while (g + 1 < gp.ngroups && id >= gp.first[g + 1]) ++g;
    // the weight-gradient layout carries the row sums of A (bias gradients) for the problems that ask for them
    gemm256_body<A_T, B_T, NT, (A_T && B_T && NT == 4) ? 4 : 0>(gp.g[g], id - gp.first[g]);
}

template <bool A_T, bool B_T, int NT>
int launch256_grouped(int ngroups, const GemmP* probs, hipStream_t stream) {
    using G = Geo<NT>;
    if (ngroups < 1 || ngroups > YAT_MAX_GROUP) return YAT_EINVAL;
    GroupedP gp{};
    gp.ngroups = ngroups;
    for (int i = 0; i < ngroups; ++i) {
        GemmP q = probs[i];
        q.nbm = (q.M + BM - 1) / BM;
        q.nbn = (q.N + G::BN - 1) / G::BN;
        q.ksplit = 1;
        q.partial = nullptr;
        gp.g[i] = q;
        gp.first[i + 1] = gp.first[i] + q.nbm * q.nbn;
    }
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gemm256_grouped_kernel<A_T, B_T, NT>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess)
            return YAT_EINVAL;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm256_grouped_kernel<A_T, B_T, NT>), dim3(gp.first[ngroups]), dim3(512), G::LDS, stream, gp);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? YAT_OK : (int)e;
}

// (gemm1w -- the same 256 x 256 x 64 tile with ONE wave per SIMD and the K loop software-pipelined inside each wave, tile
// variant 6 -- was measured in round 2, never won a shape and is gone; profiles/LOG_r01_r03.md section 9.)

template <bool A_T, bool B_T, int NT, int EPI = 0>
int launch256(const GemmP& p0, hipStream_t stream) {
    using G = Geo<NT>;
    GemmP p = p0;
    p.nbm = (p.M + BM - 1) / BM;
    p.nbn = (p.N + G::BN - 1) / G::BN;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gemm256_kernel<A_T, B_T, NT, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                G::LDS) != hipSuccess)
            return YAT_EINVAL;
        attr_set = true;
    }
    if (p.ksplit < 1) p.ksplit = 1;
    hipLaunchKernelGGL((gemm256_kernel<A_T, B_T, NT, EPI>), dim3(p.nbm * p.nbn * p.ksplit), dim3(512), G::LDS, stream, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? YAT_OK : (int)e;
}

}  // namespace

// out[m, n..n+7] = epilogue( sum_s partial[s][m][n..] ): one thread per 8 columns, 16-B accesses
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmP p) {
    const int64_t nchunk = (int64_t)p.M * (p.N >> 3);
    const int rpb = p.rows_per_batch > 0 ? p.rows_per_batch : p.M;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < nchunk; u += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(u / (p.N >> 3)), n = (int)(u % (p.N >> 3)) * 8;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int s = 0; s < p.ksplit; ++s) {
            const float* src = p.partial + ((int64_t)s * p.M + m) * p.N + n;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
            v[0] += lo[0]; v[1] += lo[1]; v[2] += lo[2]; v[3] += lo[3];
            v[4] += hi[0]; v[5] += hi[1]; v[6] += hi[2]; v[7] += hi[3];
        }
        gemm_epilogue_store8(p, v, m, n, m / rpb);
    }
}

int yat_gemm_splitk_reduce(const GemmP& p, hipStream_t stream) {
    const int64_t nchunk = (int64_t)p.M * (p.N >> 3);
    int64_t nb = (nchunk + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? YAT_OK : (int)e;
}

// grouped launch, 256 x 256 tiles (the one geometry whose tile counts add up well for the weight-gradient set)
int yat_gemm256_grouped_launch(int a_t, int b_t, int ngroups, const GemmP* probs, hipStream_t stream) {
#define YAT_CASE(AT, BT) \
    if (a_t == AT && b_t == BT) return launch256_grouped<AT, BT, 4>(ngroups, probs, stream);
    YAT_CASE(false, false)
    YAT_CASE(false, true)
    YAT_CASE(true, true)
    YAT_CASE(true, false)
#undef YAT_CASE
    return YAT_EINVAL;
}

// variant: 4 -> BN=256, 5 -> BN=320.  (A 256 x 192 tile -- no ragged column tile at N = 1152 / 3456 / 4608 -- was
// instantiated and measured at the PixArt shapes: never faster than the policy's pick, 5-25 % slower than BN=320; removed.)
int yat_gemm256_launch(int a_t, int b_t, int nt_variant, const GemmP& p, hipStream_t stream) {
    if (p.glu_u) {                 // GLU-backward epilogue: only the dgrad layout (dy W) is instantiated, no split-K
        if (a_t || !b_t || p.ksplit > 1) return YAT_EINVAL;
        return nt_variant == 5 ? launch256<false, true, 5, 1>(p, stream) : launch256<false, true, 4, 1>(p, stream);
    }
    if (p.dact_z) {                // activation-backward epilogue: dgrad layout only, no split-K
        if (a_t || !b_t || p.ksplit > 1) return YAT_EINVAL;
        return nt_variant == 5 ? launch256<false, true, 5, 3>(p, stream) : launch256<false, true, 4, 3>(p, stream);
    }
    if (p.rowsum) {                // weight gradient + bias gradient: (1,1) layout, 256 x 256 tile, whole K per workgroup
        if (!a_t || !b_t || p.ksplit > 1 || nt_variant != 4) return YAT_EINVAL;
        return launch256<true, true, 4, 4>(p, stream);
    }
    if (p.A2) {                    // second operand pair: forward layout, no split-K
        // (wired into the DEEP loop's DMA only: the diagnostic builds without that loop for the forward layout --
        //  -DYAT_GEMM_STAMPS, -DYAT_GEMM_DEEP=0, -DYAT_GEMM_DEEP_KH=0 -- do not instantiate EPI 5 and refuse the call)
#if !defined(YAT_GEMM_STAMPS) && YAT_GEMM_DEEP && YAT_GEMM_DEEP_KH
        if (a_t || b_t || p.ksplit > 1) return YAT_EINVAL;
        return nt_variant == 5 ? launch256<false, false, 5, 5>(p, stream) : launch256<false, false, 4, 5>(p, stream);
#else
        return YAT_EINVAL;
#endif
    }
    if (p.pre_add) {               // adapter addend: only the forward layout (x W^T) is instantiated, no split-K
        if (a_t || b_t || p.ksplit > 1) return YAT_EINVAL;
        return nt_variant == 5 ? launch256<false, false, 5, 2>(p, stream) : launch256<false, false, 4, 2>(p, stream);
    }
#define YAT_CASE(AT, BT)                                                     \
    if (a_t == AT && b_t == BT)                                              \
        return nt_variant == 5 ? launch256<AT, BT, 5>(p, stream) : launch256<AT, BT, 4>(p, stream);
    YAT_CASE(false, false)
    YAT_CASE(false, true)
    YAT_CASE(true, true)
    YAT_CASE(true, false)
#undef YAT_CASE
    return YAT_EINVAL;
}

#ifdef YAT_GEMM_STAMPS
extern "C" int yat_debug_gemm_stamps(unsigned int* host_dst) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(yat_gemm_stamp_buf), sizeof(unsigned int) * 72);
}
#endif
