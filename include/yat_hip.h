/* libyat_hip.so -- C ABI of the MI355X (gfx950) kernels behind the SANA training-step hot path.
 *
 * The reference (frutiemax92/YAT) is pure Python and has no FFI layer; its hot path lands on
 * third-party CUDA kernels through torch / diffusers calls.  Each entry point below replaces one
 * of those call sites (cited as file:line under /root/reference).  A reference maintainer binds
 * them with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions: all pointers are DEVICE pointers owned by the caller (torch's ROCm allocator);
 * bf16 tensors are raw uint16 storage; row-major with explicit leading dimensions in ELEMENTS;
 * `stream` is a hipStream_t passed as void*; every call only enqueues work (no sync, no
 * allocation, graph-capturable) and returns 0 on success, a negative yat error for bad arguments
 * or a positive hipError_t from the launch.  The library keeps no global mutable state except the
 * RCCL communicator of the communication section at the end, and reads no environment variable: the
 * YAT_* tuning switches named in the sources exist only in builds made with -DYAT_TUNING
 * (scripts/build_variant.py), in the product library each is its constant default.
 */
#ifndef YAT_HIP_H
#define YAT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* yat_stream_t; /* hipStream_t */

/* status codes: 0 ok; negative = argument / state error of this library; 1..999 = a hipError_t from a launch or a HIP
 * runtime call; YAT_ECOMM_BASE + ncclResult_t = an RCCL error (yat_comm_last_error() holds the text). */
#define YAT_OK 0
#define YAT_EINVAL (-1)
#define YAT_ENOCOMM (-2)      /* communicator not initialised, or librccl could not be bound */
#define YAT_ECOMM_BASE 1000

int yat_version(void);

/* ------------------------------------------------------------------------------------------ *
 * GEMM family (MFMA bf16, fp32 accumulate)
 * replaces: every nn.Linear / 1x1 Conv2d forward+backward in the SANA block
 *   attn1/attn2 projections  utils/patch_sana_attention_layers.py:39-65,94,98-104
 *   GLUMBConv 1x1 convs      utils/patch_sana_attention_layers.py:68,110-113
 *   patch_embed / proj_out / caption / time-embed linears  utils/patched_sana_transformer.py:119-136,165,284-298,333
 *   their autograd backward  common/trainer.py:344
 * ------------------------------------------------------------------------------------------ */
typedef struct yat_gemm_epilogue {
    uint32_t struct_size;  /* = sizeof(yat_gemm_epilogue) as the CALLER compiled it (yat_gemm_epilogue_size() for a binding  */
                           /* that cannot take sizeof).  The library copies that many bytes and zero-fills the rest, so a     */
                           /* caller built against an older, shorter layout keeps working when options are appended;        */
                           /* anything below the first layout (64 bytes: through rows_per_batch), above the library's own   */
                           /* size or not a multiple of 8 is rejected with YAT_EINVAL -- never read past the caller's object */
    const void* bias;      /* bf16 [N] or NULL: v += bias[n]                                       */
    void* aux_out;         /* bf16 [M, ld_aux] or NULL: stores the Linear output (pre-activation / */
                           /* pre-gate), rounded to bf16, for the backward pass                     */
    int activation;        /* 0 none, 1 SiLU, 2 GELU(tanh)                                         */
    const void* gate;      /* bf16 [M/rows_per_batch, ld_gate] or NULL: v = gate[b, n] * v          */
    const void* residual;  /* bf16 [M, ld_residual] or NULL: v += residual[m, n]  (may alias C)     */
    int ld_aux, ld_gate, ld_residual, rows_per_batch;
    const void* glu_u;     /* bf16 [M, ld_glu_u] = [u_a | u_g] (N columns each) or NULL.  GLU backward fused into the */
    int ld_glu_u;          /* GEMM that produces dy (GLUMBConv conv_point dgrad): with d = bf16(result), C is [M, 2N]:  */
                           /* C[m,n] = d*bf16(SiLU(u_g)), C[m,N+n] = bf16(d*u_a)*SiLU'(u_g); excludes the other options */
    const void* pre_add;   /* bf16 [M, ld_pre_add] or NULL (forward layout only): added to the rounded Linear output      */
    int ld_pre_add;        /* before aux_out / activation / gate / residual -- a PEFT adapter's                          */
                           /* base_layer(x) + F.linear(x, delta_w) (peft LoKr/LoRA wrap at common/trainer.py:212-238)     */
    const void* dact_z;    /* bf16 [M, ld_dact_z] or NULL (dgrad layout only): activation backward fused into the GEMM that   */
    int ld_dact_z;         /* produces the activation's output gradient: C = bf16(bf16(result) * act'(z)), `activation`       */
                           /* selecting act (1 SiLU, 2 GELU-tanh) -- FeedForward's GELU in PixArt-Sigma (net.0 -> net.2);      */
                           /* excludes the other options                                                                       */
    void* a_rowsum_out;    /* bf16 [M] or NULL (wgrad layout (1,1) only, no split-K): out[m] = bf16(sum_k A_op[m,k]) -- the     */
    int a_rowsum_accumulate; /* BIAS gradient of the Linear whose weight gradient this GEMM computes (A = dy: column sums of    */
                           /* dy), from one extra MFMA per A fragment against a fragment of ones in the workgroups of the first */
                           /* column tile instead of a second pass over dy; accumulate: out[m] = bf16(bf16(sum) + out[m])      */
                           /* (gradient accumulation, the `residual` = C convention)                                           */
    const void* a2;        /* SECOND OPERAND PAIR, or NULL (forward layout (0,0) only, no split-K): C = epilogue(A B^T + A2 B2^T), */
    const void* b2;        /* both products in one fp32 accumulator, rounded once -- a PEFT adapter's factored term folded into    */
    int k2;                /* the base Linear: base_layer(x) + T P^T (LoKr: T = x' w2_b^T, P = kron(w1, w2_a) alpha/r; peft's       */
    int a2_group_n;        /* wrap at common/trainer.py:212-238) without the [M, N] addend of `pre_add`.  A2 is bf16 [M, >= k2]   */
                           /* with A's row stride lda, B2 bf16 [N, k2] with B's row stride ldb -- they are read as k2 further     */
                           /* columns of A and B that live at other addresses; k2 a multiple of 64.  a2_group_n = g > 0 (a        */
                           /* multiple of 320, or of 256): columns [j g, (j+1) g) of C take columns [j k2, (j+1) k2) of A2 (a     */
                           /* fused q|k|v Linear with one adapter per block); excludes glu_u / pre_add / dact_z / a_rowsum_out    */
} yat_gemm_epilogue;

uint64_t yat_gemm_epilogue_size(void);   /* sizeof(yat_gemm_epilogue) in this build of the library */

/* C[M,N] = epilogue(A_op * B_op).  a_t=0: A is [M,K] (k contiguous); a_t=1: A is [K,M].
 * b_t=0: B is [N,K] (k contiguous, nn.Linear weight layout); b_t=1: B is [K,N].
 * (0,0) forward y = x W^T; (0,1) dgrad dx = dy W; (1,1) wgrad dW = dy^T x; (1,0) x^T W^T.
 * Epilogue order: +bias -> round bf16 -> aux_out -> activation -> *gate (rounded) -> +residual.  */
int yat_gemm_bf16(int a_t, int b_t, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                  void* C, int ldc, const yat_gemm_epilogue* ep, yat_stream_t stream);

/* Same, with an optional split-K workspace and a per-call POLICY WORD `variant` = v + 100*s + 10000*c:
 *   v  tile variant: 0 = automatic policy, 1 = 128x128 tile, 4 = 256x256 tile, 5 = 256x320 tile (two staggered wave
 *      groups, gemm256.hip);
 *   s  0 = the policy decides; 1..32 forces that split-K factor on variant 4 or 5 (tests / tuning);
 *   c  host hint for the automatic policy: how many independent streams of GEMMs the caller keeps in flight while this
 *      launch runs (0 or 1 = the launch has the chip to itself; the training step passes 2: two forward chains, dgrad
 *      beside wgrad; at most 8).  With more than one, a launch is charged its CU-time (but at least 3/8 of the chip)
 *      instead of whole rounds, so it is not split or narrowed to fill a round its neighbour would have filled.  The hint
 *      travels with the call -- the library keeps no policy state -- and changes tile / split choice only, i.e. the
 *      summation order of the result, never its meaning.  No reference counterpart (torch picks its GEMM algorithms per
 *      call, unaware of the caller's streams).
 * workspace (fp32, >= ksplit*M*N*4 bytes) lets the policy split K for shapes that leave CUs idle (small outputs with a
 * long reduction: weight gradients); with workspace == NULL K is never split. */
int yat_gemm_bf16_ex(int a_t, int b_t, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                     void* C, int ldc, const yat_gemm_epilogue* ep, int variant, void* workspace,
                     uint64_t workspace_bytes, yat_stream_t stream);

/* Grouped GEMM: `count` (<= 8) independent problems of ONE layout (a_t, b_t) in a single launch of 256 x 256 tiles,
 * each with its own epilogue.  Replaces the per-layer sequence of nn.Linear weight-gradient GEMMs torch autograd issues
 * one by one under loss.backward() (common/trainer.py:341): a SANA block's seven dW = dy^T x products are 81..396 tiles
 * each -- poor fills of 256 CUs on their own, ~1240 tiles (4.85 rounds) together, with no split-K slabs. */
typedef struct yat_gemm_problem {
    int M, N, K;
    const void* A; int lda;
    const void* B; int ldb;
    void* C; int ldc;
    const yat_gemm_epilogue* epilogue;   /* nullable */
} yat_gemm_problem;
int yat_gemm_grouped_bf16(int a_t, int b_t, int count, const yat_gemm_problem* problems, yat_stream_t stream);

/* out[c] (+)= sum_r x[r, c]  (bias gradients).  workspace: >= yat_colsum_workspace_bytes(rows, cols). */
uint64_t yat_colsum_workspace_bytes(int rows, int cols);
int yat_colsum_bf16(int rows, int cols, const void* x, int ld, void* out_bf16, int accumulate, void* workspace,
                    yat_stream_t stream);

/* ------------------------------------------------------------------------------------------ *
 * adaLN-single modulation table  (patch_sana_attention_layers.py:85-87, SanaModulatedNorm)
 *   mod[b, s, :] = bf16(table[s, :] + tmod[b, s*slot_stride : +D])      (slot_stride = D or 0)
 * backward: dtable[s,:] = sum_b dmod[b,s,:];  dtmod_acc[b, s*slot_stride + d] += dmod[b,s,d]  (fp32)
 * ------------------------------------------------------------------------------------------ */
int yat_modulation_fwd(int B, int S, int D, const void* table, const void* tmod, int tmod_ld, int slot_stride,
                       void* mod_out, yat_stream_t stream);
int yat_modulation_bwd(int B, int S, int D, const float* dmod, void* dtable_bf16, int accumulate_table,
                       float* dtmod_acc, int tmod_ld, int slot_stride, yat_stream_t stream);

/* ------------------------------------------------------------------------------------------ *
 * LayerNorm(no affine) + modulate   (patch_sana_attention_layers.py:38,90-92 and :53,107-108;
 *                                    final norm patched_sana_transformer.py:163-164,331)
 *   y = bf16( bf16( bf16(LN(x)) * bf16(1 + scale[b]) ) + shift[b] ),  stats in fp32
 * shift/scale: bf16 [B, mod_ld] rows (pointers already offset to the slot); b = row / rows_per_batch.
 * bwd: dx = (dres?) + LN'(dy * (1+scale));  dshift_acc[b,:] += sum_n dy;  dscale_acc[b,:] += sum_n dy * xhat
 *      (fp32 accumulators with leading dimension acc_ld).
 *      parts: 1 = dx only (the dependent chain), 2 = dshift/dscale only (needed by nobody downstream: the caller may
 *      run it on another stream), 3 = both.
 * ------------------------------------------------------------------------------------------ */
uint64_t yat_ln_bwd_workspace_bytes(int M, int D, int rows_per_batch);
int yat_ln_modulate_fwd(int M, int D, int rows_per_batch, float eps, const void* x, const void* shift,
                        const void* scale, int mod_ld, void* y, float* mean, float* rstd, yat_stream_t stream);
int yat_ln_modulate_bwd(int M, int D, int rows_per_batch, const void* x, const float* mean, const float* rstd,
                        const void* scale, int mod_ld, const void* dy, const void* dres, void* dx,
                        float* dshift_acc, float* dscale_acc, int acc_ld, void* workspace, int parts,
                        yat_stream_t stream);

/* RMSNorm with affine weight, eps inside the sqrt (caption_norm, patched_sana_transformer.py:136,298)
 *   y = bf16( bf16(x * rsqrt(mean(x^2) + eps)) * w ).   bwd: dx, dw (bf16, optional accumulate). */
uint64_t yat_rmsnorm_bwd_workspace_bytes(int M, int D);
int yat_rmsnorm_fwd(int M, int D, float eps, const void* x, const void* w, void* y, float* rstd, yat_stream_t stream);
int yat_rmsnorm_bwd(int M, int D, const void* x, const void* w, const float* rstd, const void* dy, void* dx,
                    void* dw_bf16, int accumulate_dw, void* workspace, yat_stream_t stream);

/* ------------------------------------------------------------------------------------------ *
 * ReLU linear attention (SanaLinearAttnProcessor2_0, imported patch_sana_attention_layers.py:7)
 * qkv: bf16 [B*N, ld] with q at column 0, k at column k_off, v at column v_off, heads of dim 32
 * laid out head-major inside each of q/k/v.  out: bf16 [B*N, ld_out].  fp32 math throughout:
 *   S = [V;1] relu(K),  O = S relu(Q),  out = O[:32] / (O[32] + 1e-15)
 * ------------------------------------------------------------------------------------------ */
uint64_t yat_linear_attn_workspace_bytes(int B, int N, int H);
int yat_linear_attn_fwd(int B, int N, int H, const void* qkv, int ld, int k_off, int v_off, void* out, int ld_out,
                        void* workspace, yat_stream_t stream);
/* state: the forward's per-head state = the first B*H*33*32 floats of the forward workspace if the caller kept it
 * (saves recomputing it), or NULL to recompute. */
int yat_linear_attn_bwd(int B, int N, int H, const void* qkv, int ld, int k_off, int v_off, const void* dout,
                        int ld_dout, void* dqkv, int ld_dqkv, const float* state, void* workspace, yat_stream_t stream);

/* ------------------------------------------------------------------------------------------ *
 * Masked softmax cross-attention (attn2 = AttnProcessor2_0 / F.scaled_dot_product_attention,
 * patch_sana_attention_layers.py:54-65,98-104; mask->bias patched_sana_transformer.py:275-277)
 *   q: bf16 [B*N, ldq] (head h at col h*dh), k,v: bf16 [B*T, ldkv]; key_bias: float [B, T]
 *   (0 keep / -10000 drop); kv_len[b] = 1 + last kept key (tiles past it are skipped -- exact,
 *   their probabilities underflow to 0 -- unless kv_len[b]==0, then all T keys are used).
 *   out: bf16 [B*N, ldo]; lse: float [B, H, N] (natural log).  dh <= 128, dh % 8 == 0.
 * The same entry points serve every softmax attention of the path: SANA attn2 (dh 112, T = 512), the softmax variant of
 * attn1 in SANA's modified_blocks (dh 32; patch_sana_attention_layers.py:125-131) and both attentions of a PixArt-Sigma
 * block (dh 72; self-attention over N = T = 4096 with q, k, v = the column blocks of the fused [3D] projection and a
 * zero key_bias; utils/patch_pixart_sigma_transformer.py:150-158).  The kernels pick a head-dim instantiation and the
 * workgroup shape (64 or 128 queries) from dh, N, H, B.
 * key_bias == NULL (with kv_len == NULL, no work_list, not the packed form): plain attention over all T keys of every
 * image -- the self-attentions of PixArt-Sigma and of the MMDiT (JointAttnProcessor2_0 passes no mask).  Same result as a
 * zero bias; the kernels then fold the scale into the exponential's own multiply-add and (forward) take the row sums out
 * of the P V product, which is what the long key loops (N = T = 4096 ... 4429) are short of: vector issue slots.
 * ------------------------------------------------------------------------------------------ */
int yat_sdpa_fwd(int B, int N, int T, int H, int dh, float scale, const void* q, int ldq, const void* k, const void* v,
                 int ldkv, const float* key_bias, const int* kv_len, void* out, int ldo, float* lse, yat_stream_t stream);
/* work_list (optional, device int32 [n_work][2] = (batch, key tile) for every key tile with tile*64 < kv_len[batch], n_work
 * known on the host from the embedding lengths): the dK/dV kernel then launches only those workgroups; NULL = dense
 * grid over all T/64 tiles with early exit (correct, but idle LDS-heavy workgroups cost ~0.24 us each).
 * parts: 1 = dQ and delta (what the dependent chain needs), 2 = dK/dV (reads delta; feeds only the text-side weight
 * gradients, so the caller may run it on another stream after part 1), 3 = both.
 * delta (float [B, H, N], caller-owned scratch): part 1 writes rowsum(dO * O) there for part 2. */
int yat_sdpa_bwd(int B, int N, int T, int H, int dh, float scale, const void* q, int ldq, const void* k, const void* v,
                 int ldkv, const float* key_bias, const int* kv_len, const void* out, int ldo, const void* dout, int lddo,
                 const float* lse, float* delta, void* dq, int lddq, void* dk, void* dv, int lddkv, const int* work_list,
                 int n_work, int parts, yat_stream_t stream);
/* Packed keys: the same two operations when the text side keeps NO padding rows (train_sana.py:168-176 pads every prompt
 * to 512 rows; the K / V projections of the padding are computed there and then masked out with the -10000 bias, their
 * probabilities and gradients being exactly zero).  Image b's K / V (and dK / dV) rows are
 * [kv_row_offsets[b], kv_row_offsets[b] + kv_len[b]) of one [kv_rows, ldkv] matrix (kv_len[b] >= 1 for every image);
 * key_bias / kv_len / work_list keep their [B, T] / [B] / (batch, tile) meaning.  Rows of dk / dv outside those ranges are
 * not written.  Results are bit-identical to the padded entry points on the same keys. */
int yat_sdpa_fwd_packed(int B, int N, int T, int H, int dh, float scale, const void* q, int ldq, const void* k, const void* v,
                        int ldkv, const int* kv_row_offsets, int kv_rows, const float* key_bias, const int* kv_len, void* out,
                        int ldo, float* lse, yat_stream_t stream);
int yat_sdpa_bwd_packed(int B, int N, int T, int H, int dh, float scale, const void* q, int ldq, const void* k, const void* v,
                        int ldkv, const int* kv_row_offsets, int kv_rows, const float* key_bias, const int* kv_len,
                        const void* out, int ldo, const void* dout, int lddo, const float* lse, float* delta, void* dq, int lddq,
                        void* dk, void* dv, int lddkv, const int* work_list, int n_work, int parts, yat_stream_t stream);

/* ------------------------------------------------------------------------------------------ *
 * GLUMBConv middle: SiLU -> depthwise 3x3 (pad 1, bias) -> chunk2 -> a * SiLU(g)
 * (diffusers GLUMBConv called at patch_sana_attention_layers.py:110-113), on the token-major
 * layout [B, h, w, 2*Hc] (no NCHW round trip).  z is the conv_inverted output, s = bf16(SiLU(z)); both are written by
 * the conv_inverted GEMM (activation = SiLU, aux_out = z), so SiLU is evaluated once per element.
 *   y[b,i,j,c] = u_c * silu(u_{c+Hc}),  u = bias + sum_taps wdw[c, tap] * s[b,i+di,j+dj,c]
 * wdw: bf16 [2*Hc, 9] (diffusers conv_depth.weight [2Hc,1,3,3] flattened), bdw: bf16 [2*Hc].
 * bwd: dz (bf16, includes the SiLU derivative), dwdw / dbdw partial sums reduced via workspace;
 *      dz_colsum_bf16 (nullable, [2Hc]) (+)= sum over pixels of dz: the bias gradient of conv_inverted, taken in the
 *      same pass instead of a separate yat_colsum_bf16 over dz.
 * fwd u_out (nullable, bf16 [B,h,w,2Hc]): the conv output u (pre-GLU) kept for the backward;
 * bwd du_in (nullable, bf16 [B,h,w,2Hc]): du already computed (yat_gemm_epilogue.glu_u in the GEMM that produces dy,
 *      from the kept u) -- pass 1 (recompute u, GLU backward) is skipped and `dy` / `s`-recompute are not needed.
 * `s` in the backward: read only by pass 1 (du_in == NULL: u is recomputed from s exactly as the forward computed it).
 *      Pass 2 never reads it, on any shape: every pass-2 kernel recomputes s = bf16(z * sigmoid(z)) from the `z` it loads
 *      for SiLU' (band / global-z kernels since round 5 -- a quarter of the pass's bytes --, the fallback kernel for w > 64
 *      or 2Hc not a multiple of 8 since round 6), with this library's SiLU (gemm_common.hpp silu_f: what the conv_inverted
 *      GEMM's epilogue and yat_act_fwd_bf16 store).  dwdw therefore does not depend on which kernel a shape selects; a
 *      caller whose `s` came from another SiLU implementation gets dwdw for THIS library's s.
 * ------------------------------------------------------------------------------------------ */
uint64_t yat_dwconv_glu_bwd_workspace_bytes(int B, int h, int w, int Hc);
int yat_dwconv_glu_fwd(int B, int h, int w, int Hc, const void* s, const void* wdw, const void* bdw, void* y,
                       void* u_out, yat_stream_t stream);
int yat_dwconv_glu_bwd(int B, int h, int w, int Hc, const void* s, const void* z, const void* wdw, const void* bdw,
                       const void* dy, void* dz, void* dwdw_bf16, void* dbdw_bf16, void* dz_colsum_bf16, int accumulate,
                       void* workspace, const void* du_in, yat_stream_t stream);

/* ------------------------------------------------------------------------------------------ *
 * gated residual backward: out = res + bf16(gate[b,:] * lin)   (patch_sana_attention_layers.py:95,113)
 *   dlin = bf16(gate * dout);  dgate_acc[b,:] += sum_n dout * lin   (fp32)
 *   dbias_bf16 (nullable): (+)= sum_rows dlin -- the bias gradient of the Linear that produced `lin`
 *   (attn1.to_out.0), taken in the same pass instead of a separate yat_colsum_bf16 over dlin.
 * ------------------------------------------------------------------------------------------ */
uint64_t yat_gate_bwd_workspace_bytes(int M, int D, int rows_per_batch);
int yat_gate_bwd(int M, int D, int rows_per_batch, const void* dout, const void* lin, const void* gate, int gate_ld,
                 void* dlin, float* dgate_acc, int acc_ld, void* dbias_bf16, int accumulate_bias, void* workspace,
                 yat_stream_t stream);

/* ------------------------------------------------------------------------------------------ *
 * LoKr adapters (BASELINE config 5; peft LoKrConfig wrap at common/trainer.py:226-238).
 * For a target weight [out = out_l*out_k, in = in_m*in_n]:  w1 bf16 [out_l, in_m], w2_a [out_k, r], w2_b [r, in_n].
 *   yat_lokr_delta:   delta[(i,k),(j,n)] = bf16( bf16(w1[i,j] * bf16((w2_a w2_b)[k,n])) * scale )   (scale = alpha / r)
 *   yat_lokr_project: from d_delta [out, ld] (= dy^T x, a yat_gemm wgrad): d_w1, d_w2_a, d_w2_b as autograd of the
 *                     line above produces them (fp32 sums in a fixed order, one bf16 rounding per autograd op).
 * The dense products with delta run on yat_gemm_bf16 (yat_gemm_epilogue.pre_add folds the adapter output in).
 * ------------------------------------------------------------------------------------------ */
int yat_lokr_delta(int out_l, int out_k, int in_m, int in_n, int r, const void* w1, const void* w2_a, const void* w2_b,
                   float scale, void* delta, int ld, yat_stream_t stream);
uint64_t yat_lokr_project_workspace_bytes(int out_l, int out_k, int in_n);
int yat_lokr_project(int out_l, int out_k, int in_m, int in_n, int r, const void* w1, const void* w2_a, const void* w2_b,
                     float scale, const void* d_delta, int ld, void* d_w1, void* d_w2_a, void* d_w2_b, void* workspace,
                     yat_stream_t stream);

/* Factored application of a LoKr adapter (no dense delta_w): with x viewed as [rows = M*in_m, in_n],
 *   T1 = x' w2_b^T            (yat_gemm_bf16, N = r)        P = kron(w1, w2_a) * scale  [out, in_m*r]  (yat_lokr_delta with
 *   adapter(x) = T1_flat P^T  (yat_gemm_bf16, K = in_m*r)       w2_b := identity)
 *   dx += (dy P)' w2_b,  d_P = dy^T T1_flat (then yat_lokr_project -> d_w1, d_w2_a),  d_w2_b = (dy P)'^T x'
 * -- the last one has r x in_n outputs and a rows-long reduction: out[q, n] (+)= bf16(scale * sum_row a[row, q] * x[row, n]),
 * q < r_out.  a: bf16 [rows, R] (R = 8 or 16), x: bf16 [rows, N] with row stride ldx (N % 8 == 0; column blocks of 128 are
 * separate workgroups, so N may be a whole layer width: the d_A / d_B of a plain LoRA adapter), out: bf16 [r_out, ldo]. */
/* the two row-streaming products of that path: backward=0: io[rows, R] = a[rows, N] w2_b^T;
 * backward=1: io[rows, N] = bf16(bf16(a[rows, R] w2_b) + io)   (w2_b: bf16 [R, N], R = 8 or 16, N <= 128, N % 8 == 0) */
int yat_lokr_rows(int64_t rows, int N, int R, int backward, const void* w2_b, const void* a, void* io, yat_stream_t stream);
/* the forward one with T1 laid out as the second operand of the base Linear's GEMM (yat_gemm_epilogue.a2): t1_flat[m, j R + q] =
 * t1[m in_m + j, q], row stride ldt (= the row stride of x as the GEMM's A: [M, in]), rows = M in_m */
int yat_lokr_rows_fwd_flat(int64_t rows, int N, int R, int in_m, const void* w2_b, const void* x, void* t1_flat, int ldt,
                           yat_stream_t stream);
/* nn.Dropout(p) on a LoRA adapter's input (peft lora/layer.py [RECALL]; LoraConfig(lora_dropout=...) at common/trainer.py:215):
 * keep(i) = hash(seed, i) >= p (counter-based, so the backward regenerates the mask); backward_add=0: io = bf16(x * keep / (1-p));
 * backward_add=1: io = bf16(io + bf16(x * keep / (1-p)))  (gradient through the same mask, accumulated into an input gradient) */
int yat_dropout(int64_t n, float p, uint64_t seed, int backward_add, const void* x, void* io, yat_stream_t stream);
/* plain LoRA with the adapter term as the base GEMM's second operand pair (yat_gemm_epilogue.a2 = T = x A^T in K2 = 64 columns with
 * x's row stride, b2 = scaling * lora_B with the weight's row stride): this entry writes b2 for EVERY adapter in one launch --
 * dst[tgt + n in + q] = bf16(scale * src[off + q out + n]), q < R (8 or 16), n < out, with table[4 e ..] = {off, tgt, out, in} as int64
 * (device memory), dst a zero-filled shadow of the model's flat weights; in % 8 == 0 and tgt % 8 == 0 (16-byte rows) */
int yat_lora_scatter_b(int entries, int max_out, int R, float scale, const void* table, const void* src, void* dst,
                       yat_stream_t stream);
/* rank-R expansion over a whole layer width (plain LoRA, peft lora/layer.py [RECALL]; the reference's LoraConfig branch at
 * common/trainer.py:214-219): io[row, n] = f(sum_q h[row, q] * w[q, n]), h: bf16 [rows, R], w: bf16 [R, N], io: bf16 [rows, ldio];
 * residual=0: io = bf16(bf16(sum) * scale)  (lora_B(lora_A(x)) * scaling);  residual=1: io = bf16(bf16(sum) + io)  (dx += dT A) */
int yat_rank_expand(int64_t rows, int N, int R, const void* w, const void* h, void* io, int ldio, float scale, int residual,
                    yat_stream_t stream);
uint64_t yat_lokr_small_wgrad_workspace_bytes(int64_t rows, int R, int N);
int yat_lokr_small_wgrad(int64_t rows, int R, int N, int r_out, const void* a, const void* x, int ldx, void* out, int ldo,
                         float scale, int accumulate, void* workspace, yat_stream_t stream);

/* elementwise helpers: y = act(x) and dx = dy * act'(x) on bf16 (time-embed / caption MLPs);
 * act: 1 SiLU, 2 GELU(tanh).  add: out = bf16(a + b).  f32->bf16 convert. */
int yat_act_fwd(int64_t n, int act, const void* x, void* y, yat_stream_t stream);
int yat_act_bwd(int64_t n, int act, const void* x, const void* dy, void* dx, yat_stream_t stream);
int yat_add_bf16(int64_t n, const void* a, const void* b, void* out, yat_stream_t stream);
int yat_f32_to_bf16(int64_t n, const float* x, void* y, yat_stream_t stream);

/* batched transpose in[B,R,C] -> out[B,C,R]: NCHW latents <-> token-major rows
 * (PatchEmbed flatten/transpose patched_sana_transformer.py:284; unpatchify :336-340). */
/* clears nbytes at ptr on the stream (the fp32 gradient accumulators the backward zeroes per block): an entry point rather
 * than a torch op so that a step is nothing but C-ABI calls + stream/event operations (yat_amd/flat.py launch plans) */
int yat_memset_zero(void* ptr, uint64_t nbytes, yat_stream_t stream);
int yat_transpose_bf16(int B, int R, int C, const void* in, void* out, yat_stream_t stream);

/* sinusoidal timestep projection (diffusers get_timestep_embedding(t,256,flip_sin_to_cos=True),
 * used via AdaLayerNormSingle at patched_sana_transformer.py:133,291-293): out bf16 [B, dim],
 * cos half first. */
int yat_timestep_embed_fwd(int B, int dim, const float* t, void* out, yat_stream_t stream);

/* ------------------------------------------------------------------------------------------ *
 * recipe ops (train_sana.py:168-180, 206-207, 217-218)
 * ------------------------------------------------------------------------------------------ */
/* ragged text embeddings -> padded [B, T, C] bf16 + int64 mask + float key bias + kv_len.
 * src: bf16 concatenation of the B [L_i, C] matrices; offsets: int32 [B+1] row offsets. */
int yat_pad_mask(int B, int T, int C, const void* src, const int* offsets, void* dst, int64_t* mask, float* key_bias,
                 int* kv_len, yat_stream_t stream);
/* the same without padding rows: dst [rows_padded, C] = the source rows followed by zero rows (rows_padded >= offsets[B]);
 * mask / key_bias / kv_len exactly as yat_pad_mask writes them ([B, T] / [B]).  Feeds the packed-key attention above. */
int yat_pack_mask(int B, int T, int C, int rows_padded, const void* src, const int* offsets, void* dst, int64_t* mask,
                  float* key_bias, int* kv_len, yat_stream_t stream);
/* noisy = bf16(bf16((1-s)*x) + bf16(s*n)), target = bf16(n - x); sigma: bf16 [B] */
int yat_flow_mix(int B, int64_t per_sample, const void* x, const void* noise, const void* sigma, void* noisy,
                 void* target, yat_stream_t stream);
/* loss = mean((pred - target)^2) in fp32 (written to loss[0]); dpred = bf16(2 (pred-target) * gscale / n) */
int yat_mse_fwd_bwd(int64_t n, const void* pred, const void* target, float gscale, float* loss, void* dpred,
                    float* workspace_256, yat_stream_t stream);

/* ------------------------------------------------------------------------------------------ *
 * PixArt-Sigma recipe and embedding glue (BASELINE config 3: train_pixart_sigma.py:151-185 over
 * utils/patch_pixart_sigma_transformer.py:124-198).
 * ------------------------------------------------------------------------------------------ */
/* NCHW [B,C,H,W] <-> token rows [B*(H/p)*(W/p), C*p*p].  channel_major=1: column c*p*p+pi*p+pj (PatchEmbed's
 * Conv2d(k=p,s=p) weight [D,C,p,p] flattened, patch_pixart_sigma_transformer.py:130); channel_major=0: column
 * (pi*p+pj)*C+c (unpatchify "nhwpqc->nchpwq", :186-191).  to_tokens=1 gathers image->tokens, 0 scatters tokens->image. */
int yat_patch_rearrange(int B, int C, int H, int W, int p, int channel_major, int to_tokens, const void* src, void* dst,
                        yat_stream_t stream);
/* out[r,:] = bf16(x[r,:] + pos[r % N,:]): PatchEmbed's "(latent + pos_embed).to(latent.dtype)" with the fp32 sin-cos table */
int yat_add_pos_embed(int64_t rows, int N, int D, const void* x, const float* pos, void* out, yat_stream_t stream);
/* DDPMScheduler.add_noise (train_pixart_sigma.py:176): noisy = bf16(bf16(a[b]*x) + bf16(c[b]*n)); a, c: bf16 [B] */
int yat_ddpm_add_noise(int B, int64_t per_sample, const void* x, const void* noise, const void* sqrt_alpha_prod,
                       const void* sqrt_one_minus_alpha_prod, void* noisy, yat_stream_t stream);
/* MSELoss()(pred.chunk(2,1)[0].to(bf16), noise) in bf16 (train_pixart_sigma.py:180-184): pred [B, stride] of which the first
 * `used` elements per sample are compared with target [B, used]; loss[0] (fp32 slot holding the bf16-rounded value);
 * dpred [B, stride] = bf16(bf16(bf16(2/n)*bf16(p-t)) * bf16(gscale)), zero in the dropped half.  dpred may be NULL. */
int yat_mse_bf16_chunk(int B, int64_t used, int64_t stride, const void* pred, const void* target, float gscale, float* loss,
                       void* dpred, float* workspace_256, yat_stream_t stream);

/* peft LoHa adapters (lora_algo: loha, LoHaConfig at common/trainer.py:220-224) [RECALL peft/tuners/loha/layer.py HadaWeight]:
 *   yat_hadamard_scale: out = bf16(bf16(a * b) * scale) -- delta_w = ((w1a w1b) * (w2a w2b)) * scale from the two rank-r
 *       products (ordinary yat_gemm_bf16 calls);
 *   yat_hadamard_bwd:   g = bf16(dd * scale); t1 = bf16(g * a2); t2 = bf16(g * a1) -- the element-wise half of HadaWeight's
 *       hand-written backward (d_w1a = t1 w1b^T, d_w1b = w1a^T t1, d_w2a = t2 w2b^T, d_w2b = w2a^T t2 are GEMMs again).
 *   rows x cols views (cols % 8 == 0) with row strides in elements. */
int yat_hadamard_scale(int rows, int cols, const void* a, int lda, const void* b, int ldb, float scale, void* out, int ldo,
                       yat_stream_t stream);
int yat_hadamard_bwd(int rows, int cols, const void* dd, int ldd, const void* a1, int ld1, const void* a2, int ld2, float scale,
                     void* t1, int ldt1, void* t2, int ldt2, yat_stream_t stream);

/* DoRA (LoraConfig(..., use_dora=params.lora_use_dora), common/trainer.py:215-220) [RECALL peft/tuners/lora/dora.py]:
 * with lw = lora_B lora_A (an ordinary yat_gemm_bf16 call), u = bf16(W + bf16(scaling*lw)), n_j = bf16(||u_j||) (detached) and
 * s_j = bf16(mag_j / n_j), peft's result = base(x) + (s-1)(x W^T) + s*scaling*lora(x) equals base(x) + x delta^T with
 *   yat_dora_delta: delta_j = bf16(s_j (W_j + scaling lw_j) - W_j); s_buf / n_buf (fp32 [rows]) keep s and n for the backward;
 *   yat_dora_bwd:   from dd = d_delta (the dense weight gradient dy^T x): dmag_j = bf16((sum_l dd[j,l] u[j,l]) / n_j) and
 *                   t1 = bf16(bf16(s_j*scaling) * dd) -- d_lora_B = t1 lora_A^T and d_lora_A = lora_B^T t1 are GEMMs again.
 * rows x cols views (cols % 8 == 0), row strides in elements. */
int yat_dora_delta(int rows, int cols, const void* W, int ldw, const void* lw, int ldl, const void* mag, float scaling,
                   void* delta, int ldd, float* s_buf, float* n_buf, yat_stream_t stream);
int yat_dora_bwd(int rows, int cols, const void* dd, int ldd, const void* W, int ldw, const void* lw, int ldl, float scaling,
                 const float* s_buf, const float* n_buf, void* t1, int ldt, void* dmag, yat_stream_t stream);

/* ------------------------------------------------------------------------------------------ *
 * MMDiT glue (SD3.5-Medium, BASELINE config 4: the reference trains diffusers' SD3Transformer2DModel,
 * train_sd35.py:4,188-191; JointTransformerBlock / JointAttnProcessor2_0 [RECALL], restated in oracle/sd3_ref.py).
 * Joint layout: per image, N image-token rows followed by T text-token rows (torch.cat(dim=2) of the two streams).
 *   yat_qknorm_concat_fwd: joint[:, 0:D | D:2D] = per-head RMSNorm(dh, eps, affine) of the q | k projections of both
 *       streams (attn.norm_q / norm_k on image rows, attn.norm_added_q / norm_added_k on text rows:
 *       y = bf16(bf16(x * rsqrt(mean_dh(x^2) + eps)) * w)), joint[:, 2D:3D] = v; rstd [B*(N+T), 2H] kept for the backward.
 *       T = 0 (qkv_txt NULL): the plain self-attention `attn2` of the dual-attention blocks.  dh in {32, 64, 128}.
 *   yat_qknorm_concat_bwd: d_joint (gradients of normalised q | k and of v, joint rows) -> dqkv_img / dqkv_txt in the
 *       streams' own row order, + the four norm-weight gradients [dh] (fixed-order two-stage reduction).
 *   yat_joint_rows: rows of C channels between the joint layout and the per-stream layouts (to_joint = 1: concatenate,
 *       a NULL txt writes zeros; 0: split, a NULL txt skips the text rows) -- the attention output and its gradient.
 * ------------------------------------------------------------------------------------------ */
int yat_qknorm_concat_fwd(int B, int N, int T, int H, int dh, float eps, const void* qkv_img, int ld_img, const void* qkv_txt,
                          int ld_txt, const void* wq_img, const void* wk_img, const void* wq_txt, const void* wk_txt,
                          void* joint, int ld_joint, float* rstd, yat_stream_t stream);
uint64_t yat_qknorm_concat_bwd_workspace_bytes(int B, int N, int T, int dh);
int yat_qknorm_concat_bwd(int B, int N, int T, int H, int dh, const void* qkv_img, int ld_img, const void* qkv_txt, int ld_txt,
                          const void* wq_img, const void* wk_img, const void* wq_txt, const void* wk_txt, const float* rstd,
                          const void* d_joint, int ld_dj, void* dqkv_img, int ld_dimg, void* dqkv_txt, int ld_dtxt,
                          void* dwq_img, void* dwk_img, void* dwq_txt, void* dwk_txt, int accumulate_dw, void* workspace,
                          yat_stream_t stream);
int yat_joint_rows(int B, int N, int T, int C, void* joint, int ld_joint, void* img, int ld_img, void* txt, int ld_txt,
                   int to_joint, yat_stream_t stream);

/* ------------------------------------------------------------------------------------------ *
 * optimizer (common/trainer.py:246-248,347-348,356 = torch clip_grad_norm_ + torch.optim.AdamW on
 * bf16 params with bf16 states), over ONE flat parameter buffer.
 *   seg_start: int64 [nseg+1] element offsets of each parameter tensor (16-B aligned starts).
 *   gradnorm: per-tensor bf16-rounded L2 norms -> total (bf16-rounded, as torch does on bf16 grads)
 *             -> clip_coef[0] = min(1, bf16(max_norm / (total + 1e-6))); norm_out[0] = total.
 *   adamw: torch's single-tensor op sequence with a bf16 rounding after every op; grads are
 *          multiplied by clip_coef first (rounded to bf16) when clip_coef != NULL; zero_grad
 *          clears the gradient buffer in the same pass.
 * ------------------------------------------------------------------------------------------ */
uint64_t yat_gradnorm_workspace_bytes(int64_t n, int nseg);
int yat_gradnorm_clip(int64_t n, const void* grad, int nseg, const int64_t* seg_start, float max_norm, float* norm_out,
                      float* clip_coef, void* workspace, yat_stream_t stream);
/* The same norm over PIECES (round 6; what yat_amd/optim.py FlatAdamW calls): a piece is a whole tensor or the part of a
 * tensor inside one eighth of its data-parallel bucket.  piece_start[npiece + 1]: element offsets (16-byte aligned, ascending);
 * chunk_base[npiece + 1]: prefix sums of ceil(piece length / 2^18) -- partial[chunk_base[p] + c] receives the sum of squares of
 * chunk c of piece p, or 0 when owned != NULL and owned[p] == 0; max_piece_chunks = the largest piece's chunk count.
 * finish: tensor t = pieces [tensor_first_piece[t], tensor_first_piece[t + 1]); per-tensor sums in piece / chunk order, then
 * torch's clip_grad_norm_ arithmetic as in yat_gradnorm_clip.  Partition invariance: a data-parallel rank that holds reduced
 * gradients only for its shard of every bucket sums only its own pieces; the partial arrays of all ranks are added (each slot
 * has one non-zero contributor, so the sum is exact) and finish yields the clip coefficient of the one-rank run bit for bit.
 * replaces: the same torch.nn.utils.clip_grad_norm_ (common/trainer.py:347). */
int yat_gradnorm_pieces_partial(const void* grad, int npiece, const int64_t* piece_start, const int* chunk_base,
                                int max_piece_chunks, const unsigned char* owned, float* partial, yat_stream_t stream);
int yat_gradnorm_pieces_finish(int ntensor, const int* tensor_first_piece, const int* chunk_base, const float* partial,
                               float max_norm, float* norm_out, float* clip_coef, yat_stream_t stream);
/* background: 0 = full-width launch (the update alone on the GPU); N > 0 = at most N workgroups of a 48-VGPR variant
 * that can share a CU with two resident GEMM waves -- for an update that runs under the next forward's GEMMs. */
int yat_adamw_step(int64_t n, void* param, void* grad, void* exp_avg, void* exp_avg_sq, const float* clip_coef,
                   double lr, double beta1, double beta2, double eps, double weight_decay, int step, int zero_grad,
                   void* ema_shadow, double ema_decay, int background, yat_stream_t stream);

/* ------------------------------------------------------------------------------------------ *
 * launch plans: replay a recorded sequence of entry-point calls and stream / event operations in ONE call.
 * A training step over the same buffers issues the same ~900 launches and ~500 stream / event operations every time
 * (yat_amd/flat.py records them); replaying the list from C costs ~1 us per entry instead of a host-language call each.
 * replaces: nothing in the reference -- it is the host-side launch path of the step loop (common/trainer.py:337-344 drives
 * diffusers / torch, whose per-op dispatch is what a step's host time consists of there too).
 *   op >= 0: the entry point with that id (yat_plan_op_id(name); any int-returning yat_* function); a[i] = argument i
 *            (integers and pointers in .i / .p, float and double parameters in .d);
 *   YAT_PLAN_EVENT_RECORD: hipEventRecord(a[0].p, a[1].p);   YAT_PLAN_STREAM_WAIT_EVENT: hipStreamWaitEvent(a[0].p, a[1].p).
 * Stops at the first failing entry: returns its status and stores its index in *failed_index.
 * ------------------------------------------------------------------------------------------ */
#define YAT_PLAN_MAX_ARGS 30
#define YAT_PLAN_EVENT_RECORD (-1)
#define YAT_PLAN_STREAM_WAIT_EVENT (-2)
typedef union yat_plan_arg { int64_t i; double d; void* p; } yat_plan_arg;
typedef struct yat_plan_entry { int32_t op; int32_t nargs; yat_plan_arg a[YAT_PLAN_MAX_ARGS]; } yat_plan_entry;
int yat_plan_op_id(const char* name);    /* -1: not a replayable entry point */
int yat_plan_replay(const yat_plan_entry* entries, int n, int* failed_index);

/* ------------------------------------------------------------------------------------------ *
 * communication: data-parallel gradient reduction over RCCL / xGMI, one process per GPU.
 * replaces: Accelerate's DDP wrap -- Accelerator(...) + DistributedDataParallelKwargs common/trainer.py:31-37,
 *   accelerator.prepare (rank0 -> all parameter broadcast) :253, the bucketed all-reduce(avg) fired inside
 *   accelerator.backward(loss) :344.
 * The library owns the communicator and one completion event per bucket (its only global state); the caller owns the
 * buffers and decides WHEN a bucket is ready (the backward schedule, yat_amd/sana.py) -- bucket scheduling is host logic
 * tied to the model's launch order, so it stays with the launch order; this is the transport.  RCCL is bound at run time
 * (the copy the process already holds, else the system librccl.so.1), so the library loads without it.
 *   yat_comm_available : YAT_OK when RCCL can be bound in this process (no communicator, no socket, no GPU call): lets every
 *       rank agree that the transport exists BEFORE the collective yat_comm_init, which a missing rank would hang
 *   yat_comm_unique_id : rank 0 draws the 128-byte rendezvous id; the caller ships it to the other ranks (env, store, file)
 *   yat_comm_init      : collective over all `world` ranks; one communicator per process
 *   yat_comm_broadcast : in-place broadcast of `nbytes` from `root` on `stream` (parameters at start-up)
 *   yat_bucket_allreduce_async : records an event on producer_stream, makes comm_stream wait for it, enqueues the in-place
 *       all-reduce(mean) of nbytes/2 bf16 gradients on comm_stream and records the bucket's completion event; returns at
 *       once.  producer_stream == comm_stream skips the first event.  bucket_id in [0, 256).
 *   yat_comm_allreduce : in-place all-reduce of `count` elements on `stream`, stream-ordered like any kernel; dtype 0 = bf16,
 *       1 = f32; op 0 = mean, 1 = sum.  For the bulk collectives outside the bucket schedule (the EMA mean over ranks before
 *       validation, common/trainer.py:374-377), so that they too use the library's one communicator
 *   yat_bucket_reduce_scatter_async : as yat_bucket_allreduce_async, but a reduce-scatter(mean): rank r receives, in place,
 *       the mean of bytes [r nbytes / world, (r + 1) nbytes / world) of the bucket; the other slices keep the rank's local
 *       values.  nbytes must be a multiple of 16 * world.  With yat_comm_allgather this is the sharded optimizer step (SURVEY.md
 *       section 5: reduce-scatter -> AdamW on 1 / world of every bucket -> all-gather; the same bytes on the wire as the
 *       all-reduce); completion through yat_comm_wait like any bucket
 *   yat_comm_allgather : in-place all-gather on `stream`: every rank contributes slice `rank` of the nbytes at ptr and
 *       receives all of them (the updated parameters of a bucket); nbytes a multiple of 16 * world
 *   yat_comm_wait      : compute_stream waits (on the device; the host does not block) for bucket_id, or for every
 *       outstanding bucket when bucket_id < 0
 *   yat_comm_destroy   : releases the communicator and the events
 * ------------------------------------------------------------------------------------------ */
#define YAT_COMM_ID_BYTES 128
int yat_comm_available(void);
int yat_comm_unique_id(void* id_out_128);
int yat_comm_init(int rank, int world, const void* unique_id_128);
int yat_comm_world(void);                /* 0 before yat_comm_init */
int yat_comm_rank(void);                 /* -1 before yat_comm_init */
int yat_comm_broadcast(void* ptr, uint64_t nbytes, int root, yat_stream_t stream);
int yat_bucket_allreduce_async(void* ptr, uint64_t nbytes, int bucket_id, yat_stream_t producer_stream,
                               yat_stream_t comm_stream);
int yat_comm_allreduce(void* ptr, uint64_t count, int dtype, int op, yat_stream_t stream);
int yat_bucket_reduce_scatter_async(void* ptr, uint64_t nbytes, int bucket_id, yat_stream_t producer_stream,
                                    yat_stream_t comm_stream);
int yat_comm_allgather(void* ptr, uint64_t nbytes, yat_stream_t stream);
int yat_comm_wait(int bucket_id, yat_stream_t compute_stream);
int yat_comm_destroy(void);
const char* yat_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* YAT_HIP_H */
