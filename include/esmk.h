/*
 * esmk.h - C ABI of the MI355X (gfx950) transformer-trunk kernels.
 *
 * This is the drop-in boundary for the ESM C / ESM3 trunk forward.  The reference has no
 * FFI of its own: its seam is the nn.Module boundary plus the optional third-party
 * flash-attn operators selected by `use_flash_attn` (reference esm/layers/blocks.py:90-98,
 * esm/layers/attention.py:10-13,117-119, esm/layers/rotary.py:246-261, esm/models/esmc.py:142-164).
 * Every entry point below names the reference code it replaces.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, explicit sizes, a HIP stream passed as void*; no torch types.
 *   - every function returns 0 on success, <0 on error (ESMK_ERR_*); esmk_last_error() gives text.
 *     Nothing throws, aborts or exits.  Shapes are validated on the host before any launch.
 *   - the caller owns every buffer; nothing here allocates device memory or synchronises the device
 *     (exceptions, both named where they are declared: esmk_event_elapsed_ms waits for its event, and
 *     esmk_gemm_noalloc uses a fixed device array that belongs to the library).
 *   - work is enqueued on `stream` and is ordered with other work on that stream.
 *   - process-wide state: a thread-local error string; per kernel a bit mask of the devices on which its
 *     dynamic-LDS limit has been raised (set once, idempotent); per row-wise kernel and device the grid size that
 *     fills that device (an occupancy query, cached in atomics); the per-device address of the esmk_gemm_noalloc
 *     array.  No environment variable is read by this library (the ESMK_GEMM_* switches exist only in the
 *     separate experiments build, libesmk_x.so).  Entry points are re-entrant across threads and streams unless
 *     their comment says otherwise.
 *   - activations: packed tokens, row-major [n_tokens, width]; no row padding is required.
 *   - "packed weight": bf16 [round_up(N,256), round_up(K,64)] produced by esmk_pack_weight.
 */
#ifndef ESMK_H
#define ESMK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESMK_VERSION 300

#define ESMK_OK 0
#define ESMK_ERR_INVALID_ARG (-1)
#define ESMK_ERR_UNSUPPORTED (-2)
#define ESMK_ERR_LAUNCH (-3)
#define ESMK_ERR_WORKSPACE (-4)

/* element types of activation buffers */
#define ESMK_F32 0
#define ESMK_BF16 1

typedef void* esmk_stream_t; /* hipStream_t */

int esmk_version(void);
const char* esmk_last_error(void);
/* 12 hex digits: hash of the sources this binary was built from (written by the Makefile).  bench.py and the
 * profiles under profiles/ carry it, so a measured number can be tied to one binary.  No reference counterpart. */
const char* esmk_build_id(void);

/* ------------------------------------------------------------------------------------------
 * Weight packing (once per model).
 * Replaces: nothing at run time; consumes the reference state_dict tensors (fp32 [N,K] row-major,
 * SURVEY.md 8b) and lays them out for the MFMA GEMM: bf16, rows padded to 256, K padded to 64.
 * mode ESMK_PACK_SWIGLU interleaves the gate rows [0,N/2) and up rows [N/2,N) of ffn.1.weight
 * (reference esm/layers/blocks.py:26-27) in groups of 16 so one MFMA tile pair owns gate+up of the
 * same channels.
 * ------------------------------------------------------------------------------------------ */
#define ESMK_PACK_PLAIN 0
#define ESMK_PACK_SWIGLU 1
size_t esmk_packed_weight_bytes(int N, int K);
int esmk_pack_weight(const float* w_f32, void* packed, int N, int K, int mode, esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (eps inside sqrt, biased variance - torch.nn.LayerNorm).
 * Replaces: nn.LayerNorm in layernorm_qkv[0], ffn[0], transformer.norm, RegressionHead[2],
 * geom_attn.s_norm (reference esm/layers/attention.py:26-28, blocks.py:31,
 * transformer_stack.py:62,94, regression_head.py:20, geom_attention.py:38,65).
 * x [M, d] (x_dtype) -> y [M, d] (y_dtype); beta may be NULL.
 * ------------------------------------------------------------------------------------------ */
int esmk_layernorm(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                   int M, int d, float eps, esmk_stream_t stream);

/* Fused residual update + LayerNorm:  x (f32 [M,d], in/out) += delta (bf16 [M,d]);  y = LN(x).
 * Replaces the residual add of UnifiedTransformerBlock (reference esm/layers/blocks.py:147-155; the 1/s scale is
 * applied by the producing GEMM's epilogue) fused with the LayerNorm that consumes the updated stream
 * (attention.py:26, blocks.py:31, geom_attention.py:65, transformer_stack.py:94).  delta == NULL: plain LN. */
int esmk_add_layernorm(float* x, const void* delta_bf16, const float* gamma, const float* beta, void* y,
                       int y_dtype, int M, int d, float eps, esmk_stream_t stream);
/* Two-delta form used by the trunk to touch the fp32 stream once per block instead of twice:
 *   write_x = 0, delta_b = NULL : y = LN(x + delta_a), x untouched (delta_a stays pending)
 *   write_x = 1                 : x = (x + delta_a) + delta_b (delta_b may be NULL), y = LN(x)
 * Same sums in the same order as two esmk_add_layernorm calls => bit-identical results. */
int esmk_add2_layernorm(float* x, const void* delta_a_bf16, const void* delta_b_bf16, int write_x, const float* gamma,
                        const float* beta, void* y, int y_dtype, int M, int d, float eps, esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * GEMM  out = epilogue(A[M,K] . W[N,K]^T)  on bf16 MFMA with fp32 accumulation.
 * Replaces: every nn.Linear on the path (reference attention.py:27,29,82; blocks.py:32-37;
 * geom_attention.py:43,45; regression_head.py:18,21) with the op that follows it fused:
 *   ESMK_EPI_BF16       out bf16 [M,N]            = alpha * acc            (alpha = 1/s makes it a residual delta)
 *   ESMK_EPI_F32        out f32  [M,N]            = acc
 *   ESMK_EPI_SWIGLU     out bf16 [M,N/2]          = silu(gate) * up       (blocks.py:26-27; W packed SWIGLU)
 *   ESMK_EPI_RESIDUAL   out f32  [M,N] (in/out)   += alpha * acc           (blocks.py:147-155, alpha = 1/s)
 *   ESMK_EPI_BIAS_GELU  out bf16 [M,N]            = gelu_erf(acc + bias)   (regression_head.py:18-19)
 * bias (f32 [N], 16-byte aligned, reference row order; NULL = none) is added to acc in every epilogue - the
 * bias=True block variants (blocks.py:81, vqvae.py:154); for SWIGLU it is ffn.1.bias = [gate bias | up bias].
 * A: bf16 row-major with leading dimension lda >= round_up(K,64) (pad columns must be zero or finite:
 * the packed weight is zero there).
 * ------------------------------------------------------------------------------------------ */
#define ESMK_EPI_BF16 0
#define ESMK_EPI_F32 1
#define ESMK_EPI_SWIGLU 2
#define ESMK_EPI_RESIDUAL 3
#define ESMK_EPI_BIAS_GELU 4
int esmk_gemm(const void* A_bf16, int lda, const void* W_packed, int M, int N, int K, int epilogue,
              const float* bias, float alpha, void* out, esmk_stream_t stream);
/* Same operation with a caller-owned scratch buffer: small problems (few output tiles, long K) then split K over
 * several workgroups per tile - fp32 partials in the workspace, summed in a fixed order (deterministic) by a
 * second pass that applies the epilogue.  esmk_gemm_workspace_bytes() returns 0 when the shape would not be split;
 * a NULL / too small workspace falls back to the unsplit kernels (same results up to fp32 summation order). */
size_t esmk_gemm_workspace_bytes(int M, int N, int K);
int esmk_gemm_ws(const void* A_bf16, int lda, const void* W_packed, int M, int N, int K, int epilogue,
                 const float* bias, float alpha, void* out, void* workspace, size_t workspace_bytes,
                 esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Small-batch forms (a few hundred to ~1000 tokens: every weight byte is read once from HBM and launches are the cost).
 *
 * esmk_gemm_lnfold: out = epilogue(LayerNorm(x) . W^T) with the LayerNorm FOLDED into the projection - the fused
 * `layernorm_qkv` / `swiglu_ln_ffn` / geometric `s_norm + proj` of the reference (esm/layers/attention.py:26-28,49;
 * esm/layers/blocks.py:30-33; esm/layers/geom_attention.py:65-72).  LayerNorm is affine per row, so
 *     LN(x) . W^T = rstd_t * (x . W'^T - mean_t * u) + v,   W' = W . diag(gamma), u = W' . 1, v = W . beta
 * xb = bf16 copy of the fp32 row x (rounded BEFORE normalisation, as the reference's bf16 path holds its stream);
 * stats f32 [M][parts][2] = per-row partial (sum, sum of squares) of the FP32 row over d columns, as written by
 * esmk_gemm_residual_stats (parts = esmk_res_stats_parts(d)) or esmk_row_stats (parts = 1); the kernel reduces them
 * to mean / rstd in its prologue and applies them, u and v in its epilogue.  W_folded: esmk_pack_weight of the fp32 W'
 * written by esmk_fold_layernorm (mode SWIGLU for ESMK_EPI_SWIGLU); u, v f32 [N] in the reference's row order
 * (v may be NULL: no beta).  Epilogues: ESMK_EPI_BF16, ESMK_EPI_F32, ESMK_EPI_SWIGLU.  d %% 64 == 0.
 *
 * esmk_gemm_residual_stats: x_out = x_in + alpha * (A . W^T + bias) (blocks.py:146-155; x_out may equal x_in),
 * xb (optional) = bf16(x_out), stats (optional) = the partial sums above for x_out.  No split-K, no workspace.
 * esmk_row_stats: stats (parts = 1) and xb of a stream no such epilogue has produced (the embedding output).
 * ------------------------------------------------------------------------------------------ */
int esmk_res_stats_parts(int N);
int esmk_row_stats(const float* x, int M, int d, float* stats, void* xb_bf16, esmk_stream_t stream);
int esmk_fold_layernorm(const float* w, const float* gamma, const float* beta, int N, int K, float* w_folded, float* u,
                        float* v, esmk_stream_t stream);
int esmk_gemm_lnfold(const void* xb_bf16, int lda, const float* stats, int parts, float eps, const void* W_folded_packed,
                     const float* u, const float* v, int M, int N, int d, int epilogue, void* out, esmk_stream_t stream);
int esmk_gemm_residual_stats(const void* A_bf16, int lda, const void* W_packed, int M, int N, int K, const float* bias,
                             float alpha, const float* x_in, float* x_out, void* xb_bf16, float* stats,
                             esmk_stream_t stream);

/* Allocation-free form of esmk_gemm_ws.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * allocate device memory around the call - code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  Same operation, same kernels and bit-identical results as esmk_gemm_ws with a workspace of
 * esmk_gemm_workspace_bytes(M, N, K) bytes, but the split-K partials live in a fixed device array of
 * ESMK_GEMM_NOALLOC_MAX_BYTES that the library holds on every GPU it runs on until the process ends.
 *   - a shape whose partials exceed the array is rejected with ESMK_ERR_WORKSPACE (with today's split rule no
 *     shape needs more than 25.2 MB); a shape that is not split (esmk_gemm_workspace_bytes == 0) runs as esmk_gemm.
 *   - calls no hipMalloc / hipFree and no allocating library routine; the first call on a device looks up the
 *     array's address (hipGetSymbolAddress) - make one call per device before capturing a graph.
 *   - all calls on one GPU share that GPU's array: two calls that overlap in time on one GPU (different streams or
 *     threads) race on the partials.  Calls on one stream, or ordered by events, are safe.  Nothing is carried
 *     between calls (every partial read is written first by the same call). */
#define ESMK_GEMM_NOALLOC_MAX_BYTES (32u << 20)
int esmk_gemm_noalloc(const void* A_bf16, int lda, const void* W_packed, int M, int N, int K, int epilogue,
                      const float* bias, float alpha, void* out, esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * QK-LayerNorm + rotary, in place on the packed qkv buffer.
 * Replaces: q_ln / k_ln over the FULL d_model width (bias-free) followed by NeoX rotary on every
 * 64-wide head (reference attention.py:31-33,40-46,51-55; rotary.py:36-65,198-230) and, for the
 * packed layout, flash-attn's Triton apply_rotary (rotary.py:246-261).
 * qkv bf16 [M, 3d]: columns [0,d)=q, [d,2d)=k, [2d,3d)=v (attention.py:50), head h = columns h*64..h*64+63.
 * positions int32 [M]: rotary position of each token (column index in the reference's [B,L] row).  PRECONDITION (also of
 * esmk_k_prep and esmk_attn_varlen_qfused): 0 <= positions[i] < rope_len - the values are device data the host cannot
 * see, a larger one reads the table's last row.  esmk_trunk_forward checks esmk_batch.max_position against rope_len.
 * rope_cos / rope_sin f32 [rope_len, 32] (angles computed in fp32 as rotary.py:136-181 does).
 * q is additionally multiplied by q_scale (softmax scale * log2(e)) so the attention kernel uses exp2.
 * gamma_q = gamma_k = NULL: the qk_layernorm=False block variant (attention.py:34-38: q_ln / k_ln are Identity) -
 * rotary and the q scale only.
 * ------------------------------------------------------------------------------------------ */
int esmk_qk_prep(void* qkv_bf16, const float* gamma_q, const float* gamma_k, const int32_t* positions,
                 const float* rope_cos, const float* rope_sin, int rope_len, int M, int d, int n_heads,
                 float q_scale, float eps, esmk_stream_t stream);
/* Same with biased q_ln / k_ln: the bias=True block variant keeps QK-LayerNorm as nn.LayerNorm(d_model, bias=True)
 * (reference attention.py:31-33 with blocks.py:81).  beta_q / beta_k f32 [d], both or neither (NULL = esmk_qk_prep);
 * q's beta is scaled by q_scale like the rest of q. */
int esmk_qk_prep_ex(void* qkv_bf16, const float* gamma_q, const float* gamma_k, const float* beta_q,
                    const float* beta_k, const int32_t* positions, const float* rope_cos, const float* rope_sin,
                    int rope_len, int M, int d, int n_heads, float q_scale, float eps, esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Varlen packed flash attention (non-causal, block-diagonal by segment), head dim 64.
 * Replaces: F.scaled_dot_product_attention with the sequence_id equality mask (reference
 * attention.py:65-78) and flash_attn_varlen_qkvpacked_func(qkv[N,3,H,64], cu_seqlens, max_seqlen,
 * softmax_scale) -> [N,H,64] (attention.py:117-119).
 * qkv as produced by esmk_qk_prep (q pre-scaled, so scores feed exp2 directly).
 * cu_seqlens int32 [n_segments+1] (device).  qblk_map int32 [2*n_qblk] (device) from esmk_attn_plan.
 * ctx bf16 [M, d].
 * esmk_attn_plan is HOST-side: it fills qblk_map_host with (segment, first query row in segment)
 * pairs, 128 query rows per block, and returns the number of blocks (or <0).  Pass NULL to count.
 * ------------------------------------------------------------------------------------------ */
#ifndef ESMK_ATTN_QBLK
#define ESMK_ATTN_QBLK 128
#endif
int esmk_attn_plan(const int32_t* cu_seqlens_host, int n_segments, int32_t* qblk_map_host, int capacity);
int esmk_attn_varlen(const void* qkv_bf16, void* ctx_bf16, const int32_t* cu_seqlens, const int32_t* qblk_map,
                     int n_qblk, int M, int d, int n_heads, esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The q side of QK-LayerNorm + rotary fused into the attention kernel (q makes no extra trip through HBM).
 * Replaces, together: q_ln / k_ln (reference attention.py:31-33,51-52), RotaryEmbedding (rotary.py:47-65, attention.py:
 * 53-54) and the attention itself (attention.py:65-78,117-119) - the same results as esmk_gemm + esmk_qk_prep_ex +
 * esmk_attn_varlen up to bf16 rounding, as three calls:
 *   1. esmk_gemm_qkv_stats: qkv = A . W_qkv^T (bf16 [M, 3d], what esmk_gemm(EPI_BF16) writes) and, from the fp32
 *      accumulators of the q columns, qstats f32 [M][parts][2] = (sum, sum of squares) over d/parts columns each.
 *      parts = esmk_qkv_stats_parts(M, d) <= 32: one part per wave half-tile of the 288- / 192-channel-tile kernels, or
 *      per 64 (d <= 2048) / 128 channels of the 256-tile kernel (any d % 64 == 0, the q / k boundary may lie inside a
 *      tile).  0 = no statistics-emitting kernel takes the shape (small M; d % 64 != 0; d 960 from about 12 k tokens,
 *      where 288-wide tiles that straddle the boundary win the projection): then use the three separate calls.
 *      The qkv it writes is bit-identical to esmk_gemm's.
 *   2. esmk_k_prep: k_ln (gamma_k, beta_k or NULL) + rotary on the k columns only, in place; and, when qstats is given,
 *      the partials reduced to q_mean_rstd f32 [M][2] = (mean, 1/sqrt(var + eps)) of every token's q row (qstats NULL:
 *      k only).  qstats / q_mean_rstd 8-byte aligned.
 *   3. esmk_attn_varlen_qfused: normalises (gamma_q, beta_q or NULL, q_mean_rstd), rotates (positions, rope tables as
 *      in esmk_qk_prep) and scales its query fragments in registers, then runs esmk_attn_varlen's loop.
 *      pad_tail as in esmk_batch.
 * ------------------------------------------------------------------------------------------ */
int esmk_qkv_stats_parts(int M, int d_model);
int esmk_gemm_qkv_stats(const void* A_bf16, int lda, const void* W_qkv_packed, int M, int d_model, void* qkv_bf16,
                        float* qstats, esmk_stream_t stream);
int esmk_k_prep(void* qkv_bf16, const float* gamma_k, const float* beta_k, const int32_t* positions,
                const float* rope_cos, const float* rope_sin, int rope_len, int M, int d, int n_heads, float eps,
                const float* qstats, int parts, float* q_mean_rstd, esmk_stream_t stream);
int esmk_attn_varlen_qfused(const void* qkv_bf16, void* ctx_bf16, const int32_t* cu_seqlens, const int32_t* qblk_map,
                            int n_qblk, int M, int d, int n_heads, int pad_tail, const float* q_mean_rstd,
                            const float* gamma_q, const float* beta_q, const int32_t* positions, const float* rope_cos,
                            const float* rope_sin, int rope_len, float q_scale, esmk_stream_t stream);

/* Attention of the function-token decoder's stack (reference esm/models/function_decoder.py:92-103,172-174:
 * TransformerStack(d 1024, 8 heads => head dim 128, qk_layernorm=False) on sequences of exactly 64 tokens with
 * sequence_id=None, i.e. attention.py:55-78 without mask).  qkv bf16 [n_sequences*64, 3d] straight from the QKV GEMM
 * (bias included); rotary over the full 128-wide head is applied while staging (rope tables f32 [>=64, 64] computed as
 * rotary.py:136-177 with dim = 128); ctx bf16 [n_sequences*64, d]. */
int esmk_attn_l64_d128(const void* qkv_bf16, void* ctx_bf16, const float* rope_cos, const float* rope_sin,
                       int n_sequences, int d, int n_heads, esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Embedding gather: out f32 [M, d] = table[tokens].   Replaces nn.Embedding (esmc.py:63,137).
 * ------------------------------------------------------------------------------------------ */
int esmk_embed(const int64_t* tokens, const float* table, int vocab, float* out, int M, int d,
               esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ESM3 input encoder: out f32 [M,d] = sum of the 8 track embeddings.
 * Replaces EncodeInputs.forward + rbf (reference esm/models/esm3.py:92-147, esm/utils/misc.py:105-115).
 * function_tokens int64 [M,8], residue_tokens int64 [M,16]; function tables are 8 x [260, d/8]
 * concatenated; padding index 0 contributes nothing for function/residue tracks.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const float* sequence_embed;  /* [64, d] */
    const float* plddt_w;         /* [d, 16] */
    const float* plddt_b;         /* [d] */
    const float* res_plddt_w;     /* [d, 16] */
    const float* res_plddt_b;     /* [d] */
    const float* structure_embed; /* [4101, d] */
    const float* ss8_embed;       /* [11, d] */
    const float* sasa_embed;      /* [19, d] */
    const float* function_embed;  /* [8, 260, d/8] */
    const float* residue_embed;   /* [1478, d] */
} esmk_esm3_encoder;
int esmk_esm3_encode(const esmk_esm3_encoder* enc, const int64_t* sequence, const int64_t* structure,
                     const float* average_plddt, const float* per_res_plddt, const int64_t* ss8,
                     const int64_t* sasa, const int64_t* function_tokens, const int64_t* residue_tokens,
                     float* out, int M, int d, esmk_stream_t stream);

/* The same with ESM3.forward's default filling and structure-token fix-ups (reference esm/models/esm3.py:317-359) done
 * in the kernel, so that the Python forward launches one kernel instead of a chain of fill / expand / masked_fill
 * operations and never synchronises: a NULL track takes the reference's default (sequence MASK 32, structure MASK 4096,
 * ss8 / sasa / function / residue pad 0, average_plddt 1, per_res_plddt 0); a track with rows == 1 is broadcast over
 * the B batch rows (esm3.py builds defaults with batch 1); structure tokens: -1 -> MASK, then forced to BOS 4098 /
 * PAD 4099 / EOS 4097 / chain-break 4100 where the sequence track holds BOS 0 / PAD 1 / EOS 2 / chain-break 31.
 * rows[i]: batch rows of track i in the order sequence, structure, ss8, sasa, function_tokens, residue_tokens,
 * average_plddt, per_res_plddt (ignored for NULL tracks).  err_count (optional, device int32): += ids outside their
 * table (they are clamped), where the reference's embedding lookup would fail. */
typedef struct {
    const int64_t* sequence; const int64_t* structure; const int64_t* ss8; const int64_t* sasa; /* [rows, L] */
    const int64_t* function_tokens;  /* [rows, L, 8] */
    const int64_t* residue_tokens;   /* [rows, L, 16] */
    const float* average_plddt; const float* per_res_plddt; /* [rows, L] */
    int rows[8];
} esmk_esm3_inputs;
int esmk_esm3_encode_ex(const esmk_esm3_encoder* enc, const esmk_esm3_inputs* in, float* out, int B, int L, int d,
                        int32_t* err_count, esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Backbone frames.  Replaces build_affine3d_from_coordinates (reference
 * esm/utils/structure/affine3d.py:494-560): Gram-Schmidt frames from (N, CA, C), eps 1e-10,
 * "black hole" frame for residues without coordinates, identity rotation for rows with none.
 * coords f32 [B, L, 3, 3]; affine f32 [B*L, 12] = [rot row-major 9 | trans 3]; mask u8 [B*L].
 * ------------------------------------------------------------------------------------------ */
int esmk_frames(const float* coords, float* affine, uint8_t* mask, int B, int L, esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Geometric attention core (ESM3 block 0).  Replaces GeometricReasoningOriginalImpl.forward steps
 * (4)-(10) (reference esm/layers/geom_attention.py:52-63,74-146): rotate/translate the projected
 * 3-vectors into the global frame, scores = rot.dot*softplus(w_r)/sqrt3 - |dist|*softplus(w_d)/sqrt3
 * + bias, softmax over the keys of the same batch row, P.V, inverse rotation, zero frameless rows.
 * proj f32 [M, 15*H] (s_norm + proj GEMM output).  row_cu int32 [n_rows+1]: token ranges of the
 * reference's batch rows.  seq_id / chain_id int32 [M].  out bf16 [M, ld_out], columns [0,3H)
 * written, [3H, ld_out) zeroed (ld_out = round_up(3H,64), feeds esmk_gemm).
 * scratch: f32, esmk_geom_scratch_bytes(M, H) bytes.
 * ------------------------------------------------------------------------------------------ */
size_t esmk_geom_scratch_bytes(int M, int v_heads);
int esmk_geom_attn(const float* proj, const float* affine, const uint8_t* affine_mask, const int32_t* seq_id,
                   const int32_t* chain_id, const int32_t* row_cu, int n_rows, int max_row_len,
                   const float* dist_scale, const float* rot_scale, void* out_bf16, int ld_out, void* scratch,
                   int M, int v_heads, esmk_stream_t stream);
/* Same with mask_and_zero_frameless selectable (geom_attention.py:36,145-146).  0 is the VQ-VAE structure encoder's
 * setting (GeometricEncoderStack, vqvae.py:142-159): rows without a frame are not zeroed and a query whose keys are
 * all masked gets the reference's uniform softmax over its row.  When max_row_len <= 16 (kNN neighbourhoods) one fused
 * kernel does the frame transforms and the attention: no n_rows limit, and scratch is unused (may be NULL). */
int esmk_geom_attn_ex(const float* proj, const float* affine, const uint8_t* affine_mask, const int32_t* seq_id,
                      const int32_t* chain_id, const int32_t* row_cu, int n_rows, int max_row_len,
                      const float* dist_scale, const float* rot_scale, void* out_bf16, int ld_out, void* scratch,
                      int M, int v_heads, int mask_and_zero_frameless, esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Whole trunk in one call.  Replaces TransformerStack.forward (reference
 * esm/layers/transformer_stack.py:64-94) = n_layers x UnifiedTransformerBlock.forward
 * (blocks.py:146-157) + the final bias-free LayerNorm.
 * x f32 [M, d] is the residual stream: input embeddings in, PRE-norm output out.
 * post_norm f32 [M, d] receives norm(x).  hiddens (optional, may be NULL) f32 [n_layers, M, d].
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const float* ln1_g; const float* ln1_b; /* attn.layernorm_qkv.0 */
    const void* w_qkv;                      /* packed [3d, d] */
    const float* qln_g; const float* kln_g; /* attn.q_ln / attn.k_ln */
    const void* w_o;                        /* packed [d, d] */
    const float* ln2_g; const float* ln2_b; /* ffn.0 */
    const void* w_ffn1;                     /* packed SWIGLU [2h, d] */
    const void* w_ffn2;                     /* packed [d, h] */
    /* geometric attention; geom_ln_g == NULL when the block has none */
    const float* geom_ln_g;                 /* geom_attn.s_norm.weight */
    const void* w_geom_proj;                /* packed [15*v_heads, d] */
    const void* w_geom_out;                 /* packed [d, 3*v_heads] */
    const float* geom_dist_scale; const float* geom_rot_scale; /* [v_heads] */
    /* LayerNorm-folded copies of the three projections that follow a LayerNorm (esmk_fold_layernorm + esmk_pack_weight),
     * used by the small-batch path (esmk_trunk.small_tokens); all NULL = that path is off for this layer.
     * w_*_f = packed W . diag(gamma), *_u[n] = sum_k bf16(W'[n,k]), *_v[n] = sum_k W[n,k] beta[k] (reference row order) */
    const void* w_qkv_f; const float* qkv_u; const float* qkv_v;
    const void* w_ffn1_f; const float* ffn1_u; const float* ffn1_v;       /* w_ffn1_f packed SWIGLU */
    const void* w_geom_proj_f; const float* geom_u;                       /* s_norm has no bias: no v */
} esmk_layer;

typedef struct {
    int d_model, n_heads, n_layers, ffn_hidden, v_heads;
    float residual_scale;          /* s = sqrt(n_layers/36) */
    const esmk_layer* layers;      /* HOST array of n_layers */
    const float* final_norm_g;     /* transformer.norm.weight */
    const float* rope_cos; const float* rope_sin; int rope_len;
    /* optional profiling probes (NULL = off): arrays of n_layers events (esmk_event_create) recorded on `stream`
     * immediately before / after each layer's FFN up-projection GEMM, the dominant kernel of the forward. */
    void* const* probe_start; void* const* probe_stop;
    /* Batches of at most this many tokens run the small-batch path when every layer carries its folded weights:
     * 6 launches per block (LayerNorm-folded projections esmk_gemm_lnfold, residual epilogues
     * esmk_gemm_residual_stats), no split-K partials, no add+LayerNorm passes.  0 = never. */
    int small_tokens;
} esmk_trunk;

typedef struct {
    int n_tokens, n_segments;
    const int32_t* cu_seqlens;     /* device [n_segments+1] */
    const int32_t* positions;      /* device [n_tokens] */
    const int32_t* qblk_map; int n_qblk; /* device, from esmk_attn_plan */
    /* geometric attention inputs (may be NULL when no block has geometric attention) */
    const float* affine; const uint8_t* affine_mask;
    const int32_t* seq_id; const int32_t* chain_id;
    const int32_t* row_cu; int n_rows; int max_row_len;
    /* 1: the layout of esmk_segments_from_tokens - segments come in (valid prefix, padding) pairs per padded row and
     * qblk_map holds, per row, blocks (segment 2b, q0) for q0 = 0, 128, ... < L; attention output of padding rows is 0 */
    int pad_tail;
    /* largest value in `positions` (host-side knowledge of whoever built them).  esmk_trunk_forward rejects a batch
     * whose max_position >= esmk_trunk.rope_len with ESMK_ERR_INVALID_ARG instead of reading past the rotary tables. */
    int max_position;
} esmk_batch;

size_t esmk_trunk_workspace_bytes(const esmk_trunk* t, int n_tokens);
int esmk_trunk_forward(const esmk_trunk* t, const esmk_batch* b, float* x, float* post_norm, float* hiddens,
                       void* workspace, size_t workspace_bytes, esmk_stream_t stream);
/* Same, capturing ONE block's output only: hidden_layer in [0, n_layers) -> `hiddens` is f32 [M, d] and receives the
 * output of that block (LogitsConfig.ith_hidden_layer, reference esm/sdk/api.py:397, esm/models/esmc.py:213-218:
 * the reference stacks all layers and slices; here the other layers are never materialised).  hidden_layer = -1:
 * all layers, as esmk_trunk_forward. */
int esmk_trunk_forward_ex(const esmk_trunk* t, const esmk_batch* b, float* x, float* post_norm, float* hiddens,
                          int hidden_layer, void* workspace, size_t workspace_bytes, esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Batch description built ON THE DEVICE for a right-padded [B, L] token matrix, so that the Python forward needs no
 * host synchronisation.  Replaces `sequence_id = sequence_tokens != pad` + `seqlens = mask.sum(-1)` +
 * `cu_seqlens = pad(cumsum(seqlens))` (reference esm/models/esmc.py:133-135, esm/layers/attention.py:99-101) for the
 * padded layout: all B*L positions stay in the packed buffer (n_tokens = B*L is known on the host), row b becomes
 * segment 2b = its valid prefix and segment 2b+1 = its padding:
 *     cu_seqlens int32 [2B+1] = {0, len_0, L, L+len_1, 2L, ..., B*L}
 * mask u8 [B, L] (1 = real token) or NULL (then mask = tokens != pad_id).  Use with esmk_batch.pad_tail = 1.
 * The valid prefix is the LEADING RUN of real tokens: whatever follows the first padding position counts as padding.
 * err_count int32 [1] (optional, device): incremented by the number of tokens outside [0, vocab) and of real tokens
 * that follow a padding position (rows that are not right-padded; esmk_compact_tokens below handles those rows
 * exactly) - the caller reads it when it synchronises anyway.
 * ------------------------------------------------------------------------------------------ */
int esmk_segments_from_tokens(const int64_t* tokens, const uint8_t* mask, int B, int L, int64_t pad_id, int vocab,
                              int32_t* cu_seqlens, int32_t* err_count, esmk_stream_t stream);

/* The same for ANY padding pattern - what `sequence_id = sequence_tokens != pad` means in the reference's SDPA path
 * (esm/models/esmc.py:133-135 with esm/layers/attention.py:65-72: the real tokens of a row attend to each other wherever
 * they sit, rotary position = column, esm/layers/rotary.py:207-209).  The real tokens of row b are moved to the front of
 * the row's L slots in their original order (nothing moves in a right-padded row), the padding behind them:
 *     cu_seqlens int32 [2B+1]  as above (valid tokens of row b, then its padding slots)
 *     perm       int32 [B*L]   flat source index b*L + column of the token in slot j
 *     positions  int32 [B*L]   its column (esmk_batch.positions)
 *     tokens_out int64 [B*L]   the tokens in slot order (must not alias `tokens`)
 * err_count (optional) counts tokens outside [0, vocab) only.  esmk_unpack_rows is the inverse on the outputs. */
int esmk_compact_tokens(const int64_t* tokens, int B, int L, int64_t pad_id, int vocab, int32_t* cu_seqlens,
                        int32_t* perm, int32_t* positions, int64_t* tokens_out, int32_t* err_count, esmk_stream_t stream);

/* Inverse of esmk_compact_tokens on n_mat result matrices x f32 [n_mat][B*L][w], in place: slot j of row b returns to its
 * source column and padding columns become zero - `pad_input` of the reference's flash path (esm/models/esmc.py:159-164).
 * w % 4 == 0. */
int esmk_unpack_rows(float* x, int n_mat, int B, int L, int w, const int32_t* perm, const int32_t* cu_seqlens,
                     esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * RegressionHead: logits f32 [M, out_dim].  Replaces Linear -> GELU -> LayerNorm -> Linear
 * (reference esm/layers/regression_head.py:17-22).  x_norm f32 [M, d] (post-norm trunk output).
 * workspace: esmk_head_workspace_bytes(M, d).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const void* w0; const float* b0;  /* packed [d, d], [d] */
    const float* ln_g; const float* ln_b;
    const void* w3; const float* b3;  /* packed [out_dim, d], [out_dim] */
    int out_dim;
} esmk_head;
size_t esmk_head_workspace_bytes(int n_tokens, int d_model);
int esmk_head_forward(const esmk_head* h, const float* x_norm, float* logits, int M, int d, void* workspace,
                      size_t workspace_bytes, esmk_stream_t stream);
/* Same head with hidden_dim != d_model (regression_head.py:16; the function decoder's heads: hidden = 4 d,
 * function_decoder.py:104-125).  w0 packed [hidden, d], b0 / ln_g / ln_b [hidden], w3 packed [out_dim, hidden]. */
size_t esmk_head_ex_workspace_bytes(int n_tokens, int d_model, int hidden_dim, int out_dim);
int esmk_head_forward_ex(const esmk_head* h, const float* x, float* logits, int M, int d, int hidden_dim,
                         void* workspace, size_t workspace_bytes, esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Output side: mean over the residues of every segment (LogitsConfig.return_mean_embedding /
 * return_mean_hidden_states, esm/sdk/api.py:391-392; cookbook/tutorials/2_embed.ipynb pools hidden states with
 * torch.mean(dim=-2)).  x f32 [n_tokens, ldx] packed, cu_seqlens device int32 [n_segments+1],
 * out f32 [n_segments, d]; an empty segment gives zeros.
 * ------------------------------------------------------------------------------------------ */
int esmk_segment_mean(const float* x, int ldx, const int32_t* cu_seqlens, int n_segments, float* out, int d,
                      esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sampling step of the iterative decoding loop (esm/utils/generation.py:703-733,786-822 and
 * esm/utils/sampling.py:153-200,293-313 fused into one pass per row).
 *   logits  f32 [R, V], MODIFIED IN PLACE exactly as the reference modifies its logits tensor: entries outside
 *           the top-p nucleus become finfo(float32).min, invalid ids become -inf
 *   valid   u8 [V] (1 = may be sampled) or NULL
 *   noise   f32 [R, V] of Exp(1) draws (what torch.multinomial draws internally: id = argmax(softmax(l/T)/q));
 *           ignored / may be NULL when temperature == 0 (argmax)
 *   tokens, sampling_mask  current ids and 0/1 mask [R] (both or neither): rows with mask 0 keep their token and
 *           report logprob 0, as torch.where(sampling_mask, sampled, tokens) does
 *   outputs [R]: sampled ids, entropy and top probability of softmax(modified logits), log-prob of the chosen id
 * ------------------------------------------------------------------------------------------ */
int esmk_sample_track(float* logits, int R, int V, const uint8_t* valid, float temperature, float top_p,
                      const float* noise, const int64_t* tokens, const uint8_t* sampling_mask, int64_t* sampled,
                      float* entropy, float* logprob, float* top_prob, esmk_stream_t stream);

/* Which positions the sampler may overwrite (esm/utils/sampling.py:321-342): tokens int64 [B, L] -> out u8 [B, L] = 1 except
 * in the first / last column (BOS / EOS), where the token is one of `special_ids` (a HOST array of at most 64 ids; an
 * entry equal to mask_id is ignored) and, with only_sample_masked, where the token is not mask_id. */
int esmk_sampling_mask(const int64_t* tokens, int B, int L, const int64_t* special_ids_host, int n_special,
                       int64_t mask_id, int only_sample_masked, uint8_t* out, esmk_stream_t stream);

/* SASA decoding (esm/utils/sampling.py:260-290; entropy esm/utils/generation.py:631-639): logits f32 [R = B*L, V]
 * (V = 19: 3 special classes, V-4 value bins, one open last bin), edited in place (classes with valid[v] == 0 -> -inf).
 * value[r] = sum over the value bins of softmax(logits)[3 + k] * bin_mid[k] (bin_mid f32 [V-4], device); +inf where the
 * arg-max is the open last bin or sampling_mask[r] == 0 (optional mask); 0 in the first / last column of a row of L.
 * entropy[r] = -sum p log p of softmax(logits), 0 * -inf counted as 0. */
int esmk_sample_sasa(float* logits, int R, int V, int L, const uint8_t* valid, const float* bin_mid,
                     const uint8_t* sampling_mask, float* value, float* entropy, esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Output side of the VQ-VAE structure decoder (StructureTokenDecoder.decode, reference
 * esm/models/vqvae.py:365-437).  The decoder stack itself is esmk_trunk_forward (residual_scale = 1,
 * vqvae.py:337-339) and the two RegressionHead-shaped MLPs are esmk_head_forward; these three follow them.
 *
 * esmk_dim6_frames: Dim6RotStructureHead.forward after `self.proj` (esm/layers/structure_proj.py:46-63, with
 *   affine=None and an all-False affine_mask as decode() passes).  proj f32 [M, ldp]: columns 0-2 translation,
 *   3-5 / 6-8 the two direction vectors (the 14 torsion columns are unused).  affine f32 [M, 12] =
 *   [rotation row-major 9 | translation 3] ("tensor7_affine"), bb f32 [M, 3, 3] = N, CA, C ("bb_pred").
 * esmk_categorical_mean: CategoricalMixture(logits, bins=V, start=lo, end=hi).mean() (vqvae.py:114-136):
 *   out[r] = sum_v softmax(logits[r])[v] * (lo + (v + 0.5) (hi - lo) / V).  pLDDT: V = 50, lo = 0, hi = 1.
 * esmk_pairwise_pae: PairwisePredictionHead.forward on the down-projected features (vqvae.py:83-95) restricted to
 *   the 64 predicted-aligned-error bins, fused with compute_predicted_aligned_error and compute_tm
 *   (esm/utils/structure/predicted_aligned_error.py:37-66); no [B, L, L, *] intermediate is written.
 *     qk      f32 [B*L, 128]  = downproject(x): columns 0-63 "q", 64-127 "k" (vqvae.py:79,83)
 *     w1      bf16 [128, 128] = linear1.weight (row-major, 16-byte aligned)
 *     w2      bf16 [64, 128]  = linear2.weight[pae rows] * norm.weight (LayerNorm scale folded in)
 *     b2      f32 [64]        = linear2.weight[pae rows] . norm.bias   (LayerNorm shift folded in)
 *     aa_mask u8 [B, L]       = 1 where the token is a structure code (not a special token, vqvae.py:412)
 *     pae     f32 [B, L, L]   pae[b, a, c] pairs (k_a, q_c) as in the reference's [B, L, L] output
 *     ptm     f32 [B]
 * ------------------------------------------------------------------------------------------ */
int esmk_dim6_frames(const float* proj, int ldp, float trans_scale, float* affine, float* bb, int M,
                     esmk_stream_t stream);
int esmk_categorical_mean(const float* logits, int R, int V, float lo, float hi, float* out, esmk_stream_t stream);
int esmk_pairwise_pae(const float* qk, const void* w1_bf16, const void* w2_bf16, const float* b2,
                      const uint8_t* aa_mask, int B, int L, float max_bin, float* pae, float* ptm,
                      esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Input side of the VQ-VAE structure encoder (StructureTokenEncoder.encode, reference esm/models/vqvae.py:179-322).
 * Its GeometricEncoderStack blocks (bias=True, geometric attention only) are composed from esmk_layernorm,
 * esmk_gemm (bias in every epilogue), esmk_geom_attn_ex and esmk_embed; these two surround them.
 *
 * esmk_knn_neighbourhoods: find_knn_edges / knn_graph (vqvae.py:265-289, esm/utils/misc.py:135-174) and the gathers of
 *   encode_local_structure (vqvae.py:224-248).  Per residue: the `knn` (= min(16, L)) nearest residues of its batch
 *   row by CA distance; pairs lacking coordinates rank by 1e6 + 100 |i - j|; padded pairs and pairs of different
 *   sequence_id are +inf; ascending, ties by lower index (torch.sort's order on the CPU).
 *     coords f32 [B, L, 3, 3] (N, CA, C), coord_mask u8 [B, L] (esmk_frames' mask), padding_mask u8 [B, L] (1 = pad),
 *     seq_id int32 [B, L] or NULL, residue_index int64 [B, L] or NULL (= position), affine f32 [B*L, 12]
 *     -> edges int64 [B, L, knn], edge_mask u8 (distance finite), knn_affine f32 [B*L*knn, 12], knn_mask u8,
 *        knn_seq int32, rel_idx int64 [B*L*knn] = clamp(res[j] - res[0], +-rel_bins) + rel_bins + 1 (vqvae.py:45-48)
 * esmk_vq_quantize: z = Linear(d, 128)(x rows, zeroed where keep == 0) (vqvae.py:317-318), then EMACodebook.forward
 *   (esm/layers/codebook.py:57-75): index of the nearest code (first index on ties) and that code.  fp32 throughout.
 *     x f32, row r at x + r * row_stride (row_stride = 16 d selects each neighbourhood's query node, vqvae.py:260-261)
 *     w_t f32 [d, 128] = pre_vq_proj.weight^T, bias f32 [128], codes_t f32 [128, n_codes] = embeddings^T,
 *     code_sq f32 [n_codes] = |code|^2;  z / z_q f32 [M, 128] (either may be NULL), indices int64 [M]
 * ------------------------------------------------------------------------------------------ */
int esmk_knn_neighbourhoods(const float* coords, const uint8_t* coord_mask, const uint8_t* padding_mask,
                            const int32_t* seq_id, const int64_t* residue_index, const float* affine, int B, int L,
                            int knn, int rel_bins, int64_t* edges, uint8_t* edge_mask, float* knn_affine,
                            uint8_t* knn_mask, int32_t* knn_seq, int64_t* rel_idx, esmk_stream_t stream);
int esmk_vq_quantize(const float* x, int row_stride, const uint8_t* keep, const float* w_t, const float* bias,
                     const float* codes_t, const float* code_sq, int M, int d, int d_code, int n_codes, float* z,
                     float* z_q, int64_t* indices, esmk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Timing helpers (thin wrappers over hipEvent*, so that a harness without HIP headers can time kernels on
 * the stream they are launched on).  No counterpart in the reference.
 * ------------------------------------------------------------------------------------------ */
int esmk_event_create(void** event);
int esmk_event_destroy(void* event);
int esmk_event_record(void* event, esmk_stream_t stream);
int esmk_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */

#ifdef __cplusplus
}
#endif
#endif /* ESMK_H */
