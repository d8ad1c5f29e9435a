/* libdidi_hip: C ABI of the MI355X-native DiDi sampler path.
 *
 * The reference (deepvk/didi) has NO native layer and no FFI: its sampler path is Python over
 * torch/transformers.  The entry points below are what a binding for that path binds instead;
 * each cites the reference lines whose device work it replaces (paths relative to the reference
 * root; "HF:" = transformers, which holds the decoder arithmetic -- SURVEY.md section 8c).
 * INTEGRATION.md shows the ctypes stub a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; `stream` is a hipStream_t
 *     passed as void* (NULL = the default stream); all work is enqueued asynchronously on it.
 *   - nothing here allocates, frees or synchronises (didi_fault_read, a host-side helper, is the one call that
 *     waits for the stream); workspaces are sized by the *_bytes queries and passed in.  Calls on different
 *     streams/devices are independent.  Process-wide state is limited to (a) the tuning switches of
 *     didi_set_option (defaults = the production path; nothing reads the environment), (b) a per-device cache of
 *     device attributes (CU count, the > 64 KiB LDS opt-ins), filled on the first call on each device, and (c) the
 *     per-device scratch array of didi_round_argmax_static.
 *   - return value: DIDI_OK (0) or a negative DIDI_ERR_* code.  No exceptions cross the boundary.
 *   - run-time conditions that only the device can see (see DIDI_FAULT_*) are reported through an optional
 *     device-side fault word that the caller zeroes, passes in, and reads back after synchronising.
 *   - "split planes": an fp32 tensor [R, W] held as two bf16 tensors hi,lo [R, W] with
 *     value = hi + lo (hi = RNE bf16 of the value).  lo pointers may be NULL where stated: then
 *     only hi is read/written ("bf16" precision); with both, GEMMs and attention add the hi*lo and
 *     lo*hi cross products ("bf16x3", fp32-class accuracy on the bf16 MFMA pipe).
 *   - gfx950 (MI355X) only.
 */
#ifndef DIDI_HIP_H
#define DIDI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIDI_ABI_VERSION 7

enum {
    DIDI_OK = 0,
    DIDI_ERR_BAD_ARG = -1,     /* NULL where required, size/alignment constraint violated */
    DIDI_ERR_UNSUPPORTED = -2, /* shape outside what the kernels are built for */
    DIDI_ERR_LAUNCH = -3,      /* hipLaunchKernel reported an error */
    DIDI_ERR_WORKSPACE = -4,   /* workspace too small */
    DIDI_ERR_FAULT = -5        /* didi_fault_read: the device-side fault word is non-zero (bits: DIDI_FAULT_*) */
};

/* Device-side fault word (uint32_t in device memory, owned and zeroed by the caller; NULL = not reported).
 * Kernels OR bits into it and never clear it, so one word can watch a whole sampling loop:
 *   DIDI_FAULT_LN_EXCHANGE  a workgroup of didi_gemm_ln gave up waiting for its band mates' row statistics (the launch
 *                           was not co-resident: another kernel or a CU mask held CUs).  The rows it owned are NaN.
 *   DIDI_FAULT_FP16_RANGE   a GEMM epilogue in DIDI_PREC_FP16 produced |value| > 65504 for an fp16 plane.  The planes
 *                           hold the value SATURATED: hi = +-65504, lo = the saturated remainder (conversions run in the hardware's FP16_OVFL mode: no inf is created;
 *                           an inf / NaN already in the data stays); results downstream are not to be trusted
 *                           at fp16 precision: rerun in DIDI_PREC_BF16X3 (bf16 planes have fp32's range).
 * No reference counterpart: the reference computes in fp32 on one stream. */
enum { DIDI_FAULT_LN_EXCHANGE = 1, DIDI_FAULT_FP16_RANGE = 2 };
/* Host-side helper: waits for `stream`, copies the word to *bits_host (may be NULL) and returns DIDI_OK when it is 0,
 * DIDI_ERR_FAULT otherwise.  The ONE call of this library that synchronises. */
int didi_fault_read(const uint32_t* fault, uint32_t* bits_host, void* stream);
const char* didi_fault_string(uint32_t bits);

/* Process-wide tuning switches (A/B measurements, tools/): every default selects the production path and no caller
 * needs them.  They replace the environment variables of ABI <= 3; the library itself never reads the environment.
 * Not thread-safe against concurrent launches. */
enum {
    DIDI_OPT_GEMM_LN = 0,      /* 1: fused projection + LayerNorm where supported (didi_gemm_ln); 0: GEMM then didi_layernorm */
    DIDI_OPT_GEMM_SMALL = 1,   /* 1: split-K small-batch form below DIDI_OPT_SMALL_MAX_M rows */
    DIDI_OPT_GEMM_BIG = 2,     /* 1: 256x192 tile from DIDI_OPT_BIG_MIN_M rows on; 0: 128x128 everywhere */
    DIDI_OPT_GEMM_PP = 3,      /* 1: staggered wave groups (gemm_pp_kernel); 0: in-phase loop (gemm_big_kernel) */
    DIDI_OPT_GEMM_MT = 4,      /* 4 (default) or 2: 64-row wave tiles along M of the in-phase big tile */
    DIDI_OPT_BIG_MIN_M = 5,    /* default 2048 */
    DIDI_OPT_SMALL_MAX_M = 6,  /* default 2048 */
    DIDI_OPT_X2_GEMMS = 7,     /* bit mask of the decoder GEMMs that multiply the weights' remainder plane in an FP16X2 call: bit 0 qkv,
                                  1 attention output, 2 cross-attention query, 3 cross-attention output, 4 FFN up, 5 FFN down; the others
                                  run one product there.  Default 63 (all) */
    DIDI_OPT_FUSED_ATTN = 8,   /* bit 0: the engine runs a layer's query/key/value projections and its causal self-attention as ONE launch
                                  where didi_qkv_self_attention supports the shape; bit 1: its cross-attention query projection and
                                  cross-attention as one where didi_q_cross_attention does; a clear bit: didi_gemm + didi_attention.
                                  Default 3 */
    DIDI_OPT_HI_RESIDUAL = 9,  /* 1: in DIDI_PREC_FP16 (one-product) decoder calls the residual stream between the fused projection +
                                  LayerNorm launches travels as its fp16 hi plane alone (12 bits; the remainder plane is neither written
                                  nor re-joined): half the bytes of those launches' epilogues.  0 (default): hi + lo everywhere */
    DIDI_OPT_LO8_RESIDUAL = 10, /* 1 (default): in DIDI_PREC_FP16 decoder calls (no GEMM there multiplies an activation's
                                  remainder plane) the remainder of the residual stream travels between the fused projection + LayerNorm
                                  launches as BYTES (didi_gemm_ln_lo8: hi + lo8, 14 bits and more): 3 instead of 4 bytes per value and
                                  direction in the launches whose epilogues those bytes bound.  0: 16-bit remainder plane everywhere */
    DIDI_OPT_COUNT = 11
};
int didi_set_option(int key, int value);
int didi_get_option(int key, int* value);

/* Operand precision of GEMMs and attention.  BF16 / BF16X3: bf16 planes, 1 / 3 MFMA products.
 * FP16: fp16 planes (hi = RNE fp16 of the value, lo = fp16 of the remainder), ONE product on the hi
 * planes -- 8x finer operand rounding than BF16 at the same MFMA cost; lo planes are only read where
 * a value is re-joined (residual adds).  Planes written by a call take the format of its precision
 * (fp16 for DIDI_PREC_FP16, bf16 otherwise). */
enum { DIDI_PREC_BF16 = 1, DIDI_PREC_FP16 = 2, DIDI_PREC_BF16X3 = 3,
       /* FP16X2: fp16 planes, TWO products A_hi*W_hi + A_hi*W_lo: the weights enter with their remainder (~22 bits), the
        * activations as one fp16 plane.  Rounding the weights is the error that adds up coherently over the calls of a sampling
        * loop (the same perturbation of the model at every call); the activations' rounding does not.
        * Activation planes, attention and epilogues as DIDI_PREC_FP16; W_lo required, A_lo never read. */
       DIDI_PREC_FP16X2 = 4,
       /* FP16X3: fp16 planes, all three products (A_hi*W_hi + A_hi*W_lo + A_lo*W_hi, ~22 bits): accepted by didi_gemm,
        * didi_gemm_small, didi_gemm_ln, didi_round_argmax and didi_round_ce.  The engine uses it for the FINAL rounding of an
        * fp16-plane loop (once per batch) and the K/V cache, the Python side for materialised logits; didi_attention and
        * didi_decoder_forward take it too -- BF16X3's structure (every activation with its remainder plane, three products in
        * every GEMM and both attention matmuls) on fp16 planes -- as the precision of the last calls of a precision schedule. */
       DIDI_PREC_FP16X3 = 5 };
enum { DIDI_FMT_BF16 = 0, DIDI_FMT_FP16 = 1 }; /* element format of split planes */

/* ABI version / error text (no reference counterpart). */
int didi_version(void);
const char* didi_error_string(int code);

/* ---------------------------------------------------------------- elementwise / row kernels */

/* fp32 [n] -> split planes.  Used once per weight matrix at load time and for the encoder
 * context.  (No reference counterpart: the reference keeps fp32 everywhere.)  n % 4 == 0. */
int didi_split_bf16(const float* src, uint16_t* hi, uint16_t* lo, size_t n, void* stream);
/* the same into fp16 planes (DIDI_PREC_FP16 operands) */
int didi_split_fp16(const float* src, uint16_t* hi, uint16_t* lo, size_t n, void* stream);

/* Standard-normal noise, Philox4x32-10 + Box-Muller, counter-based: element i of the call with
 * (seed, offset) depends only on (seed, offset + i/4).  Replaces `noise.normal_(0, 1)`
 * (src/sampling.py:54,74,79) and `torch.randn_like` (:13) in throughput mode; parity tests inject
 * noise tensors instead.  n % 4 == 0. */
int didi_philox_normal(float* out, size_t n, uint64_t seed, uint64_t offset, void* stream);

/* Fused sampler step + decoder prologue: ONE pass over the [M, H] state.
 *   mode DIDI_STEP_INIT : x = ce * E                                   (sampling.py:13 [+ :42])
 *   mode DIDI_STEP_DDPM : x = cx0 * X0 + ce * E                        (sampling.py:53-55,
 *                         = scale_input(x0 + sigma*eps, sigma), cx0 = 1/sqrt(sigma^2+1), ce = sigma*cx0;
 *                         src/diffusion/utils.py:30-31)
 *   mode DIDI_STEP_EULER: x = (X + (X - X0) * ratio) + ce * E           (sampling.py:98-100 of the
 *                         previous call, then utils.py:55-56 churn of this call)
 *   mode DIDI_STEP_NONE : x = X                                         (caller-supplied x_t, DiDi.forward)
 * then, if x_out != NULL, x is stored (fp32) (and cin * x to xin_out if that is not NULL), and
 *   h = LayerNorm(cin * x + time_embeds[t] + token_type[0] + pos[s]) -> split planes h_hi, h_lo
 * which is `decoder_inputs_embeds + time_embeds(time_ids)` (src/diffusion/model.py:190-191)
 * followed by BertEmbeddings (HF:models/bert/modeling_bert.py:53-108; eps from the config).
 * E == NULL with ce != 0 draws noise in-kernel (Philox, as didi_philox_normal with the same
 * seed/offset).  time_idx (int32 [M/S], per dialog) may be NULL: then every row uses `t`.
 * Row m is position s = m % S of dialog m / S.  H % 4 == 0, H <= 2048.
 * The LayerNorm part is skipped when h_hi == NULL (final Euler update, sampling.py:80). */
enum { DIDI_STEP_NONE = 0, DIDI_STEP_INIT = 1, DIDI_STEP_DDPM = 2, DIDI_STEP_EULER = 3 };
typedef struct didi_step_args {
    int mode;
    int M, S, H;
    const float* X;  /* [M,H] current state (EULER, NONE) */
    const float* X0; /* [M,H] denoiser output x0_hat (DDPM, EULER) */
    const float* E;  /* [M,H] injected noise or NULL */
    float cx0, ce, ratio, cin;
    uint64_t seed, offset; /* in-kernel Philox when E == NULL && ce != 0 */
    float* x_out;          /* [M,H] or NULL; may alias X or X0 */
    float* xin_out;        /* [M,H] or NULL: receives cin * x, the denoiser input (tests/trace) */
    const float* time_embeds; /* [T+1, H] */
    int t;                    /* row of time_embeds when time_idx == NULL */
    const int32_t* time_idx;  /* [M/S] or NULL */
    const float* pos_emb;     /* [>=S, H] decoder.embeddings.position_embeddings */
    const float* tok_type0;   /* [H] decoder.embeddings.token_type_embeddings row 0 */
    const float* ln_gamma;    /* [H] */
    const float* ln_beta;     /* [H] */
    float ln_eps;
    uint16_t* h_hi; /* [M,H] or NULL to skip the embedding/LayerNorm part */
    uint16_t* h_lo; /* [M,H] */
    int plane_format; /* DIDI_FMT_* of h_hi / h_lo */
    int act_dither;   /* fp16 planes: rounding dither of h_hi (see didi_gemm_args.act_dither); 0 = round to nearest */
} didi_step_args;
int didi_step_embed_ln(const didi_step_args* a, void* stream);

/* out[m, :] = table[ids[m], :] * scale (+ pos[m % S, :])  -- token-embedding gather of the context
 * encoder: BlenderbotScaledWordEmbedding + learned positions
 * (HF:models/blenderbot/modeling_blenderbot.py:88-98,497-503) or Bert word embeddings
 * (HF:models/bert/modeling_bert.py:103; token-type/position/LayerNorm via didi_step_embed_ln).
 * ids int64 [M]; table fp32 [V, H]; pos fp32 [>=S, H] or NULL; out fp32 [M, H].  H % 4 == 0. */
int didi_embed_gather(const int64_t* ids, const float* table, float scale, const float* pos, int M, int S, int H,
                      float* out, void* stream);

/* y = LayerNorm(x) over the last dim of fp32 x [M, H] (torch.nn.LayerNorm semantics: biased
 * variance, eps inside the sqrt).  Replaces BertSelfOutput/BertOutput.LayerNorm
 * (HF:models/bert/modeling_bert.py:292,350) -- the dense+bias+residual part is the GEMM epilogue --
 * and the encoder's pre-LN norms.  Outputs: split planes (y_hi required unless y_f32 given;
 * y_lo may be NULL; element format plane_format = DIDI_FMT_*) and/or fp32 y_f32.
 * H % 4 == 0, H <= 2048. */
int didi_layernorm(const float* x, const float* gamma, const float* beta, float eps, int M, int H,
                   uint16_t* y_hi, uint16_t* y_lo, float* y_f32, int plane_format, void* stream);

/* ---------------------------------------------------------------- GEMM (MFMA bf16) */

/* C[M,N] = epilogue( sum_p A_p[M,K] * W_p[N,K]^T ), fp32 accumulation on
 * v_mfma_f32_16x16x32_bf16 / _f16 (every precision goes here; named didi_gemm_bf16 up to ABI 3).  W is an nn.Linear weight as stored ([out, in] = [N, K], K
 * contiguous), pre-split into planes.  Replaces every nn.Linear on the path:
 * Bert query/key/value/dense (HF:models/bert/modeling_bert.py:175-177,244,253-254,290,335,348),
 * the adapter (src/diffusion/model.py:123), `classifier` (src/sampling.py:64,82).
 *   precision : DIDI_PREC_BF16 / DIDI_PREC_FP16 (A_hi*W_hi), DIDI_PREC_FP16X2 (+ A_hi*W_lo) or DIDI_PREC_BF16X3 (+ A_hi*W_lo + A_lo*W_hi)
 *   epilogue  : v = acc (+ bias[n]) ; v = act(v) ; v += residual[m,n]
 *               act: 0 none, 1 exact-erf GELU (hidden_act="gelu"; evaluated to <= 1.5e-7 in BF16X3 and to
 *               <= 3.4e-6 absolute -- a table -- in the one-product precisions, whose outputs are rounded
 *               to 16 bits right after), 2 tanh
 *               residual: fp32 [M, ldr] or split planes (hi+lo), or none
 *   outputs   : any of fp32 C_f32 [M, ldc] and split planes C_hi/C_lo [M, ldc]
 * Constraints: K % 8 == 0, N % 4 == 0, lda/ldw % 8 == 0, ldc % 4 == 0, pointers 16-byte aligned. */
enum { DIDI_ACT_NONE = 0, DIDI_ACT_GELU = 1, DIDI_ACT_TANH = 2,
       DIDI_ACT_RELU = 3,     /* T5 DenseReluDense (HF:models/t5/modeling_t5.py T5DenseActDense) */
       DIDI_ACT_GELU_NEW = 4  /* tanh form 0.5x(1+tanh(sqrt(2/pi)(x+0.044715x^3))): T5 v1.1 gated FFN (T5DenseGatedActDense) */ };
typedef struct didi_gemm_args {
    int M, N, K;
    int precision;
    const uint16_t *A_hi, *A_lo; /* [M, lda] */
    int lda;
    const uint16_t *W_hi, *W_lo; /* [N, ldw] */
    int ldw;
    const float* bias; /* [N] or NULL */
    int act;
    const float* R_f32;          /* residual fp32 [M, ldr] or NULL */
    const uint16_t *R_hi, *R_lo; /* residual split planes [M, ldr] or NULL */
    int ldr;
    float* C_f32;          /* or NULL */
    uint16_t *C_hi, *C_lo; /* or NULL */
    int ldc;
    uint32_t* fault;       /* device fault word (DIDI_FAULT_*) or NULL */
    /* Rounding dither of the fp16 hi plane this launch writes (no reference counterpart: the reference keeps fp32).  0 = round
     * to nearest even.  d in 1..8191 = truncate after adding d / 8192 of an fp16 ulp to the magnitude; a sampling loop passes
     * 1024, 3072, 5120, 7168 in turn to its one-product calls, so that the mean of four consecutive roundings of a static
     * activation is within ulp / 8 of it instead of every call rounding it the same way (errors that are the same at every
     * call add up coherently over the loop).  The remainder plane stays C_lo = rne16(v - C_hi).
     * Honoured by DIDI_PREC_FP16 launches on the 256x192 tile (M >= 2048: the sampler's hot path) and didi_gemm_ln;
     * ignored (plain rounding) by every other form. */
    int act_dither;
} didi_gemm_args;
int didi_gemm(const didi_gemm_args* a, void* stream);

/* y = LayerNorm(A W^T + bias + residual): BertSelfOutput / BertOutput in one launch -- `dense` +
 * dropout(eval) + `LayerNorm(hidden_states + input_tensor)` (HF:models/bert/modeling_bert.py:284-295,342-353).
 * The GEMM keeps its 256x192 tiles; the four workgroups that share a 256-row band exchange exact per-row
 * partial statistics through `stats` (epoch-tagged 8-byte granules: the data is the flag, no counter and no
 * fence), then each normalises its own 192 columns from registers: the pre-norm tensor and the separate
 * LayerNorm pass never touch HBM.
 * Takes a didi_gemm_args with act = NONE, residual PLANES (R_hi, R_lo), bias, and no C_* outputs; writes
 * split planes y_hi / y_lo (format of the precision; may alias the residual planes) and, if not NULL, fp32
 * y_f32, all [M, N] dense.  Supported (didi_gemm_ln_supported != 0): N == 768 (BertConfig hidden size: the
 * decoder of every DiDi config), M >= 2048, K % 64 == 0, any precision; otherwise DIDI_ERR_UNSUPPORTED and the
 * caller runs didi_gemm + didi_layernorm.  `stats`: didi_gemm_ln_stats_bytes(M) bytes, 16-byte aligned: the
 * statistics travel as 8-byte {tag = epoch, value} words.  On entry no word of it may carry `epoch` in its upper half:
 * zero the buffer once (hipMemsetAsync), then pass epoch = 1, 2, 3, ... to successive calls that share it;
 * epoch == 0 makes the call zero the buffer itself and use tag 1 (stand-alone use).
 * A launch never holds more than one workgroup per CU of the CURRENT device (rows are processed in chunks), so the
 * workgroups that wait for each other are co-resident as long as the stream has the device to itself; a concurrent
 * kernel on another stream or a CU-masked stream can break that.  The wait is bounded in any case: rows whose wait ran
 * out come back as NaN AND DIDI_FAULT_LN_EXCHANGE is raised in a->fault (check it with didi_fault_read; callers that
 * share the device between streams switch the fusion off: didi_set_option(DIDI_OPT_GEMM_LN, 0)). */
int didi_gemm_ln_supported(int M, int N, int K, int precision);
size_t didi_gemm_ln_stats_bytes(int M);
int didi_gemm_ln(const didi_gemm_args* a, const float* gamma, const float* beta, float eps, uint16_t* y_hi,
                 uint16_t* y_lo, float* y_f32, void* stats, uint32_t epoch, void* stream);

/* BertSelfAttention in ONE launch (HF:models/bert/modeling_bert.py:164-203: `query`, `key`, `value` Linears :175-177, then
 * eager_attention_forward :111-136 with the causal mask of is_decoder=True :696-702 and no padding mask, as DiDi.forward
 * calls the decoder, src/diffusion/model.py:193-197):
 *   O[b, q, head] = softmax(causal(Q K^T / sqrt(64))) V,   [Q | K | V] = A W^T + bias
 * for dialogs of exactly S = 64 target positions and 64-wide heads (BertConfig defaults: 768 / 12), M = B * 64 rows.  `a` is the
 * didi_gemm_args of the projection: W = the stacked [query; key; value] weight [3H, K] as the engine keeps it, bias [3H],
 * precision DIDI_PREC_FP16 or DIDI_PREC_FP16X2 (the attention itself is one product either way), C_hi = O [M, ldc >= H] fp16
 * plane, nothing else set (no activation, residual, C_f32, C_lo).  A 256-row tile band is four whole dialogs and a 192-column
 * tile one head's Q, K and V, so the projections never leave the chip: same results, bit for bit, as didi_gemm into
 * [M, 3H] planes followed by didi_attention(causal) -- minus 2 x 6 B per row and column of HBM traffic and one launch.
 * scale: softmax scale, 0 = 1/8.  didi_qkv_self_attention_supported: S == 64, H == 64 * heads, M >= 2048, M % 64 == 0,
 * K % 64 == 0; otherwise DIDI_ERR_UNSUPPORTED and the caller runs the two launches. */
int didi_qkv_self_attention_supported(int M, int H, int heads, int S, int K, int precision);
int didi_qkv_self_attention(const didi_gemm_args* a, int heads, int S, float scale, void* stream);

/* BertCrossAttention in ONE launch (HF:models/bert/modeling_bert.py:230-279: the `query` Linear :253, then
 * eager_attention_forward :111-136 over the encoder's keys and values with the additive padding mask of the context,
 * src/diffusion/model.py:193-197 `encoder_attention_mask`), the keys and values being the hoisted per-layer cache of
 * didi_decoder_prepare_context (they do not depend on the denoiser's input):
 *   O[b, q, head] = softmax(Q K_b^T / sqrt(64) + mask_b) V_b,   Q = A W^T + bias
 * for dialogs of exactly S = 64 target positions, C = 64 context positions and 64-wide heads, heads % 3 == 0.  `a` is the
 * didi_gemm_args of the query projection (W [H, K], bias [H], DIDI_PREC_FP16 or DIDI_PREC_FP16X2, C_hi = O [M, ldc >= H] fp16
 * plane, nothing else set).  K_hi / V_hi: fp16 planes [B * 64][ldkv], head h at columns h * 64 ..; key_mask [B][64] int32
 * (0 = padded key) or NULL; key_count [B] (didi_key_counts: contexts are right-padded, keys from the count on are neither
 * loaded nor multiplied) or NULL.  A 256-row tile band is four whole dialogs and a 192-column tile the queries of three
 * heads: the workgroup that holds the tile's accumulators streams the K / V of (dialog, head) through LDS under the
 * arithmetic.  Same results, bit for bit, as didi_gemm followed by didi_attention(key_mask, key_count) -- minus the query
 * plane's round trip and one launch.  didi_q_cross_attention_supported: S == C == 64, H == 64 * heads, heads % 3 == 0,
 * M >= 2048, M % 64 == 0, K % 64 == 0; otherwise DIDI_ERR_UNSUPPORTED and the caller runs the two launches. */
int didi_q_cross_attention_supported(int M, int H, int heads, int S, int C, int K, int precision);
int didi_q_cross_attention(const didi_gemm_args* a, const uint16_t* K_hi, const uint16_t* V_hi, int ldkv, const int32_t* key_mask,
                           const int32_t* key_count, int heads, int S, int C, float scale, void* stream);

/* didi_gemm_ln with the remainder plane of the residual and / or of the output as BYTES (no reference counterpart: the reference
 * keeps fp32 hidden states, HF:models/bert/modeling_bert.py:284-295; this is how the engine carries them between the launches
 * of a one- or two-product fp16 call).  lo_bytes bit 0: a->R_lo points to a byte plane [M, ldr]; bit 1: y_lo is written as a
 * byte plane [M, N].  byte = top byte (sign, 5 exponent, 2 mantissa bits) of fp16(256 * (v - hi)), rounded half away from zero;
 * the residual re-joins as hi + fp16(byte << 8) / 256.  DIDI_PREC_FP16 / DIDI_PREC_FP16X2 only (else DIDI_ERR_UNSUPPORTED);
 * a NULL a->R_lo means "the residual is its hi plane", a NULL y_lo "write no remainder plane" (both also in didi_gemm_ln).
 * lo_bytes = 0 is didi_gemm_ln. */
int didi_gemm_ln_lo8(const didi_gemm_args* a, const float* gamma, const float* beta, float eps, uint16_t* y_hi, void* y_lo,
                     float* y_f32, void* stats, uint32_t epoch, int lo_bytes, void* stream);

/* Small batches (M < 2048 rows: interactive use, a handful of dialogs).  The same arithmetic as didi_gemm (+ an
 * optional LayerNorm over the N output columns: gamma/beta non-NULL, as didi_gemm_ln), organised for latency: K is split
 * over workgroups, the partial sums go through `workspace` (didi_gemm_small_workspace_bytes(M, N, K, precision) bytes,
 * 16-byte aligned; 0 = this form does not apply, use didi_gemm), and one row kernel applies bias, activation,
 * residual, LayerNorm and writes the outputs (ldc == N).  N <= 3072, N % 4 == 0, K % 128 == 0.
 * Also taken, at any M, by weight-gradient shapes of the training step (K >= 8192 with fewer than 256 output tiles):
 * dW = dC^T A has a few dozen 128 x 128 tiles, each walking K = B S rows. */
size_t didi_gemm_small_workspace_bytes(int M, int N, int K, int precision);
int didi_gemm_small(const didi_gemm_args* a, const float* gamma, const float* beta, float eps, void* workspace, void* stream);

/* ids[m] = argmax_n sum_p X_p[m,:] . Wc_p[n,:]  -- the "rounding" step
 * `model.classifier(x_0)` + `logits.argmax(-1)` (src/sampling.py:64,82,32) without ever writing
 * the [M, V] logits.  First (lowest) index wins ties, as torch.argmax on CPU.  Two launches:
 * a GEMM whose epilogue keeps a per-row running (max, index) over a chunk of vocabulary columns,
 * then a reduction over chunks.  `partial` is a caller workspace of didi_round_argmax_bytes(M, V).
 * ids64 (int64, what torch returns) and/or ids32 may be NULL (not both). K % 8 == 0. */
size_t didi_round_argmax_bytes(int M, int V);
int didi_round_argmax(const uint16_t* X_hi, const uint16_t* X_lo, int ldx, const uint16_t* W_hi,
                      const uint16_t* W_lo, int ldw, int M, int V, int K, int precision, void* partial,
                      int64_t* ids64, int32_t* ids32, void* stream);

/* didi_round_argmax for callers that cannot allocate (no counterpart in the reference: it is for code under HIP graph capture,
 * tight serving loops and harnesses that forbid hipMalloc, which have nowhere to get the `partial` workspace from).
 * Same arithmetic, same kernels, same results as didi_round_argmax; the (max, index) partials live in a file-scope device
 * array of didi_round_argmax_static_max_bytes() = 16 MiB that every GPU the library runs on holds until the process ends.
 * Shapes with didi_round_argmax_bytes(M, V) = M * ceil(V / 128) * 8 above that limit are rejected with DIDI_ERR_WORKSPACE
 * (the largest M is floor(2097152 / ceil(V / 128)): M = 16384 fits at V = 8008, M = 5363 at V = 50000).  Allocates nothing,
 * frees nothing, synchronises nothing.  Calls that overlap on one GPU (two streams, two threads) share the array and race:
 * serialise them.  The first call on a device does per-device set-up that must not happen under graph capture: here the
 * array's device address (hipGetSymbolAddress), in the GEMM the device-attribute query and the kernels' LDS opt-ins
 * (hipFuncSetAttribute).  A caller that captures makes one warm-up call of the same shape on each device first. */
size_t didi_round_argmax_static_max_bytes(void);
int didi_round_argmax_static(const uint16_t* X_hi, const uint16_t* X_lo, int ldx, const uint16_t* W_hi, const uint16_t* W_lo,
                             int ldw, int M, int V, int K, int precision, int64_t* ids64, int32_t* ids32, void* stream);

/* The same GEMM with a cross-entropy epilogue: what `DiDi.validation_step` (src/diffusion/model.py:248-254) computes
 * from `sample(..., raw_output=True)` -- `logits.argmax(-1)` and the per-token terms of `calculate_batch_ce`
 * (src/metrics.py:91-93) -- without writing the [M, V] logits:
 *   ids64[m] = argmax_n logits[m, n]         (may be NULL)
 *   nll[m]   = logsumexp_n logits[m, n] - logits[m, targets[m]]      (+inf when targets[m] is outside [0, V))
 * The caller applies the non-pad mask and the means.  `partial`: didi_round_ce_bytes(M, V) bytes. */
size_t didi_round_ce_bytes(int M, int V);
int didi_round_ce(const uint16_t* X_hi, const uint16_t* X_lo, int ldx, const uint16_t* W_hi, const uint16_t* W_lo,
                  int ldw, int M, int V, int K, int precision, const int64_t* targets, void* partial, int64_t* ids64,
                  float* nll, void* stream);

/* lengths[b] = (index of first eos in ids[b, :]) + 1, or S when the row holds none:
 * `truncate_predictions` (src/sampling.py:104-113) without its per-row D2H sync. */
int didi_first_eos(const int64_t* ids, int B, int S, int64_t eos_id, int32_t* lengths, void* stream);

/* T5LayerNorm (HF:models/t5/modeling_t5.py): y = x * rsqrt(mean(x^2) + eps) * gamma, no mean subtraction, no bias.
 * Same outputs and limits as didi_layernorm. */
int didi_rmsnorm(const float* x, const float* gamma, float eps, int M, int H, uint16_t* y_hi, uint16_t* y_lo, float* y_f32,
                 int plane_format, void* stream);

/* y = a * b elementwise -> split planes (T5 v1.1's gated FFN: gelu_new(x wi_0^T) * (x wi_1^T)); n % 4 == 0. */
int didi_mul_split(const float* a, const float* b, size_t n, uint16_t* y_hi, uint16_t* y_lo, int plane_format, void* stream);

/* ---------------------------------------------------------------- attention */

/* O[b, q, head] = softmax(Q K^T / sqrt(D) + mask) V for one layer, all (dialog, head) pairs:
 * `eager_attention_forward` as BertSelfAttention / BertCrossAttention call it
 * (HF:models/bert/modeling_bert.py:111-136,164-203,230-279).
 *   causal != 0 : key j visible to query i iff j <= i  (is_decoder=True, HF:...:696-702); Sk == Sq
 *   key_mask    : int32 [B, Sk], 0 = padded key (encoder_attention_mask, src/diffusion/model.py:196),
 *                 or NULL.  Masked scores take the most negative float, as HF's additive mask: a
 *                 row whose keys are ALL masked yields the uniform average of V (4.30.2 eager
 *                 behaviour), never NaN.
 * Q/K/V/O are split planes addressed as plane[(b*S + s) * ld + head*D + d]; lo planes may be NULL
 * together with precision DIDI_PREC_BF16.  D % 8 == 0, D <= 64; Sq, Sk <= 96 (dialogs are <= 84 tokens,
 * README.md:48-49 of the reference); ld % 8 == 0. */
typedef struct didi_attn_args {
    int B, heads, Sq, Sk, D;
    int precision;
    int causal;
    const int32_t* key_mask;
    const uint16_t *Q_hi, *Q_lo;
    int ldq;
    const uint16_t *K_hi, *K_lo;
    int ldk;
    const uint16_t *V_hi, *V_lo;
    int ldv;
    uint16_t *O_hi, *O_lo;
    int ldo;
    float scale;           /* softmax scale; 0 = 1/sqrt(D) (BERT, Blenderbot).  T5 passes 1: it folds the scale into its init */
    const float* pos_bias; /* [heads][Sq][Sk] fp32 added to the scaled scores before the mask, or NULL: T5's relative
                              position bias (HF:models/t5/modeling_t5.py T5Attention.compute_bias) */
    const int32_t* key_count; /* optional, with key_mask: int32 [B] from didi_key_counts -- key_mask[b] is 1 for exactly the
                              first key_count[b] keys (right padding, commonsense_dataset.py:27-32).  The kernel then neither
                              loads nor multiplies the padded keys (their softmax weight is exactly 0): same result, less
                              traffic.  NULL, or a count of 0 or Sk: the full masked path. */
    int act_dither;        /* DIDI_PREC_FP16 / FP16X2: rounding dither (didi_gemm_args.act_dither) of the probabilities' and the
                              output's fp16 planes; 0 = round to nearest */
} didi_attn_args;
int didi_attention(const didi_attn_args* a, void* stream);
/* counts[b] = number of visible keys of mask[b, :] (int32 [B, Sk]) if they form a prefix, else Sk (no trimming). */
int didi_key_counts(const int32_t* mask, int B, int Sk, int32_t* counts, void* stream);

/* ---------------------------------------------------------------- decoder engine */

typedef struct didi_linear {
    const uint16_t* w_hi; /* [N, K] */
    const uint16_t* w_lo; /* [N, K] (may be NULL if only DIDI_PREC_BF16 is ever used) */
    const float* bias;    /* [N] or NULL */
    int N, K;
} didi_linear;

/* One BertLayer with cross-attention (HF:models/bert/modeling_bert.py:354-416).  qkv and ckv are
 * the row-concatenations [query;key;value] and [key;value] of the reference's separate Linears. */
typedef struct didi_decoder_layer {
    didi_linear qkv;       /* attention.self.{query,key,value}            [3H, H] */
    didi_linear attn_out;  /* attention.output.dense                      [H, H]  */
    const float *ln1_g, *ln1_b;
    didi_linear cq;        /* crossattention.self.query                   [H, H]  */
    didi_linear ckv;       /* crossattention.self.{key,value}             [2H, H] */
    didi_linear cross_out; /* crossattention.output.dense                 [H, H]  */
    const float *ln2_g, *ln2_b;
    didi_linear ffn_up;    /* intermediate.dense                          [I, H]  */
    didi_linear ffn_down;  /* output.dense                                [H, I]  */
    const float *ln3_g, *ln3_b;
} didi_decoder_layer;

typedef struct didi_decoder {
    int H, heads, I, L;
    float ln_eps;
    const float* time_embeds; /* [T+1, H]  DiDi.time_embeds (src/diffusion/model.py:105) */
    const float* pos_emb;     /* [max_pos, H] */
    const float* tok_type0;   /* [H] */
    const float *emb_ln_g, *emb_ln_b;
    const didi_decoder_layer* layers_host; /* HOST array of L layer records */
} didi_decoder;

/* Bytes of workspace didi_decoder_forward needs for a batch of B dialogs, S target and C context
 * positions.  0 on unsupported shapes. */
size_t didi_decoder_workspace_bytes(const didi_decoder* dec, int B, int S, int C);

/* Bytes of ONE plane set of the hoisted cross-attention K/V cache: [L][B*C][2H] bf16. */
size_t didi_decoder_kv_plane_bytes(const didi_decoder* dec, int B, int C);

/* Cross-attention K/V of the (loop-invariant) context, computed ONCE per batch for all layers:
 * kv[l] = ctx @ ckv_l^T + b.  The reference recomputes these in every one of its ~2000 denoiser
 * calls (HF:models/bert/modeling_bert.py:252-254 via a fresh cache per call, :644-649).
 * ctx is fp32 [B*C, H] as returned by DiDi._encode_context (src/diffusion/model.py:165-170).
 * kv_lo may be NULL with DIDI_PREC_BF16. */
int didi_decoder_prepare_context(const didi_decoder* dec, const float* ctx, int B, int C, int precision,
                                 void* workspace, size_t workspace_bytes, uint16_t* kv_hi, uint16_t* kv_lo,
                                 void* stream);

/* One denoiser call = DiDi.forward with a cached context (src/diffusion/model.py:172-198 with
 * context != None) fused with the sampler update that produces its input:
 *   [x <- step(...)]  ->  time-embedding + BertEmbeddings  ->  L x BertLayer  ->  x0_hat.
 * `step` describes the input (see didi_step_embed_ln; its h_hi/h_lo/embedding fields are filled in
 * by the engine from `dec`, only mode/X/X0/E/coefficients/seed/x_out/t/time_idx are read).
 * Outputs: x0_f32 [B*S, H] (required) and, for the rounding kernel, split planes x0_hi/x0_lo
 * (may be NULL).  The pooler (HF:...:680) and the per-call KV cache (:644-649) are dead code on
 * this path and are not computed. */
typedef struct didi_forward_args {
    int B, S, C;
    int precision;
    didi_step_args step;
    const uint16_t *kv_hi, *kv_lo; /* from didi_decoder_prepare_context */
    const int32_t* ctx_mask;       /* [B, C] int32 */
    const int32_t* ctx_count;      /* [B] int32 from didi_key_counts(ctx_mask) or NULL (see didi_attn_args.key_count) */
    void* workspace;
    size_t workspace_bytes;
    float* x0_f32;
    uint16_t *x0_hi, *x0_lo;
    /* optional per-launch timing of the GEMM kernel (bench.py's roofline leg): when prof_events !=
     * NULL, a hipEvent_t from this HOST array is recorded on `stream` before and after every GEMM
     * launch of this call (2 per launch, in launch order) and *prof_used_host is set to the number
     * recorded.  The caller creates/destroys the events and reads them after synchronising. */
    void** prof_events_host;
    int prof_capacity;
    int* prof_used_host;
    uint32_t* fault; /* device fault word (DIDI_FAULT_*) or NULL */
    int act_dither;  /* DIDI_PREC_FP16 calls: rounding dither of every fp16 activation plane this call writes (embeddings
                        LayerNorm, GEMM epilogues, attention probabilities and outputs): didi_gemm_args.act_dither.  0 = none */
} didi_forward_args;
int didi_decoder_forward(const didi_decoder* dec, const didi_forward_args* a, void* stream);

/* The whole reverse-diffusion loop of sample_ddpm / sample_euler (src/sampling.py:38-101) for one
 * batch, enqueued back to back with no host round trip: n_calls denoiser calls driven by the
 * host-side plan rows (didi_amd/schedule.py), the rounding argmax at the end when ids64 != NULL,
 * logits via didi_gemm by the caller otherwise.
 *   euler == 0: ddpm.  plan rows: time_row[k], c0[k] = a_k, c1[k] = b_k        (k = 0..n_calls-1)
 *   euler != 0:        plan rows: time_row[k], c0[k] = churn_k, c1[k] = c_in_k, c2[k] = ratio_k
 *   init_scale: x_T = init_scale * eps_0 (ddpm: already includes scale_input(., sigma_T))
 * Noise: `noise` NULL -> in-kernel Philox (seed, call k uses offset = base_offset + k * M*H/4);
 *        else fp32 [n_draws, M, H] in the reference's draw order (draw 0 = x_T; ddpm: draw k+1
 *        follows call k, n_draws = n_calls; euler: draw k+1 precedes call k, n_draws = n_calls+1).
 * xt_trace: NULL or fp32 [n_calls, M, H]; receives the denoiser input x_t of every call (tests).
 * x_state: fp32 [M, H] scratch (euler state); x0_f32 receives the final x0 estimate
 * (ddpm: x0_hat of the last call; euler: the final Euler update). */
typedef struct didi_sample_args {
    int B, S, C;
    int precision;
    int euler;
    int n_calls;
    const int32_t* time_row_host;
    const float *c0_host, *c1_host, *c2_host;
    float init_scale;
    const float* noise;
    uint64_t seed, base_offset;
    float* xt_trace;
    const uint16_t *kv_hi, *kv_lo;
    const int32_t* ctx_mask;
    const int32_t* ctx_count; /* as didi_forward_args.ctx_count */
    void* workspace;
    size_t workspace_bytes;
    float* x_state;
    float* x0_f32;
    uint16_t *x0_hi, *x0_lo;
    /* rounding (optional): classifier planes [V, H]; ids64 [B*S]; partial workspace */
    const uint16_t *cls_hi, *cls_lo;
    int V;
    void* argmax_partial;
    int64_t* ids64;
    /* optional: instrument n_prof_calls denoiser calls (their indices: prof_calls_host, increasing) as in didi_forward_args;
     * call j of the list records into events [j * cap, (j + 1) * cap) with cap = prof_capacity / n_prof_calls and reports
     * prof_used_host[j].  A precision schedule has one representative call per precision. */
    const int* prof_calls_host;
    int n_prof_calls;
    void** prof_events_host;
    int prof_capacity;
    int* prof_used_host;
    /* optional per-step clamp (BASELINE config 4; NOT in the reference, which never clamps): when
     * clamp_emb != NULL (fp32 [V, H], `emb.weight`), every denoiser output is replaced by the embedding
     * row of its rounded token, x0 <- emb[argmax(classifier(x0))], before the sampler update.  Needs the
     * rounding fields above (cls_*, V, argmax_partial, ids64 as scratch, x0_hi/x0_lo). */
    const float* clamp_emb;
    uint32_t* fault; /* device fault word (DIDI_FAULT_*) or NULL; watched over the whole loop */
    /* precision schedule: NULL, or a HOST array of n_calls DIDI_PREC_* values, the precision of every denoiser call (all of them
     * fp16-plane precisions -- FP16, FP16X2, FP16X3 -- like `precision`, which still names the plane format, the K/V cache and
     * the rounding).  Rounding errors of the early, noise-dominated calls are forgotten within a few hundred steps, and
     * they grow towards the end of the loop where x_t is mostly the previous estimate: the extra MFMA products are spent
     * where the bar needs them. */
    const int* precision_host;
    /* weight variants for the ONE-PRODUCT (FP16) calls of the loop: NULL, or a HOST array of n_variants decoder descriptions that
     * differ from `dec` only in the hi planes of the six per-call GEMM weights, each plane a different rounding of the same fp32
     * weights (ordered dither: the variants' mean is 1 / n_variants of an fp16 ulp from the true weight).  Call k of precision
     * FP16 runs variant k % n_variants, so the weight-rounding error is no longer the same perturbation at every call (which adds
     * up coherently over the loop) but averages out over n_variants consecutive calls.  Calls of the other precisions use `dec`. */
    const didi_decoder* const* dec_variants_host;
    int n_variants;
    /* activation dither for the ONE-PRODUCT (FP16) calls of the loop: != 0 -> call k of precision FP16 rounds its fp16
     * activation planes with didi_forward_args.act_dither = 1024 + 2048 * ((k + k / 4) % 4) (the phase advances by one
     * every four calls relative to the weight variant k % n_variants, so that over 16 calls every weight variant meets every
     * activation offset once).  Calls of the other precisions round to nearest. */
    int act_dither;
} didi_sample_args;
int didi_sample(const didi_decoder* dec, const didi_sample_args* a, void* stream);

/* ---------------------------------------------------------------- training step
 * Forward + loss of DiDi.training_step (src/diffusion/model.py:200-243) and the first backward kernels.  The decoder forward
 * is didi_decoder_forward with a per-dialog time_idx; the two cross-entropy terms are didi_round_ce.  dgrad / wgrad are
 * didi_gemm launches on transposed operand planes (didi_transpose_split):
 *     dA [M,K] = dC [M,N] W [N,K]      -> didi_gemm(A = planes(dC),   W = transpose_split(W)  as [K, N])
 *     dW [N,K] = dC^T [N,M] A [M,K]    -> didi_gemm(A = transpose_split(dC) as [N, M], W = transpose_split(A) as [K, M]) */

/* x_0 = emb + std_0 n0 (get_x0, src/diffusion/utils.py:24-27);  x_t = (x_0 + s_b n1) c_b (get_diffusion_variables, :34-40, with
 * s_b = sigma_{t_b} noise_factor^2 and c_b = 1/sqrt(sigma_{t_b}^2 + 1) per dialog, fp32 [B], computed by the caller from t).
 * n0 / n1: injected fp32 [B,S,H] draws or NULL = in-kernel Philox (seed, counters offset.. and offset + B*S*H/4..). */
int didi_train_noise(const float* emb, const float* n0, const float* n1, float std0, const float* s_b, const float* c_b, int B,
                     int S, int H, uint64_t seed, uint64_t offset, float* x0, float* xt, void* stream);
/* out[b] = { sum m (x0hat-emb)^2, sum m (x0hat-x0)^2, sum m x0^2 } over dialog b (m = target non-pad mask, int64 [B,S]):
 * the masked MSE of model.py:222-227 (both branches of its `t == 1` switch) and the t_T term of :229-231. */
int didi_train_loss_terms(const float* x0hat, const float* emb, const float* x0, const int64_t* mask, int B, int S, int H,
                          float* out, void* stream);
/* dx = m (w_emb (x0hat - emb) + w_x0 (x0hat - x0)): the gradient of that MSE with respect to x0hat. */
int didi_train_mse_backward(const float* x0hat, const float* emb, const float* x0, const int64_t* mask, float w_emb, float w_x0,
                            int M, int H, float* dx, void* stream);
/* LayerNorm backward (torch.nn.LayerNorm over the last dim): dx, dgamma, dbeta from dy, the layer's INPUT x and gamma.
 * workspace: didi_layernorm_backward_bytes(M, H).  H % 4 == 0, H <= 1280.  Deterministic (no atomics). */
size_t didi_layernorm_backward_bytes(int M, int H);
int didi_layernorm_backward(const float* dy, const float* x, const float* gamma, float eps, int M, int H, float* dx, float* dgamma,
                            float* dbeta, void* workspace, void* stream);
/* out[n] = sum_m x[m, n] (bias gradients); workspace: didi_colsum_bytes(M, N). */
size_t didi_colsum_bytes(int M, int N);
int didi_colsum(const float* x, int M, int N, float* out, void* workspace, void* stream);
/* fp32 src [R, C] -> split planes of its transpose [C, ldo], ldo >= R (columns R..ldo-1 zero: pads the GEMM's K to % 8). */
int didi_transpose_split(const float* src, int R, int C, int ldo, uint16_t* hi, uint16_t* lo, int plane_format, void* stream);
/* The same pass with the column sums of src as a by-product (out[c] = sum_r src[r, c]: the bias gradient while dC^T is being
 * prepared for the weight-gradient GEMM).  ldo == R rounded up to 8; workspace of didi_transpose_split_colsum_bytes(R, C). */
size_t didi_transpose_split_colsum_bytes(int R, int C);
int didi_transpose_split_colsum(const float* src, int R, int C, int ldo, uint16_t* hi, uint16_t* lo, int plane_format, float* colsum,
                                void* workspace, void* stream);
/* Everything nn.Linear's backward needs of dC in ONE pass over it (torch autograd makes three: mm(dC, W), mm(dC^T, A), dC.sum(0)):
 * the transposed planes t_* [C, ldo] (the weight-gradient operand), the un-transposed planes p_* [R, C] (the input-gradient
 * operand) and, when colsum != NULL, the column sums (the bias gradient; workspace as above).  C % 4 == 0, else
 * DIDI_ERR_UNSUPPORTED (callers then make the separate passes). */
/* t_hi == NULL (ABI 7): no transposed planes -- what is left of the pass when the weight gradient goes through didi_gemm_wgrad. */
int didi_split_both(const float* src, int R, int C, int ldo, uint16_t* t_hi, uint16_t* t_lo, uint16_t* p_hi, uint16_t* p_lo,
                    int plane_format, float* colsum, void* workspace, void* stream);
/* didi_split_both on a gradient that still has an elementwise backward in front of it, applied as the values are read so that it is
 * not a pass of its own: gelu_u != NULL -- src is the gradient behind BertIntermediate's GELU (HF:models/bert/modeling_bert.py:335-336)
 * and gelu_u [R, C] its pre-activation: src * gelu'(gelu_u), what didi_act_backward(DIDI_ACT_GELU) computes; drop_p > 0 -- src is a
 * gradient arriving at a hidden-dropout site (BertSelfOutput / BertOutput, :293,351): the mask of didi_dropout_add (same seed / offset). */
int didi_split_both_ex(const float* src, int R, int C, int ldo, uint16_t* t_hi, uint16_t* t_lo, uint16_t* p_hi, uint16_t* p_lo,
                       int plane_format, float* colsum, void* workspace, const float* gelu_u, float drop_p, uint64_t seed, uint64_t offset,
                       void* stream);
/* The weight gradient of nn.Linear, dW [N, K] = dC^T A (what `loss.backward()` of src/diffusion/model.py:200-243 computes for every
 * Linear of the decoder; torch: mm(dC.t(), A)), from the split planes of dC [M, ldd] and A [M, lda] AS THEY LIE IN MEMORY: the
 * contraction runs over the ROWS of both, the kernel stages [32 rows][256 | 192 columns] tiles and takes its MFMA fragments out of
 * LDS with transposing reads, so neither dC^T nor A^T is ever written (the two transposing passes per Linear that the
 * didi_gemm_small route needs).  fp32 result, dense [N, K]; split-K over the rows through `workspace`
 * (didi_gemm_wgrad_workspace_bytes(M, N, K, precision) bytes, 16-byte aligned; 0 = this shape / precision is not supported and
 * didi_gemm_wgrad returns DIDI_ERR_UNSUPPORTED: BF16X3, FP16X2 (dC rounded to its hi plane, A exact), FP16X3; M % 32 == 0,
 * M >= 8192; N >= 512, K >= 192, both % 8 == 0; at most 127 output tiles of 256 x 192 and at most half the device's CUs).  ldd, lda % 8 == 0, planes 16-byte aligned. */
size_t didi_gemm_wgrad_workspace_bytes(int M, int N, int K, int precision);
/* The input gradient of nn.Linear, dA [M, lda] = dC [M, N] W [N, K] (+ R_f32 [M, ldr], nullable: the gradient arriving over a skip
 * connection; torch: mm(dC, W)), with W's planes as the FORWARD holds them ([N, ldw] row-major: the contraction runs over W's rows,
 * which the kernel reads through transposing LDS reads -- no transposed copy of the weights per step).  fp32 out, no bias.
 * Supported (didi_gemm_dgrad_supported != 0, else DIDI_ERR_UNSUPPORTED and the caller transposes W and uses didi_gemm): BF16X3,
 * FP16X2 (dC rounded to its hi plane, W exact), FP16X3; M >= 2048; N % 32 == 0; K >= 192, K % 8 == 0 with at most 15 % of the last
 * 192-column tile empty.  ldd, ldw % 8 == 0, lda, ldr % 4 == 0, everything 16-byte aligned. */
int didi_gemm_dgrad_supported(int M, int N, int K, int precision);
int didi_gemm_dgrad(const uint16_t* dC_hi, const uint16_t* dC_lo, int ldd, const uint16_t* W_hi, const uint16_t* W_lo, int ldw, int M, int N,
                    int K, int precision, const float* R_f32, int ldr, float* dA, int lda, void* stream);
int didi_gemm_wgrad(const uint16_t* dC_hi, const uint16_t* dC_lo, int ldd, const uint16_t* A_hi, const uint16_t* A_lo, int lda, int M, int N,
                    int K, int precision, float* dW, void* workspace, void* stream);

/* y = act(x) (fp32, nullable) + its split planes (nullable): the training forward keeps the pre-activation, so GELU is its own
 * pass there.  dx = dy * act'(.) for the two activations on the trained path: DIDI_ACT_GELU takes the PRE-activation (BertIntermediate,
 * HF:models/bert/modeling_bert.py:335-336), DIDI_ACT_TANH the OUTPUT (the adapter's Tanh, src/diffusion/model.py:123). n % 4 == 0. */
int didi_act_forward(int act, const float* x, size_t n, float* y, void* hi, void* lo, int plane_format, void* stream);
int didi_act_backward(int act, const float* dy, const float* saved, size_t n, float* dx, void* stream);
/* y = a x + b y (gradient accumulation where paths meet: residual connections, x_0's consumers). n % 4 == 0. */
int didi_axpby(float a, const float* x, float b, float* y, size_t n, void* stream);
/* calculate_batch_ce backward (src/metrics.py:91-93) on MATERIALISED logits fp32 [M, V], in place:
 * logits[m, :] <- w_m (softmax(logits[m, :]) - onehot(targets[m])), w_m = scale where mask[m] != 0 (scale = 1 / kept tokens), else 0. */
/* nll (nullable) fp32 [M]: the per-token cross-entropy terms (0 on masked rows), from the same pass. */
int didi_ce_backward(float* logits, const int64_t* targets, const int64_t* mask, int M, int V, float scale, float* nll, void* stream);
/* nn.Embedding backward: grad[ids[m], :] += scale * dy[m, :] (float atomics); rows with id == skip_id (padding_idx) get none. */
int didi_embedding_backward(const int64_t* ids, const float* dy, int M, int H, int64_t V, float scale, int64_t skip_id, float* grad,
                            void* stream); /* V: rows of the table; ids outside [0, V) get no gradient */
/* Attention backward without dropout (= didi_attention_train_backward with drop_p = 0): dQ, dK, dV from dO, P recomputed. */
int didi_attention_backward(const float* Q, const float* K, const float* V, const float* dO, const int32_t* key_mask, int B, int heads,
                            int Sq, int Sk, int D, int ldq, int ldk, int ldv, int ldo, int causal, float scale, float* dQ, float* dK,
                            float* dV, void* stream);
/* Attention of the TRAINING step in fp32 (forward and backward share one mask definition for attention-probability dropout,
 * HF:models/bert/modeling_bert.py:129): O = dropout(softmax(Q K^T scale + masks)) V and its backward.  Operands fp32, addressed
 * as didi_attention addresses its planes ([B*S, ld] row-major, head h at columns h*D..); Sq, Sk <= 96, D % 4 == 0, D <= 64
 * (the forward alone: D <= 128 while one head's Q, K, V and scores fit the 160 KiB of LDS -- the context encoders with 80- or
 * 128-wide heads run their attention through it), leading dimensions % 4 == 0, bases 16-byte aligned.  Dropout: element (b, h, i, j) is dropped iff word (j & 3) of
 * Philox4x32-10(counter = offset + (((b heads + h) Sq + i) Sk4 + j) / 4, key = seed) < drop_p 2^32 (Sk4 = Sk rounded up to 4);
 * the backward must be given the forward's (drop_p, seed, offset).  The sampler never runs these. */
typedef struct didi_attn_train_args {
    const float *Q, *K, *V;   /* fp32 operands */
    const float* dO;          /* backward: gradient of O, [B*Sq, ldo] */
    const int32_t* key_mask;  /* [B, Sk] or NULL */
    int B, heads, Sq, Sk, D;
    int ldq, ldk, ldv, ldo;
    int causal;
    float scale;              /* 0 = 1/sqrt(D) */
    float drop_p;             /* 0 = no dropout */
    uint64_t seed, offset;
    float* O;                 /* forward output [B*Sq, ldo] */
    float *dQ, *dK, *dV;      /* backward outputs, laid out like Q, K, V */
    const float* pos_bias;    /* forward only: fp32 [heads, Sq, Sk] added to the scaled scores (T5), or NULL */
} didi_attn_train_args;
int didi_attention_train_forward(const didi_attn_train_args* a, void* stream);
int didi_attention_train_backward(const didi_attn_train_args* a, void* stream);
/* Hidden dropout + residual (BertEmbeddings / BertSelfOutput / BertOutput, HF:models/bert/modeling_bert.py:107,291,349):
 * out = x o mask / (1 - p) + residual (nullable); flat element e is dropped iff word (e & 3) of
 * Philox4x32-10(offset + e / 4, seed) < p 2^32.  Backward = the same call on the gradient with residual = NULL. n % 4 == 0. */
int didi_dropout_add(const float* x, const float* residual, size_t n, float p, uint64_t seed, uint64_t offset, float* out, void* stream);
/* The same followed by the block's LayerNorm in one pass (`LayerNorm(dropout(dense(x)) + input)`, HF:models/bert/modeling_bert.py:291-295,
 * 349-353 in training mode): sum_out [M, H] = x o mask / (1 - p) + residual (kept: the LayerNorm backward's input), then didi_layernorm
 * of it into y_hi / y_lo / y_f32.  H % 4 == 0, H <= 1280. */
int didi_dropout_add_layernorm(const float* x, const float* residual, float p, uint64_t seed, uint64_t offset, float* sum_out,
                               const float* gamma, const float* beta, float eps, int M, int H, uint16_t* y_hi, uint16_t* y_lo, float* y_f32,
                               int plane_format, void* stream);
/* torch.optim.AdamW step (src/diffusion/model.py:136-158: AdamW(lr = 1) x LambdaLR, so `lr` is the schedule's value) on one
 * parameter tensor, gradient pre-scaled by *gscale (device scalar or NULL: the clipping coefficient, configs/train.yaml:45). */
int didi_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int step, const float* gscale, void* stream);
/* out[0] += sum x^2 (global gradient norm; the caller zeroes out). */
int didi_sumsq_accumulate(const float* x, size_t n, float* out, void* stream);

/* Decoder input of the training forward in fp32 (the LayerNorm INPUT, which the backward needs):
 * out[b,s,:] = x_t[b,s,:] + time_embeds[t[b],:] + pos_emb[s,:] + tok0 (src/diffusion/model.py:190-191 + BertEmbeddings). t: int64 [B]. */
int didi_train_embed_sum(const float* x_t, const float* time_embeds, const int64_t* t, const float* pos_emb, const float* tok0, int B, int S,
                         int H, int T_rows, float* out, void* stream); /* T_rows: rows of time_embeds; a t outside [0, T_rows) gives NaN rows */
/* out[b, :] = sum_s x[b, s, :] (fp32 [B,S,H] -> [B,H]): per-dialog sums that scatter into the time-embedding rows. */
int didi_seq_sum(const float* x, int B, int S, int H, float* out, void* stream);
/* Gradient arriving at the target embeddings (src/diffusion/model.py:202-235), one pass:
 * out = m (-w_emb (x0hat - emb) - w_x0 (x0hat - x0) + w_T x0) + c_b[b] dxt + extra, with dxt the gradient of the decoder input x_t,
 * extra (nullable) the gradient of x_0 from cross-entropy(classifier(x_0)); w_T = 2 / ((sigma_T^2 + 1) B S H) or 0 (anchor loss). */
int didi_train_target_backward(const float* x0hat, const float* emb, const float* x0, const int64_t* mask, float w_emb, float w_x0, float w_T,
                               const float* c_b, const float* dxt, const float* extra, int B, int S, int H, float* out, void* stream);

/* hipEvent wrappers for the prof_events_host hook (bench.py): events must live in the HIP runtime
 * instance this library is bound to.  Host-side helpers, no reference counterpart.  Elapsed time
 * may only be read after the caller has synchronised the stream. */
int didi_event_create(void** ev);
int didi_event_elapsed_ms(void* start, void* stop, float* ms);
int didi_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* DIDI_HIP_H */
