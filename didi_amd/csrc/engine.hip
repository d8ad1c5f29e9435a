// Host-side engine: enqueues the kernels of one denoiser call / one whole sampling loop on a HIP
// stream.  No allocation, no synchronisation, no host<->device copies: every per-step scalar
// (time-embedding row, a_t, b_t, churn, ...) is data-independent and comes from the host-side
// plan (didi_amd/schedule.py) as kernel arguments, so the loop never waits on the device.
//
// Workspace layout for M = B*S target rows (all offsets 256-byte aligned):
//   h   planes [2][M][H]   bf16   residual stream (always hi+lo: the residual add reads both)
//   qkv planes [2][M][3H]  bf16   self-attention projections; its first [M][H] doubles as the
//                                 cross-attention query buffer once self-attention is done
//   att planes [2][M][H]   bf16   attention output (concatenated heads)
//   ffn planes [2][M][I]   bf16   GELU(FFN-up)
//   tmp        [M][H]      fp32   pre-LayerNorm sum (dense + bias + residual); unused when the fused
//                                 projection+LayerNorm kernel takes the shape
//   ln_stats                      row-statistic granules of the fused kernel (zeroed once per call, epochs 1..3L)
// didi_decoder_prepare_context uses the front of the same buffer for the context planes.
#include <algorithm>

#include <cstdlib>
#include "common.hpp"

namespace {

struct Carve {
    u16 *h_hi, *h_lo, *qkv_hi, *qkv_lo, *att_hi, *att_lo, *ffn_hi, *ffn_lo;
    float* tmp;
    void* ln_stats;
    size_t ln_stats_bytes;
    void* small_ws;  // split-K partial sums of the small-batch GEMM form (null when no shape of this batch uses it)
    size_t total;
};

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

Carve carve(const didi_decoder* d, int B, int S, char* base) {
    const size_t M = (size_t)B * S, H = d->H, I = d->I;
    Carve c{};
    size_t o = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += al256(bytes); return p; };
    c.h_hi = (u16*)take(M * H * 2);
    c.h_lo = (u16*)take(M * H * 2);
    c.qkv_hi = (u16*)take(M * 3 * H * 2);
    c.qkv_lo = (u16*)take(M * 3 * H * 2);
    c.att_hi = (u16*)take(M * H * 2);
    c.att_lo = (u16*)take(M * H * 2);
    c.ffn_hi = (u16*)take(M * I * 2);
    c.ffn_lo = (u16*)take(M * I * 2);
    c.tmp = (float*)take(M * H * 4);
    c.ln_stats_bytes = didi_gemm_ln_stats_bytes((int)M);
    c.ln_stats = take(c.ln_stats_bytes);
    size_t small = 0;  // the six GEMM shapes of a layer
    const int shapes[4][2] = {{3 * (int)H, (int)H}, {(int)H, (int)H}, {(int)I, (int)H}, {(int)H, (int)I}};
    for (const auto& nk : shapes) small = std::max(small, didi_gemm_small_workspace_bytes((int)M, nk[0], nk[1], DIDI_PREC_BF16X3));
    c.small_ws = small ? take(small) : nullptr;
    c.total = o;
    return c;
}

bool shapes_ok(const didi_decoder* d, int B, int S, int C) {
    if (!d || B <= 0 || S <= 0 || C <= 0 || d->L <= 0 || d->heads <= 0) return false;
    if (d->H % d->heads != 0) return false;
    const int D = d->H / d->heads;
    if (D % 8 != 0 || D > 64) return false;
    if (S > 96 || C > 96) return false;
    if (d->H % 8 != 0 || d->I % 8 != 0 || d->H > 2048) return false;
    return true;
}

struct Prof {
    void** ev; int cap; int used;
    void mark(void* stream) {
        if (ev && used < cap) { (void)hipEventRecord(reinterpret_cast<hipEvent_t>(ev[used]), as_stream(stream)); ++used; }
    }
};

// DIDI_OPT_GEMM_SMALL = 0: A/B switch from the split-K small-batch form back to the plain 128x128 launch
bool small_enabled() { return didi_opt(DIDI_OPT_GEMM_SMALL) != 0; }

int gemm(const didi_linear& w, int precision, int M, const u16* a_hi, const u16* a_lo, int lda, int act, const u16* r_hi,
         const u16* r_lo, int ldr, float* c_f32, u16* c_hi, u16* c_lo, int ldc, void* stream, Prof* prof = nullptr,
         void* small_ws = nullptr, uint32_t* fault = nullptr, int dither = 0) {
    didi_gemm_args g{};
    g.fault = fault;
    g.act_dither = precision == DIDI_PREC_FP16 ? dither : 0;
    g.M = M; g.N = w.N; g.K = w.K; g.precision = precision;
    g.A_hi = a_hi; g.A_lo = a_lo; g.lda = lda;
    g.W_hi = w.w_hi; g.W_lo = w.w_lo; g.ldw = w.K;
    g.bias = w.bias; g.act = act;
    g.R_hi = r_hi; g.R_lo = r_lo; g.ldr = ldr;
    g.C_f32 = c_f32; g.C_hi = c_hi; g.C_lo = c_lo; g.ldc = ldc;
    if (prof) prof->mark(stream);
    const bool small = small_ws && small_enabled() && ldc == w.N && didi_gemm_small_workspace_bytes(M, w.N, w.K, precision);
    const int rc = small ? didi_gemm_small(&g, nullptr, nullptr, 0.f, small_ws, stream) : didi_gemm(&g, stream);
    if (prof) prof->mark(stream);
    return rc;
}

static bool gemm_ln_fuses(const didi_linear& w, int precision, int M) {
    return didi_opt(DIDI_OPT_GEMM_LN) != 0 && didi_gemm_ln_supported(M, w.N, w.K, precision) && w.bias;
}

// output projection + residual + LayerNorm: the fused kernel where it supports the shape (H = 768, M >= 2048),
// else GEMM -> fp32 tmp -> LayerNorm.  `epoch` tags this launch's statistics in the shared `stats` buffer.
int gemm_ln(const didi_linear& w, int precision, int fmt, int M, const u16* a_hi, const u16* a_lo, int lda, const u16* r_hi,
            const u16* r_lo, int ldr, float* tmp, void* stats, uint32_t epoch, const float* gamma, const float* beta, float eps,
            u16* y_hi, u16* y_lo, float* y_f32, void* stream, Prof* prof, void* small_ws, uint32_t* fault, int dither = 0, int hi_only = 0, int lo_bytes = 0) {
    // hi_only: bit 0: the residual is its hi plane, bit 1: no remainder plane is written; lo_bytes: bit 0: r_lo is a byte plane
    // (split4_dither_lo8), bit 1: y_lo is written as one.  Both only where the fused launch takes the shape (the caller checked).
    const bool fuse = didi_opt(DIDI_OPT_GEMM_LN) != 0;  // 0: A/B switch back to GEMM + LayerNorm
    if ((hi_only || lo_bytes) && !gemm_ln_fuses(w, precision, M)) return DIDI_ERR_UNSUPPORTED;  // the other forms read / write 16-bit r_lo, y_lo
    if (fuse && didi_gemm_ln_supported(M, w.N, w.K, precision) && w.bias) {
        didi_gemm_args g{};
        g.fault = fault;
        g.act_dither = precision == DIDI_PREC_FP16 ? dither : 0;
        g.M = M; g.N = w.N; g.K = w.K; g.precision = precision;
        g.A_hi = a_hi; g.A_lo = a_lo; g.lda = lda;
        g.W_hi = w.w_hi; g.W_lo = w.w_lo; g.ldw = w.K;
        g.bias = w.bias; g.act = DIDI_ACT_NONE;
        g.R_hi = r_hi; g.R_lo = (hi_only & 1) ? nullptr : r_lo; g.ldr = ldr;
        if (prof) prof->mark(stream);
        const int rc = didi_gemm_ln_lo8(&g, gamma, beta, eps, y_hi, (hi_only & 2) ? nullptr : y_lo, y_f32, stats, epoch, lo_bytes, stream);
        if (prof) prof->mark(stream);
        return rc;
    }
    if (small_ws && small_enabled() && didi_gemm_small_workspace_bytes(M, w.N, w.K, precision)) {  // LayerNorm rides in the row kernel
        didi_gemm_args g{};
        g.fault = fault;
        g.M = M; g.N = w.N; g.K = w.K; g.precision = precision;
        g.A_hi = a_hi; g.A_lo = a_lo; g.lda = lda;
        g.W_hi = w.w_hi; g.W_lo = w.w_lo; g.ldw = w.K;
        g.bias = w.bias; g.act = DIDI_ACT_NONE;
        g.R_hi = r_hi; g.R_lo = r_lo; g.ldr = ldr;
        g.C_hi = y_hi; g.C_lo = y_lo; g.C_f32 = y_f32; g.ldc = w.N;
        if (prof) prof->mark(stream);
        const int rc = didi_gemm_small(&g, gamma, beta, eps, small_ws, stream);
        if (prof) prof->mark(stream);
        return rc;
    }
    int rc = gemm(w, precision, M, a_hi, a_lo, lda, DIDI_ACT_NONE, r_hi, r_lo, ldr, tmp, nullptr, nullptr, w.N, stream, prof, nullptr, fault);
    if (rc != DIDI_OK) return rc;
    return didi_layernorm(tmp, gamma, beta, eps, M, w.N, y_hi, y_lo, y_f32, fmt, stream);
}

}  // namespace

// ------------------------------------------------------------------ tuning switches (didi_set_option)
static int g_options[DIDI_OPT_COUNT] = {/*GEMM_LN*/ 1, /*GEMM_SMALL*/ 1, /*GEMM_BIG*/ 1, /*GEMM_PP*/ 1, /*GEMM_MT*/ 4,
                                        /*BIG_MIN_M*/ 2048, /*SMALL_MAX_M*/ 2048, /*X2_GEMMS*/ 63, /*FUSED_ATTN*/ 3, /*HI_RESIDUAL*/ 0, /*LO8_RESIDUAL*/ 1};
int didi_opt(int key) { return g_options[key]; }
extern "C" int didi_set_option(int key, int value) {
    DIDI_CHECK_ARG(key >= 0 && key < DIDI_OPT_COUNT);
    if (key == DIDI_OPT_GEMM_MT) DIDI_CHECK_ARG(value == 2 || value == 4);
    if (key == DIDI_OPT_BIG_MIN_M || key == DIDI_OPT_SMALL_MAX_M) DIDI_CHECK_ARG(value >= 0);
    if (key == DIDI_OPT_X2_GEMMS) DIDI_CHECK_ARG(value >= 0 && value <= 63);
    if (key == DIDI_OPT_FUSED_ATTN) DIDI_CHECK_ARG(value >= 0 && value <= 3);
    g_options[key] = value;
    return DIDI_OK;
}
extern "C" int didi_get_option(int key, int* value) {
    DIDI_CHECK_ARG(key >= 0 && key < DIDI_OPT_COUNT && value);
    *value = g_options[key];
    return DIDI_OK;
}

#define TRY(x)                     \
    do {                           \
        int rc_ = (x);             \
        if (rc_ != DIDI_OK) return rc_; \
    } while (0)

extern "C" size_t didi_decoder_workspace_bytes(const didi_decoder* dec, int B, int S, int C) {
    if (!shapes_ok(dec, B, S, C)) return 0;
    const size_t fwd = carve(dec, B, S, nullptr).total;
    const size_t ctx = 2 * al256((size_t)B * C * dec->H * 2);
    return std::max(fwd, ctx);
}

extern "C" size_t didi_decoder_kv_plane_bytes(const didi_decoder* dec, int B, int C) {
    if (!dec || B <= 0 || C <= 0) return 0;
    return (size_t)dec->L * B * C * 2 * dec->H * sizeof(u16);
}

extern "C" int didi_decoder_prepare_context(const didi_decoder* dec, const float* ctx, int B, int C, int precision,
                                            void* workspace, size_t workspace_bytes, uint16_t* kv_hi, uint16_t* kv_lo,
                                            void* stream) {
    DIDI_CHECK_ARG(dec && ctx && workspace && kv_hi && dec->layers_host);
    DIDI_CHECK_ARG(precision == DIDI_PREC_BF16 || precision == DIDI_PREC_FP16 || precision == DIDI_PREC_FP16X2 ||
                   ((precision == DIDI_PREC_BF16X3 || precision == DIDI_PREC_FP16X3) && kv_lo));
    if (!shapes_ok(dec, B, 1, C)) return DIDI_ERR_UNSUPPORTED;
    const size_t Mc = (size_t)B * C, H = dec->H;
    const size_t plane = al256(Mc * H * 2);
    if (workspace_bytes < 2 * plane) return DIDI_ERR_WORKSPACE;
    u16* c_hi = (u16*)workspace;
    u16* c_lo = (u16*)((char*)workspace + plane);
    if (precision == DIDI_PREC_FP16 || precision == DIDI_PREC_FP16X2) TRY(didi_split_fp16(ctx, c_hi, c_lo, Mc * H, stream));
    else TRY(didi_split_bf16(ctx, c_hi, c_lo, Mc * H, stream));
    // fp16-plane precisions with a kv_lo buffer: the cache is computed ONCE per batch, so it gets all three products and its
    // remainder plane (what an fp16x3 call of a precision schedule reads); one-product calls read kv_hi alone
    const bool f16 = precision == DIDI_PREC_FP16 || precision == DIDI_PREC_FP16X2 || precision == DIDI_PREC_FP16X3;
    const int kv_prec = f16 && kv_lo ? DIDI_PREC_FP16X3 : precision;
    const bool want_lo = kv_lo && (precision == DIDI_PREC_BF16X3 || f16);
    for (int l = 0; l < dec->L; ++l) {
        const didi_decoder_layer& ly = dec->layers_host[l];
        DIDI_CHECK_ARG(ly.ckv.N == 2 * dec->H && ly.ckv.K == dec->H);
        const size_t off = (size_t)l * Mc * 2 * H;
        TRY(gemm(ly.ckv, kv_prec, (int)Mc, c_hi, c_lo, (int)H, DIDI_ACT_NONE, nullptr, nullptr, 0, nullptr, kv_hi + off,
                 want_lo ? kv_lo + off : nullptr, (int)(2 * H), stream));
    }
    return DIDI_OK;
}

extern "C" int didi_decoder_forward(const didi_decoder* dec, const didi_forward_args* a, void* stream) {
    DIDI_CHECK_ARG(dec && a && dec->layers_host && a->workspace && a->x0_f32 && a->kv_hi && a->ctx_mask);
    const int prec = a->precision;
    DIDI_CHECK_ARG(prec == DIDI_PREC_BF16 || prec == DIDI_PREC_FP16 || prec == DIDI_PREC_FP16X2 ||
                   ((prec == DIDI_PREC_BF16X3 || prec == DIDI_PREC_FP16X3) && a->kv_lo));
    DIDI_CHECK_ARG(!(a->x0_lo && !a->x0_hi));
    if (!shapes_ok(dec, a->B, a->S, a->C)) return DIDI_ERR_UNSUPPORTED;
    if (a->S > 0 && dec->pos_emb == nullptr) return DIDI_ERR_BAD_ARG;
    const Carve w = carve(dec, a->B, a->S, (char*)a->workspace);
    if (a->workspace_bytes < w.total) return DIDI_ERR_WORKSPACE;
    const int M = a->B * a->S, H = dec->H, I = dec->I, D = H / dec->heads;
    // x3: every GEMM and both attention matmuls multiply all three products and every activation travels with its remainder
    // plane (bf16 planes: BF16X3; fp16 planes: FP16X3).  FP16X2 adds the remainder of the WEIGHTS only; attention has no weights,
    // so there it is one product.
    const bool x3 = prec == DIDI_PREC_BF16X3 || prec == DIDI_PREC_FP16X3;
    const bool f16 = prec == DIDI_PREC_FP16 || prec == DIDI_PREC_FP16X2 || prec == DIDI_PREC_FP16X3;
    const int fmt = f16 ? DIDI_FMT_FP16 : DIDI_FMT_BF16;
    const int attn_prec = prec == DIDI_PREC_FP16X2 ? DIDI_PREC_FP16 : prec;
    // FP16X2 calls: which GEMMs multiply the weights' remainder (DIDI_OPT_X2_GEMMS); the planes are the same either way
    const int x2_mask = didi_opt(DIDI_OPT_X2_GEMMS);
    auto pc = [&](int cls) { return prec == DIDI_PREC_FP16X2 && !((x2_mask >> cls) & 1) ? (int)DIDI_PREC_FP16 : prec; };
    const bool ffn_lo = x3;
    DIDI_CHECK_ARG(a->act_dither >= 0 && a->act_dither < 8192);
    const int dith = prec == DIDI_PREC_FP16 ? a->act_dither : 0;  // one-product calls only: the other precisions carry remainders
    const size_t Mc = (size_t)a->B * a->C;
    Prof prof{a->prof_events_host, a->prof_capacity, 0};

    // sampler step + time embedding + BertEmbeddings
    didi_step_args st = a->step;
    st.M = M; st.S = a->S; st.H = H;
    st.time_embeds = dec->time_embeds; st.pos_emb = dec->pos_emb; st.tok_type0 = dec->tok_type0;
    st.ln_gamma = dec->emb_ln_g; st.ln_beta = dec->emb_ln_b; st.ln_eps = dec->ln_eps;
    st.h_hi = w.h_hi; st.h_lo = w.h_lo; st.plane_format = fmt;
    st.act_dither = dith;
    TRY(didi_step_embed_ln(&st, stream));
    // statistic granules of the fused projection+LayerNorm launches: cleared once, then one epoch per launch
    if (hipMemsetAsync(w.ln_stats, 0, w.ln_stats_bytes, as_stream(stream)) != hipSuccess) return DIDI_ERR_LAUNCH;
    uint32_t epoch = 0;
    // one-product fp16 calls (DIDI_OPT_HI_RESIDUAL): the residual stream between the fused projection + LayerNorm launches travels
    // as its hi plane alone -- those launches are bound by the bytes of their epilogue, and the remainder plane is half of them
    bool hi_res = didi_opt(DIDI_OPT_HI_RESIDUAL) != 0 && prec == DIDI_PREC_FP16;
    for (int l = 0; l < dec->L && hi_res; ++l) {
        const didi_decoder_layer& ly = dec->layers_host[l];
        hi_res = gemm_ln_fuses(ly.attn_out, prec, M) && gemm_ln_fuses(ly.cross_out, prec, M) && gemm_ln_fuses(ly.ffn_down, prec, M);
    }

    // one-product fp16 calls (DIDI_OPT_LO8_RESIDUAL): no GEMM of such a call multiplies an activation's remainder, the
    // residual add alone re-joins it -- so between the fused projection + LayerNorm launches the remainder plane of the residual
    // stream travels as bytes (hi + lo8: 14 bits and more instead of 22; `tmp`, unused by those launches, holds them): 3 instead
    // of 4 bytes per value and direction.  The first launch reads the step kernel's 16-bit plane, the last writes one.
    // One-product calls only: a two-product call could take the byte plane as well (didi_gemm_ln_lo8 does), but "fp16x2" as a precision
    // of its own sits AT the id bar (1 of 1024 positions at f = 10), and the 172 two-product calls of the default schedule would gain 0.1 %.
    bool lo8 = !hi_res && didi_opt(DIDI_OPT_LO8_RESIDUAL) != 0 && prec == DIDI_PREC_FP16;
    for (int l = 0; l < dec->L && lo8; ++l) {
        const didi_decoder_layer& ly = dec->layers_host[l];
        lo8 = gemm_ln_fuses(ly.attn_out, pc(1), M) && gemm_ln_fuses(ly.cross_out, pc(3), M) && gemm_ln_fuses(ly.ffn_down, pc(5), M);
    }
    u16* const h_lo8 = reinterpret_cast<u16*>(w.tmp);

    for (int l = 0; l < dec->L; ++l) {
        const didi_decoder_layer& ly = dec->layers_host[l];
        const bool last = l == dec->L - 1;
        // --- causal self-attention block
        didi_attn_args at{};
        at.act_dither = dith;
        at.B = a->B; at.heads = dec->heads; at.D = D; at.precision = attn_prec;
        at.O_hi = w.att_hi; at.O_lo = x3 ? w.att_lo : nullptr; at.ldo = H;
        // one launch for the projections AND the attention where the shape allows (64 positions per dialog, 64-wide heads, fp16
        // planes, one or two products): the [M, 3H] projection planes never reach memory
        const bool fused_self = (didi_opt(DIDI_OPT_FUSED_ATTN) & 1) != 0 && !x3 && ly.qkv.bias && ly.qkv.N == 3 * H &&
                                didi_opt(DIDI_OPT_GEMM_BIG) != 0 && didi_opt(DIDI_OPT_GEMM_PP) != 0 && M >= didi_opt(DIDI_OPT_BIG_MIN_M) &&
                                didi_qkv_self_attention_supported(M, H, dec->heads, a->S, ly.qkv.K, pc(0));
        if (fused_self) {
            didi_gemm_args g{};
            g.fault = a->fault;
            g.act_dither = dith;
            g.M = M; g.N = 3 * H; g.K = ly.qkv.K; g.precision = pc(0);
            g.A_hi = w.h_hi; g.A_lo = w.h_lo; g.lda = H;
            g.W_hi = ly.qkv.w_hi; g.W_lo = ly.qkv.w_lo; g.ldw = ly.qkv.K;
            g.bias = ly.qkv.bias; g.act = DIDI_ACT_NONE;
            g.C_hi = w.att_hi; g.ldc = H;
            prof.mark(stream);
            const int rc_qkv = didi_qkv_self_attention(&g, dec->heads, a->S, 0.f, stream);
            prof.mark(stream);
            TRY(rc_qkv);
        } else {
            TRY(gemm(ly.qkv, pc(0), M, w.h_hi, w.h_lo, H, DIDI_ACT_NONE, nullptr, nullptr, 0, nullptr, w.qkv_hi,
                     x3 ? w.qkv_lo : nullptr, 3 * H, stream, &prof, w.small_ws, a->fault, dith));
            at.Sq = a->S; at.Sk = a->S;
            at.causal = 1; at.key_mask = nullptr;
            at.Q_hi = w.qkv_hi; at.Q_lo = x3 ? w.qkv_lo : nullptr; at.ldq = 3 * H;
            at.K_hi = w.qkv_hi + H; at.K_lo = x3 ? w.qkv_lo + H : nullptr; at.ldk = 3 * H;
            at.V_hi = w.qkv_hi + 2 * H; at.V_lo = x3 ? w.qkv_lo + 2 * H : nullptr; at.ldv = 3 * H;
            TRY(didi_attention(&at, stream));
        }
        TRY(gemm_ln(ly.attn_out, pc(1), fmt, M, w.att_hi, w.att_lo, H, w.h_hi, lo8 && l > 0 ? h_lo8 : w.h_lo, H, w.tmp, w.ln_stats, ++epoch, ly.ln1_g,
                    ly.ln1_b, dec->ln_eps, w.h_hi, lo8 ? h_lo8 : w.h_lo, nullptr, stream, &prof, w.small_ws, a->fault, dith, hi_res ? 3 : 0,
                    lo8 ? (l > 0 ? 3 : 2) : 0));
        // --- cross-attention block (K/V hoisted: didi_decoder_prepare_context)
        u16* q_hi = w.qkv_hi;
        u16* q_lo = x3 ? w.qkv_lo : nullptr;
        const size_t kvoff = (size_t)l * Mc * 2 * H;
        // the query projection AND the attention over the hoisted K / V in one launch, under the same conditions as above
        const bool fused_cross = (didi_opt(DIDI_OPT_FUSED_ATTN) & 2) != 0 && !x3 && ly.cq.bias && ly.cq.N == H &&
                                 didi_opt(DIDI_OPT_GEMM_BIG) != 0 && didi_opt(DIDI_OPT_GEMM_PP) != 0 && M >= didi_opt(DIDI_OPT_BIG_MIN_M) &&
                                 didi_q_cross_attention_supported(M, H, dec->heads, a->S, a->C, ly.cq.K, pc(2));
        if (fused_cross) {
            didi_gemm_args g{};
            g.fault = a->fault;
            g.act_dither = dith;
            g.M = M; g.N = H; g.K = ly.cq.K; g.precision = pc(2);
            g.A_hi = w.h_hi; g.A_lo = w.h_lo; g.lda = H;
            g.W_hi = ly.cq.w_hi; g.W_lo = ly.cq.w_lo; g.ldw = ly.cq.K;
            g.bias = ly.cq.bias; g.act = DIDI_ACT_NONE;
            g.C_hi = w.att_hi; g.ldc = H;
            prof.mark(stream);
            const int rc_cq = didi_q_cross_attention(&g, a->kv_hi + kvoff, a->kv_hi + kvoff + H, 2 * H, a->ctx_mask, a->ctx_count, dec->heads,
                                                     a->S, a->C, 0.f, stream);
            prof.mark(stream);
            TRY(rc_cq);
        } else {
            TRY(gemm(ly.cq, pc(2), M, w.h_hi, w.h_lo, H, DIDI_ACT_NONE, nullptr, nullptr, 0, nullptr, q_hi, q_lo, H, stream, &prof, w.small_ws, a->fault, dith));
            at.Sq = a->S; at.Sk = a->C; at.causal = 0; at.key_mask = a->ctx_mask; at.key_count = a->ctx_count;
            at.Q_hi = q_hi; at.Q_lo = q_lo; at.ldq = H;
            at.K_hi = a->kv_hi + kvoff; at.K_lo = x3 ? a->kv_lo + kvoff : nullptr; at.ldk = 2 * H;
            at.V_hi = a->kv_hi + kvoff + H; at.V_lo = x3 ? a->kv_lo + kvoff + H : nullptr; at.ldv = 2 * H;
            TRY(didi_attention(&at, stream));
        }
        TRY(gemm_ln(ly.cross_out, pc(3), fmt, M, w.att_hi, w.att_lo, H, w.h_hi, lo8 ? h_lo8 : w.h_lo, H, w.tmp, w.ln_stats, ++epoch, ly.ln2_g, ly.ln2_b,
                    dec->ln_eps, w.h_hi, lo8 ? h_lo8 : w.h_lo, nullptr, stream, &prof, w.small_ws, a->fault, dith, hi_res ? 3 : 0, lo8 ? 3 : 0));
        // --- feed-forward block
        TRY(gemm(ly.ffn_up, pc(4), M, w.h_hi, w.h_lo, H, DIDI_ACT_GELU, nullptr, nullptr, 0, nullptr, w.ffn_hi,
                 ffn_lo ? w.ffn_lo : nullptr, I, stream, &prof, w.small_ws, a->fault, dith));
        u16* o_hi = last && a->x0_hi ? a->x0_hi : w.h_hi;
        u16* o_lo = last && a->x0_hi ? a->x0_lo : w.h_lo;
        TRY(gemm_ln(ly.ffn_down, pc(5), fmt, M, w.ffn_hi, w.ffn_lo, I, w.h_hi, lo8 ? h_lo8 : w.h_lo, H, w.tmp, w.ln_stats, ++epoch, ly.ln3_g, ly.ln3_b,
                    dec->ln_eps, o_hi, lo8 && !last ? h_lo8 : o_lo, last ? a->x0_f32 : nullptr, stream, &prof, w.small_ws, a->fault, dith,
                    hi_res ? (last ? 1 : 3) : 0, lo8 ? (last ? 1 : 3) : 0));
    }
    if (a->prof_used_host) *a->prof_used_host = prof.used;
    return DIDI_OK;
}

extern "C" int didi_sample(const didi_decoder* dec, const didi_sample_args* s, void* stream) {
    DIDI_CHECK_ARG(dec && s && s->n_calls >= 1 && s->time_row_host && s->c0_host && s->c1_host);
    DIDI_CHECK_ARG(s->x0_f32 && s->workspace && s->kv_hi && s->ctx_mask);
    if (s->euler) DIDI_CHECK_ARG(s->c2_host && s->x_state);
    if (s->ids64) DIDI_CHECK_ARG(s->cls_hi && s->argmax_partial && s->x0_hi && s->V > 0);
    if (s->ids64 && s->precision == DIDI_PREC_BF16X3) DIDI_CHECK_ARG(s->cls_lo && s->x0_lo);
    if (s->ids64 && s->precision == DIDI_PREC_FP16X2) DIDI_CHECK_ARG(s->cls_lo);
    if (s->clamp_emb) DIDI_CHECK_ARG(s->ids64 && s->x0_hi && s->x0_lo);
    if (s->precision_host) {  // every call must work on the same plane format (the workspace planes and the K/V cache are shared)
        auto is_f16 = [](int p) { return p == DIDI_PREC_FP16 || p == DIDI_PREC_FP16X2 || p == DIDI_PREC_FP16X3; };
        DIDI_CHECK_ARG(is_f16(s->precision));
        for (int k = 0; k < s->n_calls; ++k) DIDI_CHECK_ARG(is_f16(s->precision_host[k]));
    }
    if (!shapes_ok(dec, s->B, s->S, s->C)) return DIDI_ERR_UNSUPPORTED;
    const size_t MH = (size_t)s->B * s->S * dec->H;
    const uint64_t ctr_per_draw = MH / 4;

    didi_forward_args f{};
    f.B = s->B; f.S = s->S; f.C = s->C; f.precision = s->precision;
    f.kv_hi = s->kv_hi; f.kv_lo = s->kv_lo; f.ctx_mask = s->ctx_mask; f.ctx_count = s->ctx_count;
    f.workspace = s->workspace; f.workspace_bytes = s->workspace_bytes;
    f.x0_f32 = s->x0_f32;
    f.fault = s->fault;
    auto draw = [&](int i) -> const float* { return s->noise ? s->noise + (size_t)i * MH : nullptr; };
    // precision schedule: one precision per denoiser call (all on the plane format of `precision`)
    auto prec_of = [&](int k) { return s->precision_host ? s->precision_host[k] : s->precision; };
    // one-product calls cycle through differently rounded copies of the weights (didi_sample_args.dec_variants_host)
    // ... and round their fp16 activation planes with an offset that cycles too (didi_sample_args.act_dither)
    auto dith_of = [&](int k) { return s->act_dither && prec_of(k) == DIDI_PREC_FP16 ? 1024 + 2048 * ((k + k / 4) & 3) : 0; };
    auto dec_of = [&](int k) { return s->dec_variants_host && s->n_variants > 0 && prec_of(k) == DIDI_PREC_FP16 ? s->dec_variants_host[k % s->n_variants] : dec; };
    // per-step clamp extension: x0 <- emb[argmax(classifier(x0))]; the forward leaves x0 planes for the rounding GEMM
    const bool clamp = s->clamp_emb != nullptr;
    auto clamp_x0 = [&]() -> int {
        int rc = didi_round_argmax(s->x0_hi, s->x0_lo, dec->H, s->cls_hi, s->cls_lo, dec->H, s->B * s->S, s->V, dec->H,
                                   s->precision, s->argmax_partial, s->ids64, nullptr, stream);
        if (rc != DIDI_OK) return rc;
        return didi_embed_gather(s->ids64, s->clamp_emb, 1.0f, nullptr, s->B * s->S, s->S, dec->H, s->x0_f32, stream);
    };
    auto split_x0 = [&]() -> int {
        return s->precision == DIDI_PREC_FP16 || s->precision == DIDI_PREC_FP16X2 ? didi_split_fp16(s->x0_f32, s->x0_hi, s->x0_lo, MH, stream)
                                              : didi_split_bf16(s->x0_f32, s->x0_hi, s->x0_lo, MH, stream);
    };
    auto set_prof = [&](int k) {
        f.prof_events_host = nullptr; f.prof_capacity = 0; f.prof_used_host = nullptr;
        if (!s->prof_events_host || !s->prof_calls_host || s->n_prof_calls <= 0) return;
        const int cap = s->prof_capacity / s->n_prof_calls;
        for (int j = 0; j < s->n_prof_calls; ++j)
            if (s->prof_calls_host[j] == k) {
                f.prof_events_host = s->prof_events_host + (size_t)j * cap;
                f.prof_capacity = cap;
                f.prof_used_host = s->prof_used_host ? s->prof_used_host + j : nullptr;
            }
    };

    if (!s->euler) {
        for (int k = 0; k < s->n_calls; ++k) {
            didi_step_args& st = f.step;
            st = didi_step_args{};
            st.cin = 1.0f;
            st.E = draw(k);
            st.seed = s->seed; st.offset = s->base_offset + (uint64_t)k * ctr_per_draw;
            if (k == 0) { st.mode = DIDI_STEP_INIT; st.ce = s->init_scale; }
            else { st.mode = DIDI_STEP_DDPM; st.X0 = s->x0_f32; st.cx0 = s->c0_host[k - 1]; st.ce = s->c1_host[k - 1]; }
            st.t = s->time_row_host[k];
            st.xin_out = s->xt_trace ? s->xt_trace + (size_t)k * MH : nullptr;
            set_prof(k);
            f.precision = prec_of(k);
            f.act_dither = dith_of(k);
            const bool last = k == s->n_calls - 1;
            f.x0_hi = last || clamp ? s->x0_hi : nullptr;
            f.x0_lo = last || clamp ? s->x0_lo : nullptr;
            TRY(didi_decoder_forward(dec_of(k), &f, stream));
            if (clamp) {
                TRY(clamp_x0());
                if (last) TRY(split_x0());  // the final rounding reads the planes of the clamped x0
            }
        }
    } else {
        didi_step_args init{};
        init.mode = DIDI_STEP_INIT; init.M = s->B * s->S; init.S = s->S; init.H = dec->H;
        init.E = draw(0); init.ce = s->init_scale; init.cin = 1.0f;
        init.seed = s->seed; init.offset = s->base_offset;
        init.x_out = s->x_state;
        TRY(didi_step_embed_ln(&init, stream));
        for (int k = 0; k < s->n_calls; ++k) {
            didi_step_args& st = f.step;
            st = didi_step_args{};
            st.mode = DIDI_STEP_EULER;
            st.X = s->x_state;
            st.X0 = k == 0 ? s->x_state : s->x0_f32;
            st.ratio = k == 0 ? 0.0f : s->c2_host[k - 1];
            st.ce = s->c0_host[k];
            st.cin = s->c1_host[k];
            st.E = draw(k + 1);
            st.seed = s->seed; st.offset = s->base_offset + (uint64_t)(k + 1) * ctr_per_draw;
            st.x_out = s->x_state;
            st.t = s->time_row_host[k];
            st.xin_out = s->xt_trace ? s->xt_trace + (size_t)k * MH : nullptr;
            f.x0_hi = clamp ? s->x0_hi : nullptr;
            f.x0_lo = clamp ? s->x0_lo : nullptr;
            f.precision = prec_of(k);
            f.act_dither = dith_of(k);
            set_prof(k);
            TRY(didi_decoder_forward(dec_of(k), &f, stream));
            if (clamp) TRY(clamp_x0());
        }
        // final Euler update lands on sigma = 0 (src/sampling.py:80): x0 <- x + (x - x0_hat) * ratio
        didi_step_args fin{};
        fin.mode = DIDI_STEP_EULER; fin.M = s->B * s->S; fin.S = s->S; fin.H = dec->H;
        fin.X = s->x_state; fin.X0 = s->x0_f32; fin.ratio = s->c2_host[s->n_calls - 1]; fin.ce = 0.f; fin.cin = 1.0f;
        fin.x_out = s->x0_f32;
        TRY(didi_step_embed_ln(&fin, stream));
        if (s->x0_hi) TRY(split_x0());
    }
    if (s->ids64) {
        // the FINAL rounding of the fp16x2 mode (and of any precision schedule) multiplies all three products (once per batch: free): the ids then carry the
        // loop's error only, not the operand rounding of the classifier GEMM on top of it
        const int round_prec = (s->precision == DIDI_PREC_FP16X2 || s->precision_host) && s->x0_lo && s->cls_lo ? DIDI_PREC_FP16X3 : s->precision;
        TRY(didi_round_argmax(s->x0_hi, s->x0_lo, dec->H, s->cls_hi, s->cls_lo, dec->H, s->B * s->S, s->V, dec->H,
                              round_prec, s->argmax_partial, s->ids64, nullptr, stream));
    }
    return DIDI_OK;
}

// ------------------------------------------------------------------ timing helpers for bench.py
// hipEvent wrappers so that the benchmark creates its events in the SAME HIP runtime instance this
// library is bound to (a ctypes-loaded second copy of libamdhip64 would not share streams).
extern "C" int didi_event_create(void** ev) {
    DIDI_CHECK_ARG(ev);
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return DIDI_ERR_LAUNCH;
    *ev = e;
    return DIDI_OK;
}
extern "C" int didi_event_elapsed_ms(void* start, void* stop, float* ms) {
    DIDI_CHECK_ARG(start && stop && ms);
    return hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)) == hipSuccess
               ? DIDI_OK : DIDI_ERR_LAUNCH;
}
extern "C" int didi_event_destroy(void* ev) {
    DIDI_CHECK_ARG(ev);
    return hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)) == hipSuccess ? DIDI_OK : DIDI_ERR_LAUNCH;
}
