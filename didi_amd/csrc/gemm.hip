// MFMA GEMM for the decoder's Linear layers and the rounding (classifier + argmax) step.
//
//   C[M,N] = epilogue( A[M,K] * W[N,K]^T )      fp32 accumulate, operands as split 16-bit planes
//
// Precisions: bf16 / fp16 (one product on the hi planes, v_mfma_f32_16x16x32_{bf16,f16}) and bf16x3
// (lo*hi + hi*lo + hi*hi on bf16 planes, fused into one K-loop).  Kernels, by shape:
//   gemm_pp_kernel   256(M) x 192(N) tile, 8 waves in two staggered groups      M >= 2048 (the sampler's
//                    (the fast path: see its header further down)                 B*S = 16384 rows)
//   gemm_big_kernel  same tile, in-phase K-loop (A/B baseline, DIDI_GEMM_PP=0; MT=2: 128 x 192, 4 waves)
//   gemm_kernel      128 x 128 tile, 4 waves (2 x 2 of 64 x 64)                   small M, odd N
//   tile_regstage    register-staged fallback inside gemm_kernel                  K not a multiple of the K-tile
//   gemm_argmax_kernel  128-row tiles over vocabulary chunks, running (max, index), logits never written;
//                       CE variant: also an online log-sum-exp and the target's logit (validation cross-entropy)
// The MFMA is issued "swapped" (A-operand = weight rows, B-operand = activation rows) so that every
// lane ends up holding CONSECUTIVE OUTPUT COLUMNS n of one output row m: the epilogue then loads
// bias as float4s and stores 8/16-byte plane packs and 16-byte fp32 vectors instead of scalars.
//
// LDS-DMA main loops: operand tiles go HBM/L2 -> LDS by global_load_lds_dwordx4 (no staging
// registers, no ds_write -- the register-staged first version of this kernel was bound by
// ds_write_b128 issue, profiles/r01_v1_*), two LDS stages, counted vmcnt + raw s_barrier.
//   one product : K-tile 64;  stage = {A, W} tiles of [rows][128 B]
//   bf16x3      : K-tile 32;  stage = {A_hi, A_lo, W_hi, W_lo} tiles of [rows][64 B]; the three
//                 products are issued from ONE staging of the four tiles (72 MFMAs per 20 fragment
//                 reads per wave in the big tile) instead of three trips over K.
// LDS image: an LDS-DMA instruction writes 1 KiB = lane-linear 16-byte slots, so the bank-conflict
// swizzle is applied to the per-lane SOURCE address and again on the fragment read
// (cdna_hip_programming.md rule 21): slot s of row r holds k-chunk s ^ f(r), with
// f(r) = r & 7 for 128-byte rows and f(r) = (-(r >> 2)) & 3 for 64-byte rows; both make every
// 16-lane ds_read_b128 group hit 16 distinct 16-byte slots of the 256-byte bank row.
//
// Grid: 1-D, XCD-aware: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs,
// so id -> (id % 8) * chunk + id / 8 gives each XCD (private 4 MiB L2) a contiguous band of output
// row-tiles: a band's activation rows are fetched from HBM once per XCD while the small weight
// matrix (<= 4.7 MB) stays L2-resident.
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <type_traits>

#include "common.hpp"

#define BM 128
#define BN 128
#define STAGE_BYTES 32768
#define SMEM_BYTES (2 * STAGE_BYTES)

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

struct GemmParams {
    int M, N, K;
    int npass;
    int f16;  // plane element format: 0 bf16, 1 fp16 (single-product mode only)
    const u16 *A_hi, *A_lo, *W_hi, *W_lo;
    int lda, ldw;
    const float* bias;
    int act;
    const float* R_f32;
    const u16 *R_hi, *R_lo;
    int ldr;
    float* C_f32;
    u16 *C_hi, *C_lo;
    int ldc;
    // argmax variant
    int tiles_m, tiles_n, nchunks, tiles_per_chunk;
    float2* partial;  // [M][nchunks] (value, index bits)
    int k_split;               // split-K form (gemm_kernel, gridDim.y > 1): K columns per split; partial sums go to C_f32 + y*M*N
    float2* partial2;          // CE variant: [M][nchunks] (sum exp(logit - chunk max), target logit or -inf)
    const int64_t* targets;    // CE variant: [M] target ids
    // fused LayerNorm epilogue (gemm_pp_kernel<.., LN = true>): y = LayerNorm(acc + bias + residual) -> C_*
    const float *ln_gamma, *ln_beta;
    float ln_eps;
    unsigned long long* ln_stats;  // [tiles_m][tiles_n][256][2] granules {epoch, bits} of each tile's per-row (sum, M2)
    unsigned ln_epoch;             // tag of this launch's granules (>= 1; older tags and zero are stale)
    uint32_t* fault;               // device fault word (DIDI_FAULT_*) or null
    const u16 *attn_K, *attn_V;    // ATTN == 2 (cross-attention): hoisted K / V planes [B * 64][attn_ldkv], head h at column h * 64
    int attn_ldkv, attn_B;
    const int32_t *attn_mask, *attn_count;  // [B][64] key mask (0 = padded) or null; [B] visible-key counts or null
    int lo8;                       // LN form, fp16 planes: bit 0: R_lo is a BYTE plane (split4_dither_lo8), bit 1: C_lo is written as one
    float attn_scale;              // ATTN form: softmax scale * log2(e)
    float attn_mask0;              // ATTN form: the additive mask of a visible key, 0.0f, as a run-time value: the scores are then
                                   // fmaf(acc, scale, mask) exactly as in attn_kernel (a literal 0 would let the compiler fold the
                                   // product into the subtraction of the row maximum: another rounding, 1 ulp in 2e-5 of the outputs)
    unsigned long long* stamps;    // diagnostic builds (-DDIDI_STAMPS) only: s_memtime stamps [workgroup][wave group][16]
    int dither;                    // fp16 activation-plane rounding dither of this launch (common.hpp f16x2_dither), 0 = RNE;
                                   // honoured by the one-product fp16 form of gemm_pp_kernel (the sampler's hot path)
};

// In-kernel stamps of gemm_pp_kernel (diagnostic build only: in the product library PP_STAMP expands to nothing).  Lane 0 of
// waves 0 (early wave group) and 4 (late group) writes s_memtime at fixed points of the kernel into p.stamps, a buffer of
// its own that nothing else reads (MI355X_MICROARCH.md, DVFS give-back item 6); slots 14 / 15 hold s_memrealtime (100 MHz) at
// the first and last stamp so the in-kernel clock can be derived.  tools/ln_stamps.py reads them.
#ifdef DIDI_STAMPS
static unsigned long long* g_stamp_buf = nullptr;
extern "C" void didi_debug_set_stamps(void* buf) { g_stamp_buf = reinterpret_cast<unsigned long long*>(buf); }
#define PP_STAMP(i)                                                                                          \
    do {                                                                                                     \
        if (p.stamps && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) & 3) == 0) {                          \
            unsigned long long* s_ = p.stamps + ((size_t)blockIdx.x * 2 + (threadIdx.x >> 8)) * 16;          \
            s_[i] = __builtin_amdgcn_s_memtime();                                                            \
            if ((i) == 0) { s_[14] = __builtin_amdgcn_s_memrealtime(); s_[12] = t_entry; }                   \
            if ((i) == 7) s_[15] = __builtin_amdgcn_s_memrealtime();                                         \
        }                                                                                                    \
    } while (0)
#else
#define PP_STAMP(i) do {} while (0)
#endif

__device__ __forceinline__ int swz128(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ int swz64(int row, int chunk) { return row * 64 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 4); }

__device__ __forceinline__ void glds16(const u16* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)lds_dst, 16, 0, 0);
}
// The same instruction out of the compiler's sight.  hipcc tracks the builtin as a pending write to LDS and puts
// s_waitcnt vmcnt(0) in front of later LDS reads it cannot prove disjoint (float and transposed reads in particular): right
// where everything is waited for anyway, fatal for a loop that keeps the NEXT buffer's pieces (and its stores) in flight across
// reads of the current one.  Every use is followed by a hand-counted s_waitcnt vmcnt and a barrier before the first read.
__device__ __forceinline__ void glds16_raw(const u16* src, char* lds_dst) {
    const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void*)lds_dst);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m) : "memory", "m0");
#pragma clang diagnostic pop
}

// ------------------------------------------------------------------ LDS-DMA main loop
// Accumulate one 128x128 output tile at (m0, n0).  Requires K % (NPASS == 3 ? 32 : 64) == 0.
template <int NPASS, int NSTAGE, bool F16>
__device__ __forceinline__ void tile_glds(const GemmParams& p, char* smem, int m0, int n0, f32x4 (&acc)[4][4]) {
    constexpr int BKK = NPASS == 1 ? 64 : 32;
    constexpr int ROWB = BKK * 2;              // bytes per LDS row
    constexpr int TILEB = 128 * ROWB;          // one operand tile
    constexpr int CPR = ROWB / 16;             // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;              // rows covered by one wave-instruction (1 KiB)
    constexpr int IPW = 128 / RPI / 4;         // DMA instructions per tile per wave
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lg = lane >> 4;

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane DMA source offsets (elements): instruction j of this wave fills LDS rows
    // (wave*IPW + j)*RPI ... ; lane -> row = base + lane / CPR, slot = lane % CPR, k-chunk = slot ^ f(row)
    size_t a_src[IPW], w_src[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        const int row = (wave * IPW + j) * RPI + lane / CPR;
        const int slot = lane % CPR;
        const int g = NPASS != 1 ? (slot ^ ((0 - (row >> 2)) & 3)) : (slot ^ (row & 7));
        a_src[j] = (size_t)min(m0 + row, p.M - 1) * p.lda + g * 8;
        w_src[j] = (size_t)min(n0 + row, p.N - 1) * p.ldw + g * 8;
    }
    const int nt = p.K / BKK;

    auto issue = [&](int t, int buf) {
        char* st = smem + buf * STAGE_BYTES + wave * IPW * 1024;  // wave-uniform; the DMA adds lane*16
        const int k0 = t * BKK;
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            if constexpr (NPASS == 3) {
                glds16(p.A_hi + k0 + a_src[j], st + 0 * TILEB + j * 1024);
                glds16(p.A_lo + k0 + a_src[j], st + 1 * TILEB + j * 1024);
                glds16(p.W_hi + k0 + w_src[j], st + 2 * TILEB + j * 1024);
                glds16(p.W_lo + k0 + w_src[j], st + 3 * TILEB + j * 1024);
            } else if constexpr (NPASS == 2) {  // fp16x2: the activation remainder is not multiplied, its tile slot stays empty
                glds16(p.A_hi + k0 + a_src[j], st + 0 * TILEB + j * 1024);
                glds16(p.W_hi + k0 + w_src[j], st + 2 * TILEB + j * 1024);
                glds16(p.W_lo + k0 + w_src[j], st + 3 * TILEB + j * 1024);
            } else {
                glds16(p.A_hi + k0 + a_src[j], st + 0 * TILEB + j * 1024);
                glds16(p.W_hi + k0 + w_src[j], st + 1 * TILEB + j * 1024);
            }
        }
    };

    // NSTAGE-deep ring: K-tiles t+1 .. t+NSTAGE-2 stay in flight across the barrier (counted vmcnt,
    // raw s_barrier: __syncthreads() would drain the DMA queue, cdna_hip_programming.md section 5)
    constexpr int G = IPW * (NPASS == 3 ? 4 : NPASS == 2 ? 3 : 2);  // DMA instructions per stage per wave
    static_assert(G == 8 || G == 6, "vmcnt immediates below assume 8 (or, fp16x2, 6) DMA instructions per stage");
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nt) issue(s, s);
    int buf = 0, nbuf = NSTAGE - 1;
    for (int t = 0; t < nt; ++t) {
        // this wave's share of K-tile t has landed once at most the younger stages remain outstanding
        const int ahead = min(NSTAGE - 2, nt - 1 - t);
        if (ahead <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (ahead == 1) { if constexpr (G == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else { if constexpr (G == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();  // everyone's share has landed; ring slot of K-tile t-1 is free
        if (t + NSTAGE - 1 < nt) issue(t + NSTAGE - 1, nbuf);
        const char* st = smem + buf * STAGE_BYTES;
        buf = buf + 1 == NSTAGE ? 0 : buf + 1;
        nbuf = nbuf + 1 == NSTAGE ? 0 : nbuf + 1;
        if constexpr (NPASS != 1) {
            const char *Ah = st, *Al = st + TILEB, *Wh = st + 2 * TILEB, *Wl = st + 3 * TILEB;
            bf16x8 whf[4], wlf[4], ahf[4], alf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int wo = swz64(wn * 64 + i * 16 + lr, lg);
                const int ao = swz64(wm * 64 + i * 16 + lr, lg);
                whf[i] = *reinterpret_cast<const bf16x8*>(Wh + wo);
                wlf[i] = *reinterpret_cast<const bf16x8*>(Wl + wo);
                ahf[i] = *reinterpret_cast<const bf16x8*>(Ah + ao);
                if constexpr (NPASS == 3) alf[i] = *reinterpret_cast<const bf16x8*>(Al + ao);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    acc[ni][mi] = mma16<F16>(wlf[ni], ahf[mi], acc[ni][mi]);
                    if constexpr (NPASS == 3) acc[ni][mi] = mma16<F16>(whf[ni], alf[mi], acc[ni][mi]);
                    acc[ni][mi] = mma16<F16>(whf[ni], ahf[mi], acc[ni][mi]);
                }
        } else {
            const char *As = st, *Ws = st + TILEB;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 wf[4], af[4];
                const int chunk = ks * 4 + lg;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    wf[i] = *reinterpret_cast<const bf16x8*>(Ws + swz128(wn * 64 + i * 16 + lr, chunk));
                    af[i] = *reinterpret_cast<const bf16x8*>(As + swz128(wm * 64 + i * 16 + lr, chunk));
                }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        acc[ni][mi] = mma16<F16>(wf[ni], af[mi], acc[ni][mi]);
            }
        }
    }
    __syncthreads();  // the staging buffers may be reused right away (argmax variant: next tile)
}

// ------------------------------------------------------------------ register-staged fallback (any K % 8 == 0)
template <bool F16>
__device__ __forceinline__ void tile_regstage(const GemmParams& p, char* smem, int m0, int n0, f32x4 (&acc)[4][4]) {
    constexpr int BK = 64, TILE_BYTES = 128 * 128;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lg = lane >> 4;

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging assignment: 4 chunks (16 B) of each operand tile per thread; rows past M / N are
    // clamped to the last valid row (their products land in accumulators that are never stored)
#define GEMM_STAGE(I)                                                        \
    const int c##I = tid + 256 * I;                                          \
    const int so##I = swz128(c##I >> 3, c##I & 7);                           \
    const size_t a_off##I = (size_t)min(m0 + (c##I >> 3), p.M - 1) * p.lda + (c##I & 7) * 8; \
    const size_t w_off##I = (size_t)min(n0 + (c##I >> 3), p.N - 1) * p.ldw + (c##I & 7) * 8;
    GEMM_STAGE(0) GEMM_STAGE(1) GEMM_STAGE(2) GEMM_STAGE(3)
#undef GEMM_STAGE
    const int kt_per_pass = (p.K + BK - 1) / BK;  // last K-tile may be partial: chunks past K load as zero
    const int nt = kt_per_pass * p.npass;

    // (no lambdas / runtime-indexed arrays here: either leaves the prefetch registers in scratch)
    // passes: 0 = A_lo*W_hi, 1 = A_hi*W_lo, 2 = the dominant A_hi*W_hi; npass of them are run, the last ones
#define GEMM_GLOAD(T)                                                                   \
    {                                                                                   \
        const int pass_ = 3 - p.npass + (T) / kt_per_pass; /* npass 1: hi*hi; 2: hi*lo, hi*hi; 3: all */ \
        const int k0_ = ((T) % kt_per_pass) * BK;                                       \
        const u16* Ab_ = (pass_ == 0 ? p.A_lo : p.A_hi) + k0_;                          \
        const u16* Wb_ = (pass_ == 1 ? p.W_lo : p.W_hi) + k0_;                          \
        const uint4 z_ = make_uint4(0, 0, 0, 0);                                        \
        const bool in0_ = k0_ + (c0 & 7) * 8 < p.K, in1_ = k0_ + (c1 & 7) * 8 < p.K;    \
        const bool in2_ = k0_ + (c2 & 7) * 8 < p.K, in3_ = k0_ + (c3 & 7) * 8 < p.K;    \
        ra0 = in0_ ? *reinterpret_cast<const uint4*>(Ab_ + a_off0) : z_;                \
        ra1 = in1_ ? *reinterpret_cast<const uint4*>(Ab_ + a_off1) : z_;                \
        ra2 = in2_ ? *reinterpret_cast<const uint4*>(Ab_ + a_off2) : z_;                \
        ra3 = in3_ ? *reinterpret_cast<const uint4*>(Ab_ + a_off3) : z_;                \
        rw0 = in0_ ? *reinterpret_cast<const uint4*>(Wb_ + w_off0) : z_;                \
        rw1 = in1_ ? *reinterpret_cast<const uint4*>(Wb_ + w_off1) : z_;                \
        rw2 = in2_ ? *reinterpret_cast<const uint4*>(Wb_ + w_off2) : z_;                \
        rw3 = in3_ ? *reinterpret_cast<const uint4*>(Wb_ + w_off3) : z_;                \
    }
#define GEMM_LSTORE(BUF)                                                                \
    {                                                                                   \
        char* As_ = smem + (BUF) * 2 * TILE_BYTES;                                      \
        char* Ws_ = As_ + TILE_BYTES;                                                   \
        *reinterpret_cast<uint4*>(As_ + so0) = ra0; *reinterpret_cast<uint4*>(Ws_ + so0) = rw0; \
        *reinterpret_cast<uint4*>(As_ + so1) = ra1; *reinterpret_cast<uint4*>(Ws_ + so1) = rw1; \
        *reinterpret_cast<uint4*>(As_ + so2) = ra2; *reinterpret_cast<uint4*>(Ws_ + so2) = rw2; \
        *reinterpret_cast<uint4*>(As_ + so3) = ra3; *reinterpret_cast<uint4*>(Ws_ + so3) = rw3; \
    }
    uint4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
    GEMM_GLOAD(0);
    GEMM_LSTORE(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const bool more = t + 1 < nt;
        if (more) GEMM_GLOAD(t + 1);
        const char* As = smem + buf * 2 * TILE_BYTES;
        const char* Ws = As + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 wf[4], af[4];
            const int chunk = ks * 4 + lg;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wf[i] = *reinterpret_cast<const bf16x8*>(Ws + swz128(wn * 64 + i * 16 + lr, chunk));
                af[i] = *reinterpret_cast<const bf16x8*>(As + swz128(wm * 64 + i * 16 + lr, chunk));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = mma16<F16>(wf[ni], af[mi], acc[ni][mi]);
        }
        if (more) GEMM_LSTORE(buf ^ 1);
        __syncthreads();
    }
#undef GEMM_GLOAD
#undef GEMM_LSTORE
}

// MODE 0: register-staged fallback, 1: LDS-DMA one product, 2: LDS-DMA fp16x2 (A_hi*W_hi + A_hi*W_lo), 3: LDS-DMA bf16x3
template <int MODE, int NSTAGE, bool F16>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, char* smem, int m0, int n0, f32x4 (&acc)[4][4]) {
    if constexpr (MODE == 0) tile_regstage<F16>(p, smem, m0, n0, acc);
    else tile_glds<MODE, NSTAGE, F16>(p, smem, m0, n0, acc);
}

// erf to <= 1.5e-7 absolute (Abramowitz & Stegun 7.1.26): rcp + exp + 5 fma instead of libm's
// ~40-instruction erff, which cost the FFN-up epilogue ~40 us per launch (profiles/r01 notes).
// GELU(x) = 0.5 x (1 + erf(x/sqrt2)) then carries <= 1.5e-7 relative error: far inside the
// bf16x3 budget (1e-5) and invisible next to the bf16 split rounding of the stored value.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * __expf(-ax * ax);
    return copysignf(r, x);
}

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == DIDI_ACT_GELU) return 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752440f));
    if (act == DIDI_ACT_TANH) return tanhf(v);
    if (act == DIDI_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == DIDI_ACT_GELU_NEW) return 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v)));
    return v;
}

// GELU for the one-product precisions (gemm_pp_kernel): GELU(x) = max(x, 0) - g(|x|) with
// g(a) = a (1 - Phi(a)) read from a 1024-interval table in LDS and interpolated linearly (3.4e-6
// absolute, the result is rounded to an 11- or 8-bit mantissa right after).  8 VALU + one ds_read_b64
// instead of 14 VALU + v_rcp + v_exp: the erf form cost the FFN-up launch 45 us of pure VALU time
// with the MFMA pipe idle (tools/epi_probe.py).  bf16x3 keeps the 1.5e-7 erf form.
__device__ const float2 didi_gelu_table[1024] = {
#include "gelu_table.inc"
};
__device__ __forceinline__ float gelu_lut(float x, const float2* tbl) {
    const float u = fminf(fabsf(x) * (1024.0f / 6.0f), 1023.0f);
    const int i = (int)u;
    const float f = u - (float)i;
    const float2 c = tbl[i];
    return fmaxf(x, 0.0f) - (c.x + f * c.y);
}

// epilogue for 4 consecutive output columns n..n+3 of row m: bias, activation, residual, stores
// `amax`: the thread's running max |value| of what it stores as planes (the kernel reports DIDI_FAULT_FP16_RANGE from it at the end)
__device__ __forceinline__ void epilogue4(const GemmParams& p, int m, int n, float v0, float v1, float v2, float v3, float& amax) {
    if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;
    }
    if (p.act != DIDI_ACT_NONE) {
        v0 = act_apply(v0, p.act); v1 = act_apply(v1, p.act); v2 = act_apply(v2, p.act); v3 = act_apply(v3, p.act);
    }
    if (p.R_f32) {
        const float4 r = *reinterpret_cast<const float4*>(p.R_f32 + (size_t)m * p.ldr + n);
        v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
    } else if (p.R_hi) {
        const uint2 rh = *reinterpret_cast<const uint2*>(p.R_hi + (size_t)m * p.ldr + n);
        float r[4];
        unpack4(rh, r, p.f16);
        v0 += r[0]; v1 += r[1]; v2 += r[2]; v3 += r[3];
        if (p.R_lo) {
            const uint2 rl = *reinterpret_cast<const uint2*>(p.R_lo + (size_t)m * p.ldr + n);
            unpack4(rl, r, p.f16);
            v0 += r[0]; v1 += r[1]; v2 += r[2]; v3 += r[3];
        }
    }
    const size_t o = (size_t)m * p.ldc + n;
    if (p.C_f32) *reinterpret_cast<float4*>(p.C_f32 + o) = make_float4(v0, v1, v2, v3);
    if (p.C_hi) {
        const float v[4] = {v0, v1, v2, v3};
        amax = amax2(amax2(amax, v0, v1), v2, v3);
        uint2 h, l;
        split4(v, h, l, p.f16);
        *reinterpret_cast<uint2*>(p.C_hi + o) = h;
        if (p.C_lo) *reinterpret_cast<uint2*>(p.C_lo + o) = l;
    }
}

// the same for 8 consecutive output columns: 16-byte plane stores / loads instead of 8-byte ones (the
// epilogue is store-issue bound: half the instructions, twice the contiguous bytes per row)
// `pre` != 0: the residual planes of these 8 columns were loaded ahead by the caller (rh_pre, rl_pre)
// `bias8`: this tile's bias values for columns n..n+7, staged in LDS by the kernel's prologue (null: read p.bias from memory --
// a dependent global load at the head of the epilogue cost the qkv launch 5 us per round, tools/epi_cost.py)
__device__ __forceinline__ void epilogue8(const GemmParams& p, int m, int n, f32x4 lo4, f32x4 hi4, const float2* gelu_tbl, float& amax,
                                          int pre = 0, uint4 rh_pre = uint4{0, 0, 0, 0}, uint4 rl_pre = uint4{0, 0, 0, 0},
                                          const float* bias8 = nullptr, int dith = 0) {
    float v[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    if (p.bias) {
        const float* bsrc = bias8 ? bias8 : p.bias + n;
        const float4 b0 = *reinterpret_cast<const float4*>(bsrc);
        const float4 b1 = *reinterpret_cast<const float4*>(bsrc + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    if (p.act == DIDI_ACT_GELU && gelu_tbl) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = gelu_lut(v[i], gelu_tbl);
    } else if (p.act != DIDI_ACT_NONE) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = act_apply(v[i], p.act);
    }
    if (p.R_f32) {
        const float4 r0 = *reinterpret_cast<const float4*>(p.R_f32 + (size_t)m * p.ldr + n);
        const float4 r1 = *reinterpret_cast<const float4*>(p.R_f32 + (size_t)m * p.ldr + n + 4);
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
        v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
    } else if (p.R_hi) {
        const uint4 rh = pre ? rh_pre : *reinterpret_cast<const uint4*>(p.R_hi + (size_t)m * p.ldr + n);
        float r[4];
        unpack4(make_uint2(rh.x, rh.y), r, p.f16);
        v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
        unpack4(make_uint2(rh.z, rh.w), r, p.f16);
        v[4] += r[0]; v[5] += r[1]; v[6] += r[2]; v[7] += r[3];
        if (p.R_lo) {
            const uint4 rl = pre ? rl_pre : *reinterpret_cast<const uint4*>(p.R_lo + (size_t)m * p.ldr + n);
            unpack4(make_uint2(rl.x, rl.y), r, p.f16);
            v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
            unpack4(make_uint2(rl.z, rl.w), r, p.f16);
            v[4] += r[0]; v[5] += r[1]; v[6] += r[2]; v[7] += r[3];
        }
    }
    const size_t o = (size_t)m * p.ldc + n;
    if (p.C_f32) {
        *reinterpret_cast<float4*>(p.C_f32 + o) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p.C_f32 + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    amax = amax2(amax2(amax2(amax2(amax, v[0], v[1]), v[2], v[3]), v[4], v[5]), v[6], v[7]);
    if (p.C_hi && p.C_lo) {
        uint2 h0, l0, h1, l1;
        if (dith) { split4_dither(v, h0, l0, dith); split4_dither(v + 4, h1, l1, dith); }
        else { split4(v, h0, l0, p.f16); split4(v + 4, h1, l1, p.f16); }
        *reinterpret_cast<uint4*>(p.C_hi + o) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        *reinterpret_cast<uint4*>(p.C_lo + o) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    } else if (p.C_hi) {  // one-product precisions keep no remainder of GEMM-only operands
        const uint2 h0 = dith ? pack4_hi_dither(v, dith) : pack4_hi(v, p.f16), h1 = dith ? pack4_hi_dither(v + 4, dith) : pack4_hi(v + 4, p.f16);
        *reinterpret_cast<uint4*>(p.C_hi + o) = make_uint4(h0.x, h0.y, h1.x, h1.y);
    }
}

// bijective XCD remap of a 1-D grid (cdna_hip_programming.md, "XCD swizzle must be bijective")
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = id & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

template <int MODE, int NSTAGE, bool F16>
__global__ void __launch_bounds__(256, NSTAGE <= 2 ? 2 : 1) gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    if (p.k_split) {  // this workgroup multiplies K columns [y*k_split, ...) and writes raw partial sums
        const int k0 = blockIdx.y * p.k_split;
        p.A_hi += k0; p.A_lo += k0; p.W_hi += k0; p.W_lo += k0;
        p.K = min(p.k_split, p.K - k0);
        p.C_f32 += (size_t)blockIdx.y * p.M * p.ldc;
    }
    if constexpr (F16) f16_saturate_mode();
    f32x4 acc[4][4];
    gemm_tile<MODE, NSTAGE, F16>(p, smem, m0, n0, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lg = lane >> 4;
    float amax = 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + lr;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + lg * 4;
            if (n >= p.N) continue;
            epilogue4(p, m, n, acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3], amax);
        }
    }
    if constexpr (F16) raise_fault(p.fault, amax > DIDI_F16_MAX, DIDI_FAULT_FP16_RANGE);
}

// ------------------------------------------------------------------ big tile: 256(M) x 192(N), 8 waves
// The 128x128 tile at two workgroups per CU asks the XCD's L2 for 32 KiB per 512 (bf16) / 768
// (bf16x3) MFMA cycles and workgroup = 42-64 B/clk/CU at full MFMA rate, which is the L2's whole
// rate (MI355X_MICROARCH.md: ~34.5 TB/s = 56 B/clk/CU): the small tile is L2-bound.  This tile
// halves the bytes per FLOP (56 KiB per 1536 / 2304 MFMA cycles per SIMD = 36 / 24 B/clk/CU) and
// its 192 columns divide the decoder's widths exactly (768 = 4, 2304 = 12, 3072 = 16 tiles): with
// M = 16384 rows the grids are 256 / 768 / 1024 workgroups = 1 / 3 / 4 full rounds of the 256 CUs.
// 8 waves as 4(M) x 2(N), wave tile 64 x 96 = 4 x 6 MFMA accumulators; one stage = 56 one-KiB
// LDS-DMA pieces, wave w issues pieces w, w+8, ..., w+48; two stages = 112 KiB, one workgroup per CU.
//
// MT = 64-row wave tiles along M: MT = 4 is the 256-row tile above (8 waves, 1 workgroup/CU);
// MT = 2 is a 128 x 192 tile with 4 waves and 40 KiB stages, two workgroups per CU.
// What was measured on this loop (M = 16384, bf16x3, K = 3072, N = 768):
//   MFMAs only (no DMA after the first stage) 157 us, DMA only (no MFMAs) 170 us, both 208 us;
//   rocprofv3 PMC: MFMA pipe 56 % busy, SQ_WAIT_ANY 37 %, no LDS bank conflicts, L2 hit rate 88 %,
//   HBM fetch ~320 MB (A is fetched once: the four column tiles of a row band share it in L2).
//   The L2->LDS stream runs at ~8 TB/s chip-wide whatever carries it: LDS-DMA, register staging
//   (T14), both stages in flight (early stage release), MT = 2, a touch-ahead L2 prefetch and
//   padded leading dimensions were all within +-4 % of this form or slower, so it stays.
#define BBN 192
// dynamic LDS of gemm_pp_kernel: two stages, the 8-KiB GELU table, the tile's 192 bias (+ LayerNorm gamma, beta) values
#define PP_LDS_BYTES (2 * BigGeom<4>::STAGE + 8192 + 3072)
template <int MT> struct BigGeom {
    static constexpr int BM_ = MT * 64, WAVES = MT * 2, THREADS = WAVES * 64;
    static constexpr int A_BYTES = MT * 64 * 128;       // A region: [BM][128 B], or hi+lo tiles of [BM][64 B]
    static constexpr int NA = A_BYTES / 1024, NW = 24;  // 1-KiB DMA pieces per region
    static constexpr int STAGE = A_BYTES + NW * 1024;
    static constexpr int PPW = (NA + NW) / WAVES;       // pieces per wave per stage
};
template <int NPASS, int MT, bool F16>
__global__ void __launch_bounds__(BigGeom<MT>::THREADS, 2) gemm_big_kernel(GemmParams p) {
    using G = BigGeom<MT>;
    constexpr int BKK = NPASS == 3 ? 32 : 64;
    constexpr int ROWB = BKK * 2, CPR = ROWB / 16, RPI = 64 / CPR;
    constexpr int W_OFF = G::A_BYTES;                   // W tile(s) start
    constexpr int LO_A = G::A_BYTES / 2, LO_W = 12288;  // bf16x3: offset of the lo tile after its hi tile
    constexpr int NA = G::NA, PPW = G::PPW, WAVES = G::WAVES, BIG_STAGE_BYTES = G::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / p.tiles_n) * G::BM_, n0 = (tile % p.tiles_n) * BBN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lg = lane >> 4;

    f32x4 acc[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // DMA piece q = wave + WAVES*j (1 KiB at stage offset q*1024): which operand plane, which rows
    const u16* src[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int q = wave + WAVES * j;
        const u16* base;
        int row, ld, lim, r0;
        if (NPASS == 3) {
            if (q < NA / 2) { base = p.A_hi; row = q * RPI; }
            else if (q < NA) { base = p.A_lo; row = (q - NA / 2) * RPI; }
            else if (q < NA + 12) { base = p.W_hi; row = (q - NA) * RPI; }
            else { base = p.W_lo; row = (q - NA - 12) * RPI; }
        } else {
            if (q < NA) { base = p.A_hi; row = q * RPI; }
            else { base = p.W_hi; row = (q - NA) * RPI; }
        }
        if (q < NA) { ld = p.lda; lim = p.M - 1; r0 = m0; } else { ld = p.ldw; lim = p.N - 1; r0 = n0; }
        row += lane / CPR;
        const int slot = lane % CPR;
        const int g = NPASS == 3 ? (slot ^ ((0 - (row >> 2)) & 3)) : (slot ^ (row & 7));
        src[j] = base + (size_t)min(r0 + row, lim) * ld + g * 8;
    }
    const int nt = p.K / BKK;
    auto issue = [&](int t, int buf) {
        char* st = smem + buf * BIG_STAGE_BYTES + wave * 1024;
#pragma unroll
        for (int j = 0; j < PPW; ++j) glds16(src[j] + t * BKK, st + j * WAVES * 1024);
    };

    issue(0, 0);
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < nt) issue(t + 1, buf ^ 1);
        const char* st = smem + buf * BIG_STAGE_BYTES;
        if constexpr (NPASS == 3) {
            const char *Ah = st, *Al = st + LO_A, *Wh = st + W_OFF, *Wl = st + W_OFF + LO_W;
            bf16x8 whf[6], wlf[6], ahf[4], alf[4];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int wo = swz64(wn * 96 + i * 16 + lr, lg);
                whf[i] = *reinterpret_cast<const bf16x8*>(Wh + wo);
                wlf[i] = *reinterpret_cast<const bf16x8*>(Wl + wo);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ao = swz64(wm * 64 + i * 16 + lr, lg);
                ahf[i] = *reinterpret_cast<const bf16x8*>(Ah + ao);
                alf[i] = *reinterpret_cast<const bf16x8*>(Al + ao);
            }
#pragma unroll
            for (int ni = 0; ni < 6; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    acc[ni][mi] = mma16<F16>(wlf[ni], ahf[mi], acc[ni][mi]);
                    acc[ni][mi] = mma16<F16>(whf[ni], alf[mi], acc[ni][mi]);
                    acc[ni][mi] = mma16<F16>(whf[ni], ahf[mi], acc[ni][mi]);
                }
        } else {
            const char *As = st, *Ws = st + W_OFF;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 wf[6], af[4];
                const int chunk = ks * 4 + lg;
#pragma unroll
                for (int i = 0; i < 6; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(Ws + swz128(wn * 96 + i * 16 + lr, chunk));
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + swz128(wm * 64 + i * 16 + lr, chunk));
#pragma unroll
                for (int ni = 0; ni < 6; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        acc[ni][mi] = mma16<F16>(wf[ni], af[mi], acc[ni][mi]);
            }
        }
    }
    if constexpr (F16) f16_saturate_mode();
    float amax = 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + lr;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < 6; ++ni) {
            const int n = n0 + wn * 96 + ni * 16 + lg * 4;
            if (n >= p.N) continue;
            epilogue4(p, m, n, acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3], amax);
        }
    }
    if constexpr (F16) raise_fault(p.fault, amax > DIDI_F16_MAX, DIDI_FAULT_FP16_RANGE);
}

// ------------------------------------------------------------------ big tile, staggered wave groups
// Same 256x192 tile, stage layout, swizzles and accumulation order as gemm_big_kernel<*, 4>, with the
// K-loop rescheduled (tools/gemm_lab.hip / gemm_lab1.hip hold the measurements).  s_memtime stamps of
// the in-phase loop showed where a K-tile's cycles went: all 8 waves issue their LDS-DMA at once and
// the CU's address path takes ~16 cycles per 1-KiB piece, so 56 pieces stall every wave for ~490
// cycles with the MFMA pipe idle, then all waves wait for their fragment reads together.  Here waves
// 0-3 (group X) and 4-7 (group Y; waves w and w+4 share a SIMD) run half a K-tile apart, so one
// group's fragment reads and DMA issue sit under the other group's MFMAs.  Slots of iteration i
// (each closed by a barrier every wave executes):
//   A_i : X reads tile i, then issues the weight pieces W(i+1)   | Y multiplies tile i-1
//   B1_i: X multiplies tile i (first third)                      | Y reads tile i   -> buffer i&1 is free
//   B2_i: X multiplies tile i (rest), waits for W(i+1)           | Y issues the activation pieces A(i+2)
//                                                                   into buffer i&1, waits for A(i+1)
// The activation pieces (the ones that can miss L2) get 1.5 K-tiles of flight, the weight pieces
// (L2 hits) one; every DMA wait is a counted vmcnt followed by a barrier before the first read.
__device__ __forceinline__ bool better(float v, int i, float bv, int bi);  // argmax order: larger value, then lower index

// helpers of the ATTN form (the same LDS image and transposed reads as attention.hip)
// LDS map of the cross-attention form (ATTN == 2): Q images at 0 and K / V buffers 1, 2 inside the stage area (free when the
// K-loop ends), K / V buffer 0, the key bias and the tile's bias values behind it
#define XA_KV0 (2 * BigGeom<4>::STAGE)
#define XA_KV1 49152
#define XA_MB (XA_KV0 + 32768)
#define XA_BIAS (XA_MB + 1024)
#define XA_LDS_BYTES (XA_BIAS + 768)
static_assert(XA_KV1 + 2 * 32768 == XA_KV0, "K / V buffers 1 and 2 end where the stages end");
__device__ __forceinline__ void xa_wait_vm(int n) {  // s_waitcnt vmcnt(n), n wave-uniform and <= 8
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}
typedef short v4s_ __attribute__((ext_vector_type(4)));
typedef short v8s_ __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int img64(int row, int ch) { return row * 128 + ((ch ^ (row & 7)) << 4); }
__device__ __forceinline__ bf16x8 lds_tr8(const char* p0, const char* p1) {
    const v4s_ a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_*)p0);
    const v4s_ b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_*)p1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// AMAX: the rounding form.  No output tile: each row's (max, index) over this tile's 192 vocabulary columns goes to
// p.partial[m][tile column]; the logits never leave the registers.
// ATTN: the fused self-attention form (didi_qkv_self_attention): BertSelfAttention's query / key / value projections AND its
// causal attention in one launch, for dialogs of exactly 64 target positions and 64-wide heads.  A 256-row band is then four
// whole dialogs and a 192-column tile is one head's [Q | K | V] (the weight rows of tile h are gathered as rows h*64.. of each
// of the three projections: no permuted copy of the weights), so the workgroup that holds the tile's accumulators holds
// everything the attention of (4 dialogs, head h) needs: Q, K, V go to LDS as fp16 (the stage buffers are free by then), and
// each wave runs two 16-query blocks of one dialog exactly as attn_kernel does.  The [M, 3H] projection planes (75 MB written
// and 75 MB read back per layer at the benchmark shape) and the attention launch disappear; the outputs are bit-identical to
// didi_gemm + didi_attention (same conversions, same MFMA and summation order).
// TN: the weight-gradient form (didi_gemm_wgrad): dW[n][k] = sum_m dC[m][n] A[m][k] reads BOTH operands as they lie in memory, the
// contraction index m being the ROW index of the sources (A_* = dC planes [K][lda], W_* = activation planes [K][ldw]; p.M / p.N
// are the output's rows / columns, p.K the number of source rows).  A K-tile is then [32 source rows][256 or 192 columns] in LDS
// (16-byte pieces along the columns: coalesced, no transposed copy of either operand is ever made), and the MFMA fragments come
// out of it through ds_read_b64_tr_b16, which hands lane lr column lr of a 4 x 16 block addressed row-wise by the 16 lanes.
// Lane (lr, lg)'s eight contraction indices are then {4 lg .. 4 lg + 3, 16 + 4 lg ..} in BOTH fragments instead of 8 lg .. 8 lg + 7:
// a permutation of the sum, nothing else.  Raw fp32 sums out (split-K partials), accumulator tile ni = 16 consecutive columns.
// TN = 3 is that form.  TN = 2 is the input-gradient form (didi_gemm_dgrad): dA[m][k] = sum_n dC[m][n] W[n][k] takes ONLY the weight
// side from a row-indexed source (W_* = the forward's own weight planes [K][ldw]); the activation side stays as it is, so the
// transposing reads take rows 8 lg .. 8 lg + 3 and 8 lg + 4 .. of the K-tile and the fragments pair index for index.  fp32 out
// (+ the fp32 residual), no bias.
template <int NPASS, bool F16, bool LN, bool AMAX = false, int ATTN = 0, int TN = 0>
__global__ void __launch_bounds__(512, 2) gemm_pp_kernel(GemmParams p) {
    using G = BigGeom<4>;
    static_assert(TN == 0 || ((TN == 2 || TN == 3) && NPASS != 1 && !LN && !AMAX && ATTN == 0), "the transposed-source forms: two- / three-product launches only");
    constexpr int BKK = NPASS == 1 ? 64 : 32;
    constexpr int CPR = BKK * 2 / 16, RPI = 64 / CPR;   // 16-byte chunks per LDS row, rows per DMA piece
    constexpr int NYP = NPASS == 2 ? 4 : 8;             // activation pieces per wave of group Y (fp16x2 stages no A_lo tile)
    constexpr int W_OFF = G::A_BYTES, LO_A = G::A_BYTES / 2, LO_W = 12288, STAGE = G::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef DIDI_STAMPS
    const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();  // the wave's first instructions (slot 13, 100 MHz)
#endif
    // every kernel argument the prologue and the loop need, fetched in ONE batch of scalar loads: left to itself hipcc loads them where
    // they are first used, four groups each behind its own s_waitcnt lgkmcnt(0) -- four scalar-memory latencies before the first DMA
    asm volatile("" ::"s"(p.A_hi), "s"(p.A_lo), "s"(p.W_hi), "s"(p.W_lo), "s"(p.lda), "s"(p.ldw), "s"(p.M), "s"(p.N), "s"(p.K), "s"(p.tiles_n),
                 "s"(p.bias), "s"(p.act), "s"(p.C_hi), "s"(p.C_lo), "s"(p.C_f32), "s"(p.ldc), "s"(p.k_split), "s"(p.dither));
    if constexpr (LN)
        asm volatile("" ::"s"(p.R_hi), "s"(p.R_lo), "s"(p.ldr), "s"(p.ln_gamma), "s"(p.ln_beta), "s"(p.ln_stats), "s"(p.ln_epoch), "s"(p.ln_eps), "s"(p.lo8),
                     "s"(p.fault));
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / p.tiles_n) * 256, n0 = (tile % p.tiles_n) * BBN;
    if constexpr (!LN && !AMAX && ATTN == 0) {
        if (p.k_split) {  // split-K form (didi_gemm_small on weight-gradient shapes): K columns [y k_split, ...), raw partial sums
            const int k0 = blockIdx.y * p.k_split;
            const size_t ka = TN == 3 ? (size_t)k0 * p.lda : k0, kw = TN ? (size_t)k0 * p.ldw : k0;
            p.A_hi += ka; p.A_lo += ka; p.W_hi += kw; p.W_lo += kw;
            p.K = min(p.k_split, p.K - k0);
            p.C_f32 += (size_t)blockIdx.y * p.M * p.ldc;
        }
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lg = lane >> 4;
    const int grp = wave >> 2, w4 = wave & 3;

    f32x4 acc[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // group Y: activation pieces q = w4 + 4j (j < 8) of 32; group X: weight pieces q = w4 + 4j (j < 6) of 24
    const u16* src[8];
    int qoff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = w4 + 4 * (grp ? (j < NYP ? j : NYP - 1) : (j < 6 ? j : 5));
        const u16* base;
        int row;
        if (NPASS != 1) {
            if (grp) { base = q < 16 ? p.A_hi : p.A_lo; row = (q & 15) * RPI; }
            else { base = q < 12 ? p.W_hi : p.W_lo; row = (q < 12 ? q : q - 12) * RPI; }
        } else {
            base = grp ? p.A_hi : p.W_hi; row = q * RPI;
        }
        if (TN == 3 || (TN == 2 && !grp)) {
            // piece q (1 KiB of LDS, lane l at byte 16 l) of a [32][256] activation-side or [32][192] weight-side K-tile: LDS row kk,
            // 16-byte position c_ in it.  The source chunk that goes there: activation side c_ ^ 2 (kk & 7), weight side the 32-byte
            // pair rotated by kk >> 1 (mod 12) -- the eight rows one ds_read_b64_tr_b16 touches per half wave then fall on eight
            // different bank groups (row strides 512 B and 384 B).  TN = 2 reads rows {0..3, 8..11} (+ 4, + 16) together: rotation
            // ((kk >> 1) & 1) + 2 ((kk >> 3) & 1)
            int kk, c;
            if (grp) { kk = 2 * (q & 15) + (lane >> 5); c = (lane & 31) ^ ((kk & 7) << 1); }
            else {
                const int idx = (q < 12 ? q : q - 12) * 64 + lane;
                kk = idx / 24;
                const int c_ = idx - kk * 24;
                int pr = (c_ >> 1) - (TN == 3 ? (kk >> 1) % 12 : ((kk >> 1) & 1) + 2 * ((kk >> 3) & 1));
                pr += pr < 0 ? 12 : 0;
                c = 2 * pr + (c_ & 1);
            }
            const int lim = grp ? p.M : p.N, o0 = grp ? m0 : n0;
            src[j] = base + (size_t)kk * (grp ? p.lda : p.ldw) + min(o0 + 8 * c, lim - 8);
            qoff[j] = (grp ? 0 : W_OFF) + q * 1024;
            continue;
        }
        row += lane / CPR;
        const int slot = lane % CPR;
        const int g = NPASS != 1 ? (slot ^ ((0 - (row >> 2)) & 3)) : (slot ^ (row & 7));
        // LDS weight row r' = 96*wn + 16*ni + rho holds output column 96*wn + 32*(ni>>1) + 8*(rho>>2) + 4*(ni&1) + (rho&3):
        // accumulator tiles 2k and 2k+1 then give every lane 8 CONSECUTIVE columns (epilogue8)
        const int wn_ = row / 96, rr_ = row % 96, ni_ = rr_ >> 4, rho_ = rr_ & 15;
        const int col = wn_ * 96 + 32 * (ni_ >> 1) + 8 * (rho_ >> 2) + 4 * (ni_ & 1) + (rho_ & 3);
        // ATTN: tile `tn` = head tn; its column c is row (c / 64) * H + tn * 64 + c % 64 of the stacked [query; key; value] weight
        const int wrow = ATTN == 1 ? (col >> 6) * (p.N / 3) + (tile % p.tiles_n) * 64 + (col & 63) : min(n0 + col, p.N - 1);
        src[j] = base + (size_t)(grp ? min(m0 + row, p.M - 1) * (size_t)p.lda : wrow * (size_t)p.ldw) + g * 8;
        qoff[j] = (grp ? 0 : W_OFF) + q * 1024;
    }
    const int nt = p.K / BKK;
    const size_t kstep = grp ? (TN == 3 ? (size_t)BKK * p.lda : BKK) : (TN ? (size_t)BKK * p.ldw : BKK);  // elements from one K-tile to the next
#define PP_ISSUE(T, BUF, N)                                                                          \
    do {                                                                                             \
        char* st_ = smem + (BUF) * STAGE;                                                            \
        _Pragma("unroll") for (int j = 0; j < (N); ++j) {                                            \
            if constexpr (TN != 0) glds16_raw(src[j] + (T) * kstep, st_ + qoff[j]); /* no compiler-made vmcnt(0) before the transposing reads */ \
            else glds16(src[j] + (T) * kstep, st_ + qoff[j]);                                        \
        }                                                                                            \
    } while (0)
    // TN: byte offsets of this lane's transposing reads.  The lane addresses row 4 lg + (lr >> 2) (and 16 below it) of the K-tile,
    // columns 4 (lr & 3) .. + 3 of the fragment's 16
    int ta[4], tw0[6], tw1[6];
    if constexpr (TN == 2) {
        const int kk = 8 * lg + (lr >> 2), sub = 8 * (lr & 1), hc = (lr & 3) >> 1;  // second read: 4 rows below (same rotation)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            int p0 = wn * 6 + i + ((kk >> 1) & 1) + 2 * ((kk >> 3) & 1);
            p0 -= p0 >= 12 ? 12 : 0;
            tw0[i] = kk * 384 + p0 * 32 + hc * 16 + sub;
            tw1[i] = tw0[i] + 4 * 384;
        }
    }
    if constexpr (TN == 3) {
        const int kk = 4 * lg + (lr >> 2), sub = 8 * (lr & 1), hc = (lr & 3) >> 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) ta[i] = kk * 512 + (((wm * 8 + i * 2 + hc) ^ ((kk & 7) << 1)) << 4) + sub;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            int p0 = wn * 6 + i + (kk >> 1), p1 = p0 + 8;  // kk < 16: kk >> 1 < 8, ((kk + 16) >> 1) % 12 = ((kk >> 1) + 8) % 12
            p0 -= p0 >= 12 ? 12 : 0;
            p1 -= p1 >= 24 ? 24 : p1 >= 12 ? 12 : 0;
            tw0[i] = kk * 384 + p0 * 32 + hc * 16 + sub;
            tw1[i] = (kk + 16) * 384 + p1 * 32 + hc * 16 + sub;
        }
    }

    // fragments: NPASS 1: fw[ks*6 + ni], fa[ks*4 + mi]; NPASS 3: fw[ni] hi, fw[6 + ni] lo, fa[mi] hi, fa[4 + mi] lo (NPASS 2: no fa lo)
    bf16x8 fw[12], fa[8];
#define PP_READ(ST)                                                                                        \
    do {                                                                                                   \
        if constexpr (TN != 0) {                                                                           \
            _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                \
                fw[i] = lds_tr8((ST) + W_OFF + tw0[i], (ST) + W_OFF + tw1[i]);                             \
                fw[6 + i] = lds_tr8((ST) + W_OFF + LO_W + tw0[i], (ST) + W_OFF + LO_W + tw1[i]);           \
            }                                                                                              \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
                if constexpr (TN == 3) {                                                                   \
                    fa[i] = lds_tr8((ST) + ta[i], (ST) + ta[i] + 8192);                                    \
                    if constexpr (NPASS == 3) fa[4 + i] = lds_tr8((ST) + LO_A + ta[i], (ST) + LO_A + ta[i] + 8192); \
                } else {                                                                                   \
                    const int ao = swz64(wm * 64 + i * 16 + lr, lg);                                       \
                    fa[i] = *reinterpret_cast<const bf16x8*>((ST) + ao);                                   \
                    if constexpr (NPASS == 3) fa[4 + i] = *reinterpret_cast<const bf16x8*>((ST) + LO_A + ao); \
                }                                                                                          \
            }                                                                                              \
        } else if constexpr (NPASS != 1) {                                                                 \
            _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                \
                const int wo = swz64(wn * 96 + i * 16 + lr, lg);                                           \
                fw[i] = *reinterpret_cast<const bf16x8*>((ST) + W_OFF + wo);                               \
                fw[6 + i] = *reinterpret_cast<const bf16x8*>((ST) + W_OFF + LO_W + wo);                    \
            }                                                                                              \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
                const int ao = swz64(wm * 64 + i * 16 + lr, lg);                                           \
                fa[i] = *reinterpret_cast<const bf16x8*>((ST) + ao);                                       \
                if constexpr (NPASS == 3) fa[4 + i] = *reinterpret_cast<const bf16x8*>((ST) + LO_A + ao);  \
            }                                                                                              \
        } else {                                                                                           \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                             \
                _Pragma("unroll") for (int i = 0; i < 6; ++i)                                              \
                    fw[ks * 6 + i] = *reinterpret_cast<const bf16x8*>((ST) + W_OFF + swz128(wn * 96 + i * 16 + lr, ks * 4 + lg)); \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                              \
                    fa[ks * 4 + i] = *reinterpret_cast<const bf16x8*>((ST) + swz128(wm * 64 + i * 16 + lr, ks * 4 + lg)); \
            }                                                                                              \
        }                                                                                                  \
    } while (0)
    // PART 0: what runs in slot B1, PART 1: the rest, PART 2: everything
#define PP_MFMA(PART)                                                                                      \
    do {                                                                                                   \
        if constexpr (NPASS != 1) {                                                                        \
            _Pragma("unroll") for (int ni = ((PART) == 1 ? 2 : 0); ni < ((PART) == 0 ? 2 : 6); ++ni)       \
                _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                         \
                    acc[ni][mi] = mma16<F16>(fw[6 + ni], fa[mi], acc[ni][mi]);                             \
                    if constexpr (NPASS == 3) acc[ni][mi] = mma16<F16>(fw[ni], fa[4 + mi], acc[ni][mi]);   \
                    acc[ni][mi] = mma16<F16>(fw[ni], fa[mi], acc[ni][mi]);                                 \
                }                                                                                          \
        } else {                                                                                           \
            _Pragma("unroll") for (int ni = ((PART) == 1 ? 4 : 0); ni < ((PART) == 0 ? 4 : 6); ++ni)       \
                _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                           \
                    acc[ni][mi] = mma16<F16>(fw[ni], fa[mi], acc[ni][mi]);                                 \
            if constexpr ((PART) != 0) {                                                                   \
                _Pragma("unroll") for (int ni = 0; ni < 6; ++ni)                                           \
                    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                       \
                        acc[ni][mi] = mma16<F16>(fw[6 + ni], fa[4 + mi], acc[ni][mi]);                     \
            }                                                                                              \
        }                                                                                                  \
    } while (0)

    // one-product GELU launches: the 8-KiB table rides behind the stages, one LDS-DMA piece per wave,
    // issued first so that every later counted wait covers it
    const bool use_lut = !TN && NPASS != 3 && p.act == DIDI_ACT_GELU;
    if (use_lut) glds16(reinterpret_cast<const u16*>(didi_gelu_table) + wave * 512 + lane * 8, smem + 2 * STAGE + wave * 1024);
    // the tile's 192 bias values ride along too (one 768-byte piece, wave 0): the epilogue then reads them from LDS instead of
    // opening with a dependent global load
    float* bias_lds = reinterpret_cast<float*>(smem + (ATTN == 2 ? XA_BIAS : 2 * STAGE + 8192));
    // cross-attention form: visible keys per dialog of the band (right-padded contexts: keys from the count on are never loaded nor
    // multiplied), and the K / V pieces of phase ph = (pair of dialogs ph / 3, head 3 tn + ph % 3) into buffer ph % 3: piece q of
    // 32 = (dialog q >> 4, K or V (q >> 3) & 1, rows 8 (q & 7) ..); wave w issues pieces 4 w .. 4 w + 3 and gets back how many
    // it issued (rows at or past ceil32(count) are not fetched: their key tiles are skipped)
    const int b0 = m0 >> 6;
    int xa_kc[4] = {64, 64, 64, 64}, xa_nkt[4] = {4, 4, 4, 4};
    float xa_mbv = 0.f;
    auto xa_buf = [](int i) { return i == 0 ? XA_KV0 : XA_KV1 + (i - 1) * 32768; };
    auto xa_issue_kv = [&](int ph) -> int {
        const int hf = ph >= 3, hh = ph - 3 * hf;
        char* buf = smem + xa_buf(ph % 3);
        int n = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = wave * 4 + j;
            const int d2 = q >> 4, isv = (q >> 3) & 1, r0 = (q & 7) * 8;
            const int d4 = hf * 2 + d2;
            const int rows = ((d4 == 0 ? xa_nkt[0] : d4 == 1 ? xa_nkt[1] : d4 == 2 ? xa_nkt[2] : xa_nkt[3]) + 1) >> 1 << 5;
            if (r0 < rows) {
                const int row = r0 + (lane >> 3), slot = lane & 7;
                const int bb = min(b0 + d4, p.attn_B - 1);
                const u16* s_ = (isv ? p.attn_V : p.attn_K) + ((size_t)bb * 64 + row) * p.attn_ldkv + (3 * (tile % p.tiles_n) + hh) * 64 + ((slot ^ (row & 7)) << 3);
                glds16_raw(s_, buf + q * 1024);
                ++n;
            }
        }
        return n;
    };
    if constexpr (ATTN == 2) {
#pragma unroll
        for (int d4 = 0; d4 < 4; ++d4) {
            if (p.attn_count && b0 + d4 < p.attn_B) {
                const int c = __builtin_amdgcn_readfirstlane(p.attn_count[b0 + d4]);
                if (c > 0 && c < 64) xa_kc[d4] = c;
            }
            xa_nkt[d4] = (xa_kc[d4] + 15) >> 4;
        }
        if (tid < 256) {  // waves 0-3: the additive key bias of (dialog wave, key lane): 0, NEG (masked) or -inf (past the count)
            const int kc_ = wave == 0 ? xa_kc[0] : wave == 1 ? xa_kc[1] : wave == 2 ? xa_kc[2] : xa_kc[3];
            if (lane >= kc_) xa_mbv = -INFINITY;
            else if (p.attn_mask && b0 + wave < p.attn_B && p.attn_mask[(size_t)(b0 + wave) * 64 + lane] == 0) xa_mbv = -3.4028234663852886e38f;
        }
    }
    if constexpr (ATTN == 1) {
        if (wave == 0 && lane < 48)
            glds16(reinterpret_cast<const u16*>(p.bias + ((4 * lane) >> 6) * (p.N / 3) + (tile % p.tiles_n) * 64 + ((4 * lane) & 63)),
                   reinterpret_cast<char*>(bias_lds));
    } else if (!AMAX && !TN && p.bias && wave == 0 && lane < 48 && n0 + 4 * lane < p.N)
        glds16(reinterpret_cast<const u16*>(p.bias + n0 + 4 * lane), reinterpret_cast<char*>(bias_lds));
    float* gamma_lds = bias_lds + 256;  // LayerNorm variant: gamma and beta of the tile's columns, waves 1 and 2
    float* beta_lds = bias_lds + 512;
    if constexpr (LN) {
        if (wave == 1 && lane < 48) glds16(reinterpret_cast<const u16*>(p.ln_gamma + n0 + 4 * lane), reinterpret_cast<char*>(gamma_lds));
        if (wave == 2 && lane < 48) glds16(reinterpret_cast<const u16*>(p.ln_beta + n0 + 4 * lane), reinterpret_cast<char*>(beta_lds));
    }
    if constexpr (ATTN == 2) (void)xa_issue_kv(0);  // behind the stages: in flight under the K-loop (whose waits cover it)
    PP_STAMP(0);
    // tile 0 must have landed before the first read; tile 1 stays in flight (first waits: end of B2_0)
    if (grp == 0) {
        PP_ISSUE(0, 0, 6);
        if (nt > 1) { PP_ISSUE(1, 1, 6); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        PP_ISSUE(0, 0, NYP);
        if (nt > 1) {
            PP_ISSUE(1, 1, NYP);
            if constexpr (NYP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    PP_STAMP(1);  // first K-tile landed: the loop starts
    if (grp == 0) {
        for (int i = 0; i < nt; ++i) {
            const char* st = smem + (i & 1) * STAGE;
            PP_READ(st);
            if (i >= 1 && i + 1 < nt) PP_ISSUE(i + 1, (i + 1) & 1, 6);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();  // end of A_i
            __builtin_amdgcn_s_setprio(1);
            PP_MFMA(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();  // end of B1_i
            PP_MFMA(1);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // end of B2_i
        }
    } else {
        for (int i = 0; i < nt; ++i) {
            if (i >= 1) {
                __builtin_amdgcn_s_setprio(1);
                PP_MFMA(2);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_barrier();  // end of A_i
            const char* st = smem + (i & 1) * STAGE;
            PP_READ(st);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();  // end of B1_i
            if (i + 2 < nt) {
                PP_ISSUE(i + 2, i & 1, NYP);
                if constexpr (NYP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();  // end of B2_i
        }
        PP_MFMA(2);
    }
#undef PP_ISSUE
#undef PP_READ
#undef PP_MFMA
    PP_STAMP(2);  // K-loop done (this wave group)
    if constexpr (TN != 0) {  // raw sums (+ fp32 residual): acc[ni][mi][r] = C[m0 + 64 wm + 16 mi + lr][n0 + 96 wn + 16 ni + 4 lg + r]
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wm * 64 + mi * 16 + lr;
            float4 rr[6];
#pragma unroll
            for (int ni = 0; ni < 6; ++ni) {  // the residual's six loads first, then the stores
                const int n = n0 + wn * 96 + ni * 16 + 4 * lg;
                rr[ni] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (TN == 2 && p.R_f32 && m < p.M && n < p.N) rr[ni] = *reinterpret_cast<const float4*>(p.R_f32 + (size_t)m * p.ldr + n);
            }
#pragma unroll
            for (int ni = 0; ni < 6; ++ni) {
                const int n = n0 + wn * 96 + ni * 16 + 4 * lg;
                if (m < p.M && n < p.N)  // N % 8 == 0: n + 3 < N too
                    *reinterpret_cast<float4*>(p.C_f32 + (size_t)m * p.ldc + n) =
                        make_float4(acc[ni][mi][0] + rr[ni].x, acc[ni][mi][1] + rr[ni].y, acc[ni][mi][2] + rr[ni].z, acc[ni][mi][3] + rr[ni].w);
            }
        }
        return;
    }
    if constexpr (ATTN == 2) {
        // ---- BertCrossAttention: the tile's 192 columns are the queries of heads 3 tn .. 3 tn + 2 for the band's four dialogs; K and
        // V come from the hoisted cache [B * 64][ldkv] (K at column head * 64, V a plane offset further).  Six phases of (two
        // dialogs, one head), three K / V buffers of 32 KB: phase 0's arrives under the K-loop (its buffer lies behind the stages),
        // phases 1 and 2 are issued when the loop ends, phase i + 2 when phase i starts; every wait is a COUNTED vmcnt (the
        // outputs of phase i - 1 are stored at the start of phase i, so neither a store nor a younger K / V piece is ever
        // waited for).  Each wave runs ONE 16-query block per phase exactly as attn_kernel does (masked / padded keys, key
        // counts, summation order: bit-identical outputs).
        constexpr float NEG = -3.4028234663852886e38f;
        f16_saturate_mode();
        const int dith = NPASS == 1 ? p.dither : 0;
        const int dq = dith ? dith : DIDI_RNE_AS_DITHER;
        const int tn = tile % p.tiles_n;
        char* Qimg = smem;  // [2 dialogs][3 heads][64 rows][128 B]
        float* mb = reinterpret_cast<float*>(smem + XA_MB);  // [4 dialogs][64 keys] additive key bias
        if (tid < 256) mb[tid] = xa_mbv;
        const int n1 = xa_issue_kv(1), n2 = xa_issue_kv(2);
        // the wave's queries (+ bias) as fp16, 8 consecutive head dims per register quad; dialogs 0, 1 go to LDS now, 2, 3 wait in
        // registers (48 instead of the accumulators' 96) until phase 3
        uint4 qp[4][3];
        {
            float amax = 0.f;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int nl = wn * 96 + 32 * k + 8 * lg;
                    const float4 c0 = *reinterpret_cast<const float4*>(bias_lds + nl), c1 = *reinterpret_cast<const float4*>(bias_lds + nl + 4);
                    const float v[8] = {acc[2 * k][mi][0] + c0.x, acc[2 * k][mi][1] + c0.y, acc[2 * k][mi][2] + c0.z, acc[2 * k][mi][3] + c0.w,
                                        acc[2 * k + 1][mi][0] + c1.x, acc[2 * k + 1][mi][1] + c1.y, acc[2 * k + 1][mi][2] + c1.z, acc[2 * k + 1][mi][3] + c1.w};
                    amax = amax2(amax2(amax2(amax2(amax, v[0], v[1]), v[2], v[3]), v[4], v[5]), v[6], v[7]);
                    const uint2 h0 = pack4_hi_dither(v, dq), h1 = pack4_hi_dither(v + 4, dq);
                    qp[mi][k] = make_uint4(h0.x, h0.y, h1.x, h1.y);
                }
            raise_fault(p.fault, amax > DIDI_F16_MAX, DIDI_FAULT_FP16_RANGE);
        }
        auto dump_q = [&]() {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int nl = wn * 96 + 32 * k + 8 * lg;
                    *reinterpret_cast<uint4*>(Qimg + (((wm & 1) * 3 + (nl >> 6)) * 64) * 128 + img64(mi * 16 + lr, (nl & 63) >> 3)) = qp[mi][k];
                }
        };
        if (wm < 2) dump_q();
        const int trow = 4 * lg + (lr >> 2), tsub = 8 * (lr & 1);
        const int d2 = wave >> 2, qb = wave & 3;  // this wave's dialog within the phase's pair, and its query block
        const int q = qb * 16 + lr;
#ifdef DIDI_STAMPS
        unsigned long long xt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // kept in registers, written at the end: a stamp's store would count in vmcnt
        xt[6] = __builtin_amdgcn_s_memtime();
#endif
        uint2 outv[4];
        u16* optr = nullptr;  // where the previous phase's outputs go (lane's row and head; null: dialog past the batch)
        int s_prev = 0, s_prev2 = 0, n_next = n2;  // stores of phase ph - 1 / ph - 2, K / V pieces of phase ph + 1 (this wave's)
        for (int ph = 0; ph < 6; ++ph) {
            const int hf = ph >= 3, hh = ph - 3 * hf;
            if (ph == 3) {  // second pair of dialogs: its queries replace the first pair's once every wave is done with them
                __builtin_amdgcn_s_barrier();
                if (wm >= 2) dump_q();
            }
            // this wave's K / V pieces of phase ph have landed once everything issued after them is all that is in flight:
            // the stores of phase ph - 2's outputs and the pieces of phase ph + 1
            xa_wait_vm(ph == 0 ? n1 + n2 : s_prev2 + n_next);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // ... and every other wave's: phase ph's K / V, the Q images, the key bias are visible
#ifdef DIDI_STAMPS
            xt[ph] = __builtin_amdgcn_s_memtime();
#endif
            if (ph > 0) {
                if (optr) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<uint2*>(optr + dt * 16) = outv[dt];
                }
                s_prev2 = s_prev;
                n_next = ph + 2 < 6 ? xa_issue_kv(ph + 2) : 0;  // into the buffer phase ph - 1 read
            } else {
                n_next = n2;
            }
            const int d4 = hf * 2 + d2;
            const int nkt = d4 == 0 ? xa_nkt[0] : d4 == 1 ? xa_nkt[1] : d4 == 2 ? xa_nkt[2] : xa_nkt[3];  // wave-uniform
            const int nks = (nkt + 1) >> 1;
            const char* Qs = Qimg + ((d2 * 3 + hh) * 64) * 128;
            const char* Ks = smem + xa_buf(ph % 3) + d2 * 16384;
            const char* Vs = Ks + 8192;
            const float* mbd = mb + d4 * 64;
            const bf16x8 qh0 = *reinterpret_cast<const bf16x8*>(Qs + img64(q, lg));
            const bf16x8 qh1 = *reinterpret_cast<const bf16x8*>(Qs + img64(q, 4 + lg));
            float sc[4][4];
            float rmax = NEG, rsum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (kt < nkt) {
                    f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
                    s4 = mma16<true>(*reinterpret_cast<const bf16x8*>(Ks + img64(kt * 16 + lr, lg)), qh0, s4);
                    s4 = mma16<true>(*reinterpret_cast<const bf16x8*>(Ks + img64(kt * 16 + lr, 4 + lg)), qh1, s4);
                    const float4 m4 = *reinterpret_cast<const float4*>(mbd + kt * 16 + lg * 4);
                    sc[kt][0] = fmaf(s4[0], p.attn_scale, m4.x);
                    sc[kt][1] = fmaf(s4[1], p.attn_scale, m4.y);
                    sc[kt][2] = fmaf(s4[2], p.attn_scale, m4.z);
                    sc[kt][3] = fmaf(s4[3], p.attn_scale, m4.w);
                    rmax = fmaxf(rmax, fmaxf(fmaxf(sc[kt][0], sc[kt][1]), fmaxf(sc[kt][2], sc[kt][3])));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[kt][r] = -INFINITY;
                }
            }
            rmax = fmaxf(rmax, __shfl_xor(rmax, 16, 64));
            rmax = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
            v8s_ ph8[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                ph8[s] = v8s_{0, 0, 0, 0, 0, 0, 0, 0};
                if (s < nks) {
                    float pr[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        pr[e] = __builtin_amdgcn_exp2f(sc[2 * s + (e >> 2)][e & 3] - rmax);
                        rsum += pr[e];
                    }
                    if (dith) {
                        const uint2 a0 = pack4_hi_dither(pr, dith), a1 = pack4_hi_dither(pr + 4, dith);
                        ph8[s] = __builtin_bit_cast(v8s_, make_uint4(a0.x, a0.y, a1.x, a1.y));
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ph8[s][e] = (short)f16_bits(pr[e]);
                    }
                }
            }
            rsum += __shfl_xor(rsum, 16, 64);
            rsum += __shfl_xor(rsum, 32, 64);
            const float rinv = 1.0f / rsum;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 o4 = {0.f, 0.f, 0.f, 0.f};
                const int cch = 2 * dt + ((lr & 3) >> 1);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (s < nks) {
                        const bf16x8 vh = lds_tr8(Vs + img64(32 * s + trow, cch) + tsub, Vs + img64(32 * s + 16 + trow, cch) + tsub);
                        o4 = mma16<true>(vh, __builtin_bit_cast(bf16x8, ph8[s]), o4);
                    }
                }
                float o[4] = {o4[0] * rinv, o4[1] * rinv, o4[2] * rinv, o4[3] * rinv};
                asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));  // product, THEN conversion (see the self-attention form)
                outv[dt] = dith ? pack4_hi_dither(o, dith) : pack4_hi(o, 1);
            }
            const bool in_batch = m0 + d4 * 64 < p.M;  // wave-uniform (M % 64 == 0)
            optr = in_batch ? p.C_hi + (size_t)(m0 + d4 * 64 + q) * p.ldc + (3 * tn + hh) * 64 + lg * 4 : nullptr;
            s_prev = in_batch ? 4 : 0;
#ifdef DIDI_STAMPS
            if (ph == 0) xt[7] = __builtin_amdgcn_s_memtime();
            if (ph == 5) xt[8] = __builtin_amdgcn_s_memtime();
#endif
        }
        if (optr) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<uint2*>(optr + dt * 16) = outv[dt];
        }
#ifdef DIDI_STAMPS
        // slots: 3 queries converted (before phase 0's wait), 8 + ph after phase ph's barrier, 4 / 5 end of phase 0's / 5's arithmetic
        if (p.stamps && lane == 0 && (wave & 3) == 0) {
            unsigned long long* s_ = p.stamps + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 16;
            s_[3] = xt[6]; s_[4] = xt[7]; s_[5] = xt[8];
            for (int i = 0; i < 6; ++i) s_[8 + i] = xt[i];
        }
#endif
        PP_STAMP(6);
#ifdef DIDI_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        PP_STAMP(7);
        return;
    }
    if constexpr (ATTN == 1) {
        constexpr float NEG = -3.4028234663852886e38f;
        f16_saturate_mode();
        const int dith = NPASS == 1 ? p.dither : 0;
        const int dq = dith ? dith : DIDI_RNE_AS_DITHER;
        // ---- Q, K, V of (dialog wm, this head) -> LDS images [dialog][Q, K, V][64 rows][128 B], 16-byte chunks XOR-swizzled
        float amax = 0.f;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int row = mi * 16 + lr;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int nl = wn * 96 + 32 * k + 8 * lg;
                const float4 b0 = *reinterpret_cast<const float4*>(bias_lds + nl), b1 = *reinterpret_cast<const float4*>(bias_lds + nl + 4);
                const float v[8] = {acc[2 * k][mi][0] + b0.x, acc[2 * k][mi][1] + b0.y, acc[2 * k][mi][2] + b0.z, acc[2 * k][mi][3] + b0.w,
                                    acc[2 * k + 1][mi][0] + b1.x, acc[2 * k + 1][mi][1] + b1.y, acc[2 * k + 1][mi][2] + b1.z, acc[2 * k + 1][mi][3] + b1.w};
                amax = amax2(amax2(amax2(amax2(amax, v[0], v[1]), v[2], v[3]), v[4], v[5]), v[6], v[7]);
                const uint2 h0 = pack4_hi_dither(v, dq), h1 = pack4_hi_dither(v + 4, dq);
                *reinterpret_cast<uint4*>(smem + ((wm * 3 + (nl >> 6)) * 64) * 128 + img64(row, (nl & 63) >> 3)) = make_uint4(h0.x, h0.y, h1.x, h1.y);
            }
        }
        raise_fault(p.fault, amax > DIDI_F16_MAX, DIDI_FAULT_FP16_RANGE);
        __syncthreads();
        PP_STAMP(3);
        // ---- causal attention: wave w takes dialog w / 2 and the query blocks {0, 3} or {1, 2} (1 + 4 or 2 + 3 key tiles)
        const int dl = wave >> 1;
        const char* Qs = smem + (dl * 3) * 8192;
        const char* Ks = Qs + 8192;
        const char* Vs = Ks + 8192;
        const int trow = 4 * lg + (lr >> 2), tsub = 8 * (lr & 1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int qb = (wave & 1) ? 1 + u : 3 * u;  // wave-uniform
            const int nkt = qb + 1, nks = (nkt + 1) >> 1;
            const int q = qb * 16 + lr;
            const bf16x8 qh0 = *reinterpret_cast<const bf16x8*>(Qs + img64(q, lg));
            const bf16x8 qh1 = *reinterpret_cast<const bf16x8*>(Qs + img64(q, 4 + lg));
            float sc[4][4];
            float rmax = NEG, rsum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (kt < nkt) {
                    f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
                    s4 = mma16<true>(*reinterpret_cast<const bf16x8*>(Ks + img64(kt * 16 + lr, lg)), qh0, s4);
                    s4 = mma16<true>(*reinterpret_cast<const bf16x8*>(Ks + img64(kt * 16 + lr, 4 + lg)), qh1, s4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sc[kt][r] = fmaf(s4[r], p.attn_scale, p.attn_mask0);
                        if (kt == qb && kt * 16 + lg * 4 + r > q) sc[kt][r] = fminf(sc[kt][r], NEG);  // future keys (diagonal tile only)
                    }
                    rmax = fmaxf(rmax, fmaxf(fmaxf(sc[kt][0], sc[kt][1]), fmaxf(sc[kt][2], sc[kt][3])));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[kt][r] = -INFINITY;
                }
            }
            rmax = fmaxf(rmax, __shfl_xor(rmax, 16, 64));
            rmax = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
            v8s_ ph[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                ph[s] = v8s_{0, 0, 0, 0, 0, 0, 0, 0};
                if (s < nks) {
                    float pr[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        pr[e] = __builtin_amdgcn_exp2f(sc[2 * s + (e >> 2)][e & 3] - rmax);
                        rsum += pr[e];
                    }
                    if (dith) {
                        const uint2 a0 = pack4_hi_dither(pr, dith), a1 = pack4_hi_dither(pr + 4, dith);
                        ph[s] = __builtin_bit_cast(v8s_, make_uint4(a0.x, a0.y, a1.x, a1.y));
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ph[s][e] = (short)f16_bits(pr[e]);
                    }
                }
            }
            rsum += __shfl_xor(rsum, 16, 64);
            rsum += __shfl_xor(rsum, 32, 64);
            const float rinv = 1.0f / rsum;
            const int m = m0 + dl * 64 + q;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 o4 = {0.f, 0.f, 0.f, 0.f};
                const int cch = 2 * dt + ((lr & 3) >> 1);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (s < nks) {
                        const bf16x8 vh = lds_tr8(Vs + img64(32 * s + trow, cch) + tsub, Vs + img64(32 * s + 16 + trow, cch) + tsub);
                        o4 = mma16<true>(vh, __builtin_bit_cast(bf16x8, ph[s]), o4);
                    }
                }
                if (m < p.M) {
                    // the fp32 products are made opaque before the 16-bit conversion: hipcc otherwise fuses product and conversion into one
                    // v_fma_mixlo_f16 (a single rounding) in SOME instantiations, and the fused QKV form and attn_kernel would differ by an ulp in 1 of 50 000 values
                    float o[4] = {o4[0] * rinv, o4[1] * rinv, o4[2] * rinv, o4[3] * rinv};
                    asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
                    *reinterpret_cast<uint2*>(p.C_hi + (size_t)m * p.ldc + (tile % p.tiles_n) * 64 + dt * 16 + lg * 4) =
                        dith ? pack4_hi_dither(o, dith) : pack4_hi(o, 1);
                }
            }
        }
        PP_STAMP(6);
#ifdef DIDI_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        PP_STAMP(7);
        return;
    }
    if constexpr (AMAX) {
        // lane (lr, lg) of wave (wm, wn) holds, for row wm*64 + mi*16 + lr, the columns wn*96 + 32k + 8lg + 4h + r of tile ni = 2k + h
        float best[4];
        int bidx[4];
        // a lane walks its 24 columns in increasing order, so a strict '>' keeps the lowest index among equal maxima; tiles that
        // lie wholly inside the vocabulary (all but the last column tile) skip the per-element bound check
        const int nb = n0 + wn * 96 + 8 * lg;
        const bool inside = n0 + BBN <= p.N;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            best[mi] = -INFINITY; bidx[mi] = 0x7fffffff;
#pragma unroll
            for (int ni = 0; ni < 6; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int off = 32 * (ni >> 1) + 4 * (ni & 1) + r;
                    const float v = acc[ni][mi][r];
                    if ((inside || nb + off < p.N) && v > best[mi]) { best[mi] = v; bidx[mi] = nb + off; }
                }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float ov = __shfl_xor(best[mi], o, 64);
                const int oi = __shfl_xor(bidx[mi], o, 64);
                if (better(ov, oi, best[mi], bidx[mi])) { best[mi] = ov; bidx[mi] = oi; }
            }
        }
        float* rv = reinterpret_cast<float*>(smem);        // [256] the wn = 1 half's result per row (the stages are free now)
        int* ri = reinterpret_cast<int*>(smem + 1024);     // [256]
        if (wn == 1 && lg == 0) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) { rv[wm * 64 + mi * 16 + lr] = best[mi]; ri[wm * 64 + mi * 16 + lr] = bidx[mi]; }
        }
        __syncthreads();
        if (wn == 0 && lg == 0) {
            const int tn = tile % p.tiles_n;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int ml = wm * 64 + mi * 16 + lr, m = m0 + ml;
                if (m < p.M) {
                    float v = best[mi]; int i = bidx[mi];
                    if (better(rv[ml], ri[ml], v, i)) { v = rv[ml]; i = ri[ml]; }
                    p.partial[(size_t)m * p.nchunks + tn] = make_float2(v, __int_as_float(i));
                }
            }
        }
        return;
    }
    const float2* gelu_tbl = use_lut ? reinterpret_cast<const float2*>(smem + 2 * STAGE) : nullptr;
    // residual planes: all 12 loads of half a wave tile are issued before the first use (the fragment
    // registers are dead by now), instead of one exposed memory round trip per 8 columns
    const int pre = p.R_hi != nullptr && p.R_f32 == nullptr;
    const int dith = NPASS == 1 && F16 ? p.dither : 0;  // activation-plane rounding dither: one-product fp16 launches only
    if constexpr (F16) f16_saturate_mode();
    if constexpr (!LN && F16) {
        // Straight-line epilogue for the sampler's own launches (qkv, cross-attention query, FFN up): full tile, bias from LDS,
        // no residual, planes out only.  The general epilogue8 below decides all of that per 8 values with scalar branches and
        // converts every value to BOTH plane formats behind a select (p.f16 is a run-time field there): stamps put it at
        // 7.4 us per tile against 11 us for the whole 12-tile K-loop (profiles/r03_ln_stamps.log).  Here the format is the
        // template's, the conversions are packed (f16x2_dither: 1.5 VALU per value) and nothing branches inside the loops.
        const bool fast = p.bias && !p.R_f32 && !p.R_hi && !p.C_f32 && p.C_hi && m0 + 256 <= p.M && n0 + BBN <= p.N &&
                          (p.act == DIDI_ACT_NONE || use_lut);
        if (fast) {
            const int d = dith ? dith : DIDI_RNE_AS_DITHER;
            float amax = 0.f;
            auto run = [&](auto gelu_c, auto lo_c) {
                constexpr bool GELU = decltype(gelu_c)::value, LO = decltype(lo_c)::value;
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const size_t row = (size_t)(m0 + wm * 64 + mi * 16 + lr) * p.ldc + n0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int nl = wn * 96 + 32 * k + 8 * lg;
                        const float4 b0 = *reinterpret_cast<const float4*>(bias_lds + nl), b1 = *reinterpret_cast<const float4*>(bias_lds + nl + 4);
                        float v[8] = {acc[2 * k][mi][0] + b0.x, acc[2 * k][mi][1] + b0.y, acc[2 * k][mi][2] + b0.z, acc[2 * k][mi][3] + b0.w,
                                      acc[2 * k + 1][mi][0] + b1.x, acc[2 * k + 1][mi][1] + b1.y, acc[2 * k + 1][mi][2] + b1.z, acc[2 * k + 1][mi][3] + b1.w};
                        if constexpr (GELU) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] = gelu_lut(v[i], gelu_tbl);
                        }
                        amax = amax2(amax2(amax2(amax2(amax, v[0], v[1]), v[2], v[3]), v[4], v[5]), v[6], v[7]);
                        if constexpr (LO) {
                            uint2 h0, l0, h1, l1;
                            split4_dither(v, h0, l0, d);
                            split4_dither(v + 4, h1, l1, d);
                            *reinterpret_cast<uint4*>(p.C_hi + row + nl) = make_uint4(h0.x, h0.y, h1.x, h1.y);
                            *reinterpret_cast<uint4*>(p.C_lo + row + nl) = make_uint4(l0.x, l0.y, l1.x, l1.y);
                        } else {
                            const uint2 h0 = pack4_hi_dither(v, d), h1 = pack4_hi_dither(v + 4, d);
                            *reinterpret_cast<uint4*>(p.C_hi + row + nl) = make_uint4(h0.x, h0.y, h1.x, h1.y);
                        }
                    }
                }
            };
            using T_ = std::true_type;
            using F_ = std::false_type;
            if (use_lut) { if (p.C_lo) run(T_{}, T_{}); else run(T_{}, F_{}); }
            else { if (p.C_lo) run(F_{}, T_{}); else run(F_{}, F_{}); }
            PP_STAMP(6);
#ifdef DIDI_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            PP_STAMP(7);
            raise_fault(p.fault, amax > DIDI_F16_MAX, DIDI_FAULT_FP16_RANGE);
            return;
        }
    }
    if constexpr (!LN) {
        float amax = 0.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint4 rh[2][3], rl[2][3];
#pragma unroll
            for (int mj = 0; mj < 2; ++mj)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    rh[mj][k] = uint4{0, 0, 0, 0};
                    rl[mj][k] = uint4{0, 0, 0, 0};
                    const int m = m0 + wm * 64 + (half * 2 + mj) * 16 + lr;
                    const int n = n0 + wn * 96 + 32 * k + 8 * lg;
                    if (pre && m < p.M && n + 8 <= p.N) {
                        rh[mj][k] = *reinterpret_cast<const uint4*>(p.R_hi + (size_t)m * p.ldr + n);
                        if (p.R_lo) rl[mj][k] = *reinterpret_cast<const uint4*>(p.R_lo + (size_t)m * p.ldr + n);
                    }
                }
#pragma unroll
            for (int mj = 0; mj < 2; ++mj) {
                const int mi = half * 2 + mj;
                const int m = m0 + wm * 64 + mi * 16 + lr;
                if (m >= p.M) continue;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int n = n0 + wn * 96 + 32 * k + 8 * lg;
                    if (n + 8 <= p.N) epilogue8(p, m, n, acc[2 * k][mi], acc[2 * k + 1][mi], gelu_tbl, amax, pre, rh[mj][k], rl[mj][k], bias_lds + (n - n0),
                                                   F16 ? (dith ? dith : DIDI_RNE_AS_DITHER) : 0);  // fp16 planes of this kernel round one way in every tile, edge tiles included
                    else if (n < p.N) epilogue4(p, m, n, acc[2 * k][mi][0], acc[2 * k][mi][1], acc[2 * k][mi][2], acc[2 * k][mi][3], amax);
                }
            }
        }
        PP_STAMP(6);  // every store issued
#ifdef DIDI_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        PP_STAMP(7);  // ... and drained
        if constexpr (F16) raise_fault(p.fault, p.C_hi != nullptr && amax > DIDI_F16_MAX, DIDI_FAULT_FP16_RANGE);
    } else {
        // ---- fused LayerNorm: the tiles_n (= 4) workgroups of a 256-row band each hold 192 of the 768 columns
        // of every row.  Each computes exact local row statistics (sum, M2 about its local mean), publishes them,
        // collects its band mates', combines the four partials (Chan's formula) and normalises its own slice
        // from registers: the pre-norm tensor never goes to memory.  Host side: a launch holds at most one
        // workgroup per CU, so all mates are co-resident; the wait is bounded anyway.
        float* red = reinterpret_cast<float*>(smem);          // [2 wn][256 rows] partial sums
        float* red2 = red + 512;                              // [2 wn][256 rows] partial M2
        float2* mr = reinterpret_cast<float2*>(red2 + 512);   // [256] (mean, rstd)
        // 1. v = acc + bias + residual, in place
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint4 rh[2][3], rl[2][3];
            // the three forms of the remainder plane (bytes, 16 bit, none) are three straight-line runs of loads: a branch PER load made
            // hipcc put s_waitcnt vmcnt(0) between the loads (registers shared across the paths), one memory latency each
            const int lo_form = (F16 && (p.lo8 & 1)) ? 2 : p.R_lo ? 1 : 0;
            auto r_at = [&](int mj, int k) {
                const int m = min(m0 + wm * 64 + (half * 2 + mj) * 16 + lr, p.M - 1);
                return (size_t)m * p.ldr + (n0 + wn * 96 + 32 * k + 8 * lg);
            };
            if (lo_form == 2) {
#pragma unroll
                for (int mj = 0; mj < 2; ++mj)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const size_t o = r_at(mj, k);
                        rh[mj][k] = *reinterpret_cast<const uint4*>(p.R_hi + o);
                        const uint2 b8 = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(p.R_lo) + o);  // 8 B per lane; x 1 / 256 below
                        rl[mj][k] = make_uint4(b8.x, b8.y, 0, 0);
                    }
            } else if (lo_form == 1) {
#pragma unroll
                for (int mj = 0; mj < 2; ++mj)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const size_t o = r_at(mj, k);
                        rh[mj][k] = *reinterpret_cast<const uint4*>(p.R_hi + o);
                        rl[mj][k] = *reinterpret_cast<const uint4*>(p.R_lo + o);
                    }
            } else {  // the residual is its hi plane
#pragma unroll
                for (int mj = 0; mj < 2; ++mj)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        rh[mj][k] = *reinterpret_cast<const uint4*>(p.R_hi + r_at(mj, k));
                        rl[mj][k] = uint4{0, 0, 0, 0};
                    }
            }
            const float rl_scale = F16 && (p.lo8 & 1) ? 1.0f / DIDI_LO8_SCALE : 1.0f;
#pragma unroll
            for (int mj = 0; mj < 2; ++mj) {
                const int mi = half * 2 + mj;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int n = n0 + wn * 96 + 32 * k + 8 * lg;
                    const float4 b0 = *reinterpret_cast<const float4*>(bias_lds + (n - n0)), b1 = *reinterpret_cast<const float4*>(bias_lds + (n - n0) + 4);
                    const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                    float rH[8], rL[8];
                    unpack4(make_uint2(rh[mj][k].x, rh[mj][k].y), rH, F16);  // the template's format: p.f16 would convert both ways
                    unpack4(make_uint2(rh[mj][k].z, rh[mj][k].w), rH + 4, F16);
                    if (F16 && (p.lo8 & 1)) {
                        unpack4(lo8_unpack4(rl[mj][k].x), rL, 1);
                        unpack4(lo8_unpack4(rl[mj][k].y), rL + 4, 1);
                    } else {
                        unpack4(make_uint2(rl[mj][k].x, rl[mj][k].y), rL, F16);
                        unpack4(make_uint2(rl[mj][k].z, rl[mj][k].w), rL + 4, F16);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[2 * k][mi][i] = fmaf(rL[i], rl_scale, (acc[2 * k][mi][i] + b[i]) + rH[i]);  // rl_scale = 1: the plain sum, bit for bit
                        acc[2 * k + 1][mi][i] = fmaf(rL[4 + i], rl_scale, (acc[2 * k + 1][mi][i] + b[4 + i]) + rH[4 + i]);
                    }
                }
            }
        }
        PP_STAMP(3);  // bias + residual added: both halves' residual planes have arrived
        // 2. local sums over this workgroup's 192 columns
        // (the four row groups' cross-lane steps are issued together: one LDS-crossbar round trip per step, not one per row group)
        float ml[4], ts[4], tu[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            ts[mi] = 0.f;
#pragma unroll
            for (int ni = 0; ni < 6; ++ni) ts[mi] += (acc[ni][mi][0] + acc[ni][mi][1]) + (acc[ni][mi][2] + acc[ni][mi][3]);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) tu[mi] = __shfl_xor(ts[mi], 16, 64);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) ts[mi] += tu[mi];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) tu[mi] = __shfl_xor(ts[mi], 32, 64);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            ts[mi] += tu[mi];
            if (lg == 0) red[wn * 256 + wm * 64 + mi * 16 + lr] = ts[mi];
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int row = wm * 64 + mi * 16 + lr;
            ml[mi] = (red[row] + red[256 + row]) * (1.0f / BBN);
        }
        // 3. local M2 about the local mean
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            float q = 0.f;
#pragma unroll
            for (int ni = 0; ni < 6; ++ni)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d = acc[ni][mi][i] - ml[mi];
                    q += d * d;
                }
            ts[mi] = q;
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) tu[mi] = __shfl_xor(ts[mi], 16, 64);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) ts[mi] += tu[mi];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) tu[mi] = __shfl_xor(ts[mi], 32, 64);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            ts[mi] += tu[mi];
            if (lg == 0) red2[wn * 256 + wm * 64 + mi * 16 + lr] = ts[mi];
        }
        __syncthreads();
        PP_STAMP(4);  // local statistics done
        // 4.-6. exchange with the band mates, placement-independent and fence-free: every statistic travels as an
        // 8-byte {tag = epoch, value} granule written and polled with relaxed agent-scope atomics (write-through
        // stores, L1-bypassing loads; cdna_hip_programming.md Guideline 16, recipe R2: the data is the flag).
        // Thread t owns row t: it publishes this tile's (sum, M2) and sweeps the 2 x tiles_n granules of its row
        // until every tag matches.  Stale granules carry an older epoch (the buffer is zeroed once per batch of
        // launches, epochs count up from 1).  The sweep is bounded: a lost mate gives NaN rows, never a hung GPU.
        const int band = tile / p.tiles_n, tn = tile % p.tiles_n;
        typedef __attribute__((address_space(1))) unsigned long long gu64;
        gu64* gran = (gu64*)(p.ln_stats + (size_t)band * p.tiles_n * 512);  // [tiles_n][256 rows][2]
        if (tid < 256) {
            const unsigned long long ep = (unsigned long long)p.ln_epoch << 32;
            const float S_loc = red[tid] + red[256 + tid], M2_loc = red2[tid] + red2[256 + tid];
            __hip_atomic_store(gran + ((size_t)tn * 256 + tid) * 2, ep | __builtin_bit_cast(unsigned, S_loc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gran + ((size_t)tn * 256 + tid) * 2 + 1, ep | __builtin_bit_cast(unsigned, M2_loc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float sj[4], qj[4];
            bool complete = false;
            for (int spins = 0; spins < (1 << 20); ++spins) {
                // the eight granules of the row (tiles_n == 4 in this form) as eight loads in flight, then the tags: a branch or a
                // short-circuit between the loads costs one L2 round trip per pair
                unsigned long long x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    x[j] = __hip_atomic_load(gran + ((size_t)(j >> 1) * 256 + tid) * 2 + (j & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned bad = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) bad |= (unsigned)(x[j] >> 32) ^ p.ln_epoch;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sj[j] = __builtin_bit_cast(float, (unsigned)x[2 * j]);
                    qj[j] = __builtin_bit_cast(float, (unsigned)x[2 * j + 1]);
                }
                if (__all(bad == 0)) { complete = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            // combine: mean, then M2 = sum_j [M2_j + n_j (mean_j - mean)^2]
            const float n_all = (float)(p.tiles_n * BBN);
            const float mean = ((sj[0] + sj[1]) + (sj[2] + sj[3])) / n_all;
            float M2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < p.tiles_n) {
                    const float dj = sj[j] * (1.0f / BBN) - mean;
                    M2 += qj[j] + (float)BBN * dj * dj;
                }
            }
            // a sweep that gave up (a band mate never ran: the launch was not co-resident) poisons its rows, so the failure
            // shows as NaN downstream instead of as plausible numbers
            mr[tid] = complete ? make_float2(mean, 1.0f / sqrtf(M2 / n_all + p.ln_eps)) : make_float2(__builtin_nanf(""), __builtin_nanf(""));
            raise_fault(p.fault, !complete, DIDI_FAULT_LN_EXCHANGE);
        }
        __syncthreads();
        PP_STAMP(5);  // band statistics collected
        // 7. normalise this workgroup's slice and store it
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int n = n0 + wn * 96 + 32 * k + 8 * lg;
            const float4 g0 = *reinterpret_cast<const float4*>(gamma_lds + (n - n0)), g1 = *reinterpret_cast<const float4*>(gamma_lds + (n - n0) + 4);
            const float4 c0 = *reinterpret_cast<const float4*>(beta_lds + (n - n0)), c1 = *reinterpret_cast<const float4*>(beta_lds + (n - n0) + 4);
            const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int row = wm * 64 + mi * 16 + lr;
                const int m = m0 + row;
                if (m >= p.M) continue;
                const float2 s2 = mr[row];
                float y[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    y[i] = (acc[2 * k][mi][i] - s2.x) * s2.y * g[i] + c[i];
                    y[4 + i] = (acc[2 * k + 1][mi][i] - s2.x) * s2.y * g[4 + i] + c[4 + i];
                }
                const size_t o = (size_t)m * p.ldc + n;
                if (p.C_f32) {
                    *reinterpret_cast<float4*>(p.C_f32 + o) = make_float4(y[0], y[1], y[2], y[3]);
                    *reinterpret_cast<float4*>(p.C_f32 + o + 4) = make_float4(y[4], y[5], y[6], y[7]);
                }
                uint2 h0, l0, h1, l1;
                if constexpr (F16) {  // packed truncating conversions (f16x2_dither), the offset that rounds to nearest when no dither is asked for
                    const int d = dith ? dith : DIDI_RNE_AS_DITHER;
                    if (!p.C_lo) {  // hi plane only (the residual stream of a one-product call, DIDI_OPT_HI_RESIDUAL)
                        h0 = pack4_hi_dither(y, d);
                        h1 = pack4_hi_dither(y + 4, d);
                        *reinterpret_cast<uint4*>(p.C_hi + o) = make_uint4(h0.x, h0.y, h1.x, h1.y);
                        continue;
                    }
                    if (p.lo8 & 2) {  // remainder plane as bytes
                        uint32_t b0, b1;
                        split4_dither_lo8(y, h0, b0, d);
                        split4_dither_lo8(y + 4, h1, b1, d);
                        *reinterpret_cast<uint4*>(p.C_hi + o) = make_uint4(h0.x, h0.y, h1.x, h1.y);
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(p.C_lo) + o) = make_uint2(b0, b1);
                        continue;
                    }
                    split4_dither(y, h0, l0, d);
                    split4_dither(y + 4, h1, l1, d);
                } else {
                    split4(y, h0, l0, 0);
                    split4(y + 4, h1, l1, 0);
                }
                *reinterpret_cast<uint4*>(p.C_hi + o) = make_uint4(h0.x, h0.y, h1.x, h1.y);
                if (p.C_lo) *reinterpret_cast<uint4*>(p.C_lo + o) = make_uint4(l0.x, l0.y, l1.x, l1.y);
            }
        }
        PP_STAMP(6);  // every store issued
#ifdef DIDI_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        PP_STAMP(7);  // ... and drained
    }
}

static int fill_operands(GemmParams& p, int precision, const u16* A_hi, const u16* A_lo, const u16* W_hi, const u16* W_lo) {
    p.A_hi = A_hi; p.W_hi = W_hi;
    if (precision == DIDI_PREC_BF16 || precision == DIDI_PREC_FP16) {
        p.npass = 1;
        p.f16 = precision == DIDI_PREC_FP16;
        p.A_lo = A_hi; p.W_lo = W_hi;  // never read
        return DIDI_OK;
    }
    if (precision == DIDI_PREC_BF16X3) {
        if (!A_lo || !W_lo) return DIDI_ERR_BAD_ARG;
        p.npass = 3;
        p.A_lo = A_lo; p.W_lo = W_lo;
        return DIDI_OK;
    }
    if (precision == DIDI_PREC_FP16X3) {  // all three products on fp16 planes (~22 bits): the final rounding / logits of the fp16 modes
        if (!A_lo || !W_lo) return DIDI_ERR_BAD_ARG;
        p.npass = 3;
        p.f16 = 1;
        p.A_lo = A_lo; p.W_lo = W_lo;
        return DIDI_OK;
    }
    if (precision == DIDI_PREC_FP16X2) {  // A_hi * (W_hi + W_lo) on fp16 planes: the weights enter exactly, the activations rounded
        if (!W_lo) return DIDI_ERR_BAD_ARG;
        p.npass = 2;
        p.f16 = 1;
        p.A_lo = A_hi; p.W_lo = W_lo;  // A_lo never read
        return DIDI_OK;
    }
    return DIDI_ERR_BAD_ARG;
}

// 0 = register-staged fallback, else the LDS-DMA variant for the precision
static int pick_mode(const GemmParams& p) {
    if (p.npass != 1) return p.K % 32 == 0 ? p.npass : 0;
    return p.K % 64 == 0 ? 1 : 0;
}

// experiment knob (tools/gemm_bench.py): DIDI_OPT_GEMM_BIG = 0 forces the 128x128 tile everywhere
static bool gemm_big_enabled() { return didi_opt(DIDI_OPT_GEMM_BIG) != 0; }
// experiment knob: DIDI_OPT_GEMM_MT = 2 selects the 128-row, 4-wave form of the big tile (default 4: measured faster)
static int gemm_big_mt() { return didi_opt(DIDI_OPT_GEMM_MT) == 2 ? 2 : 4; }
template <int NPASS, int MT, bool F16>
static bool opt_in_lds() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_big_kernel<NPASS, MT, F16>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BigGeom<MT>::STAGE) == hipSuccess;
}
// experiment knob: DIDI_OPT_GEMM_PP = 0 selects the in-phase K-loop (gemm_big_kernel) for A/B timing
static bool gemm_pp_enabled() { return didi_opt(DIDI_OPT_GEMM_PP) != 0; }
// Per-device set-up, done on the first call on each device: the > 64 KiB dynamic-LDS opt-ins of every big-tile kernel
// (function attributes live per device) and the device's CU count.  Keyed by the CURRENT device, so a process that
// drives several GPUs gets each one's own values.
template <int NPASS, bool F16, bool LN, bool AMAX, int ATTN = 0, int TN = 0>
static bool opt_in_pp() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<NPASS, F16, LN, AMAX, ATTN, TN>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, ATTN == 2 ? XA_LDS_BYTES : PP_LDS_BYTES) == hipSuccess;
}
struct DeviceInfo { int state; int ncu; };  // state: 0 = not probed, 1 = ready, -1 = failed
static const DeviceInfo* device_info() {
    constexpr int MAXDEV = 64;
    static DeviceInfo info[MAXDEV];
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    DeviceInfo& d = info[dev];
    if (d.state == 0) {
        bool ok = opt_in_lds<3, 4, false>() && opt_in_lds<1, 4, false>() && opt_in_lds<1, 4, true>() &&
                  opt_in_lds<3, 2, false>() && opt_in_lds<1, 2, false>() && opt_in_lds<1, 2, true>();
        ok = ok && opt_in_pp<3, false, false, false>() && opt_in_pp<1, false, false, false>() && opt_in_pp<1, true, false, false>();
        ok = ok && opt_in_pp<3, false, true, false>() && opt_in_pp<1, false, true, false>() && opt_in_pp<1, true, true, false>();
        ok = ok && opt_in_pp<3, false, false, true>() && opt_in_pp<1, false, false, true>() && opt_in_pp<1, true, false, true>();
        ok = ok && opt_in_pp<2, true, false, false>() && opt_in_pp<2, true, true, false>() && opt_in_pp<2, true, false, true>();
        ok = ok && opt_in_pp<3, true, false, false>() && opt_in_pp<3, true, false, true>() && opt_in_pp<3, true, true, false>();
        ok = ok && opt_in_pp<1, true, false, false, 1>() && opt_in_pp<2, true, false, false, 1>();
        ok = ok && opt_in_pp<1, true, false, false, 2>() && opt_in_pp<2, true, false, false, 2>();
        ok = ok && opt_in_pp<3, false, false, false, 0, 3>() && opt_in_pp<3, true, false, false, 0, 3>() && opt_in_pp<2, true, false, false, 0, 3>();
        ok = ok && opt_in_pp<3, false, false, false, 0, 2>() && opt_in_pp<3, true, false, false, 0, 2>() && opt_in_pp<2, true, false, false, 0, 2>();
        int c = 0;
        ok = ok && hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && c > 0;
        d.ncu = c;
        d.state = ok ? 1 : -1;
    }
    return d.state == 1 ? &d : nullptr;
}
static bool allow_big_lds() { return device_info() != nullptr; }
template <int NPASS, int MT, bool F16>
static void launch_big(GemmParams& p, hipStream_t st) {
    using G = BigGeom<MT>;
    p.tiles_m = cdiv(p.M, G::BM_);
    gemm_big_kernel<NPASS, MT, F16><<<p.tiles_m * p.tiles_n, G::THREADS, 2 * G::STAGE, st>>>(p);
}

static inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
static inline bool al8(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 7) == 0; }

extern "C" int didi_gemm(const didi_gemm_args* a, void* stream) {
    DIDI_CHECK_ARG(a && a->A_hi && a->W_hi && a->M >= 0 && a->N > 0 && a->K > 0);
    DIDI_CHECK_ARG(a->C_f32 || a->C_hi);
    DIDI_CHECK_ARG(!(a->C_lo && !a->C_hi));
    if (a->K % 8 != 0 || a->N % 4 != 0) return DIDI_ERR_UNSUPPORTED;
    DIDI_CHECK_ARG(a->lda % 8 == 0 && a->ldw % 8 == 0 && a->ldc % 4 == 0 && a->lda >= a->K && a->ldw >= a->K && a->ldc >= a->N);
    DIDI_CHECK_ARG(al16(a->A_hi) && al16(a->A_lo) && al16(a->W_hi) && al16(a->W_lo) && al16(a->bias) && al16(a->R_f32));
    DIDI_CHECK_ARG(al8(a->R_hi) && al8(a->R_lo) && al16(a->C_f32) && al8(a->C_hi) && al8(a->C_lo));
    if (a->R_f32 || a->R_hi) DIDI_CHECK_ARG(a->ldr % 4 == 0 && a->ldr >= a->N);
    DIDI_CHECK_ARG(a->act >= DIDI_ACT_NONE && a->act <= DIDI_ACT_GELU_NEW);
    if (a->M == 0) return DIDI_OK;
    GemmParams p{};
    p.M = a->M; p.N = a->N; p.K = a->K;
    int rc = fill_operands(p, a->precision, a->A_hi, a->A_lo, a->W_hi, a->W_lo);
    if (rc != DIDI_OK) return rc;
    p.lda = a->lda; p.ldw = a->ldw;
    p.bias = a->bias; p.act = a->act;
    p.R_f32 = a->R_f32; p.R_hi = a->R_hi; p.R_lo = a->R_lo; p.ldr = a->ldr;
    p.C_f32 = a->C_f32; p.C_hi = a->C_hi; p.C_lo = a->C_lo; p.ldc = a->ldc;
    p.fault = a->fault;
    DIDI_CHECK_ARG(a->act_dither >= 0 && a->act_dither < 8192);
    p.dither = a->act_dither;
#ifdef DIDI_STAMPS
    p.stamps = g_stamp_buf;
#endif
    hipStream_t st = as_stream(stream);
    const int mode = pick_mode(p);
    const int big_tiles_n = cdiv(a->N, BBN);
    const int big_min_m = didi_opt(DIDI_OPT_BIG_MIN_M);  // tuning knob
    const bool mt4 = gemm_big_mt() == 4;
    const bool wide_ok = a->ldc % 8 == 0 && al16(a->C_hi) && al16(a->C_lo) && al16(a->R_hi) && al16(a->R_lo) &&
                         (!(a->R_f32 || a->R_hi) || a->ldr % 8 == 0);
    const bool big = mode != 0 && gemm_big_enabled() && a->M >= big_min_m && a->N >= BBN &&
                     (long)big_tiles_n * BBN * 100 <= (long)a->N * 115 &&  // <= 15 % padded columns
                     (!(mode == 2 || (mode == 3 && p.f16)) || (mt4 && wide_ok && gemm_pp_enabled()));  // fp16x2 / fp16x3: staggered kernel only
    if (big) {
        if (!allow_big_lds()) return DIDI_ERR_LAUNCH;
        p.tiles_n = big_tiles_n;
        if (mt4 && wide_ok && gemm_pp_enabled()) {
            p.tiles_m = cdiv(p.M, 256);
            const int grid = p.tiles_m * p.tiles_n;
            const size_t lds = PP_LDS_BYTES;  // + GELU table
            if (mode == 3 && p.f16) gemm_pp_kernel<3, true, false><<<grid, 512, lds, st>>>(p);
            else if (mode == 3) gemm_pp_kernel<3, false, false><<<grid, 512, lds, st>>>(p);
            else if (mode == 2) gemm_pp_kernel<2, true, false><<<grid, 512, lds, st>>>(p);
            else if (p.f16) gemm_pp_kernel<1, true, false><<<grid, 512, lds, st>>>(p);
            else gemm_pp_kernel<1, false, false><<<grid, 512, lds, st>>>(p);
            DIDI_LAUNCH_CHECK();
            return DIDI_OK;
        }
        if (mode == 2 || (mode == 3 && p.f16)) return DIDI_ERR_UNSUPPORTED;  // excluded by `big` above
        if (mode == 3) { if (mt4) launch_big<3, 4, false>(p, st); else launch_big<3, 2, false>(p, st); }
        else if (p.f16) { if (mt4) launch_big<1, 4, true>(p, st); else launch_big<1, 2, true>(p, st); }
        else { if (mt4) launch_big<1, 4, false>(p, st); else launch_big<1, 2, false>(p, st); }
    } else {
        p.tiles_m = cdiv(a->M, BM); p.tiles_n = cdiv(a->N, BN);
        const int grid = p.tiles_m * p.tiles_n;
        if (mode == 3 && p.f16) gemm_kernel<3, 2, true><<<grid, 256, SMEM_BYTES, st>>>(p);
        else if (mode == 3) gemm_kernel<3, 2, false><<<grid, 256, SMEM_BYTES, st>>>(p);
        else if (mode == 2) gemm_kernel<2, 2, true><<<grid, 256, SMEM_BYTES, st>>>(p);
        else if (mode == 1 && p.f16) gemm_kernel<1, 2, true><<<grid, 256, SMEM_BYTES, st>>>(p);
        else if (mode == 1) gemm_kernel<1, 2, false><<<grid, 256, SMEM_BYTES, st>>>(p);
        else if (p.f16) gemm_kernel<0, 2, true><<<grid, 256, SMEM_BYTES, st>>>(p);
        else gemm_kernel<0, 2, false><<<grid, 256, SMEM_BYTES, st>>>(p);
    }
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ query / key / value projections + causal self-attention
extern "C" int didi_qkv_self_attention_supported(int M, int H, int heads, int S, int K, int precision) {
    return (precision == DIDI_PREC_FP16 || precision == DIDI_PREC_FP16X2) && S == 64 && heads > 0 && H == heads * 64 && M >= 2048 &&
           M % 64 == 0 && K % 64 == 0 && K >= 128 && 3 * H == heads * BBN;
}

extern "C" int didi_qkv_self_attention(const didi_gemm_args* a, int heads, int S, float scale, void* stream) {
    DIDI_CHECK_ARG(a && a->A_hi && a->W_hi && a->bias && a->C_hi && a->M >= 0 && a->N > 0 && a->N % 3 == 0 && a->K > 0 && scale >= 0.f);
    DIDI_CHECK_ARG(a->act == DIDI_ACT_NONE && !a->C_f32 && !a->C_lo && !a->R_f32 && !a->R_hi && !a->R_lo);
    const int H = a->N / 3;
    if (!didi_qkv_self_attention_supported(a->M, H, heads, S, a->K, a->precision)) return DIDI_ERR_UNSUPPORTED;
    DIDI_CHECK_ARG(a->lda % 8 == 0 && a->ldw % 8 == 0 && a->lda >= a->K && a->ldw >= a->K && a->ldc % 4 == 0 && a->ldc >= H);
    DIDI_CHECK_ARG(al16(a->A_hi) && al16(a->W_hi) && al16(a->W_lo) && al16(a->bias) && al8(a->C_hi));
    DIDI_CHECK_ARG(a->act_dither >= 0 && a->act_dither < 8192);
    if (a->M == 0) return DIDI_OK;
    if (!allow_big_lds()) return DIDI_ERR_LAUNCH;
    GemmParams p{};
    p.M = a->M; p.N = a->N; p.K = a->K;
    int rc = fill_operands(p, a->precision, a->A_hi, a->A_lo, a->W_hi, a->W_lo);
    if (rc != DIDI_OK) return rc;
    p.lda = a->lda; p.ldw = a->ldw; p.bias = a->bias; p.act = DIDI_ACT_NONE;
    p.C_hi = a->C_hi; p.ldc = a->ldc; p.fault = a->fault; p.dither = a->act_dither;
    p.attn_scale = 1.4426950408889634f * (scale > 0.f ? scale : 0.125f);  // softmax scale * log2(e); 0 = 1 / sqrt(64)
    p.attn_mask0 = 0.0f;
#ifdef DIDI_STAMPS
    p.stamps = g_stamp_buf;
#endif
    p.tiles_m = cdiv(p.M, 256); p.tiles_n = heads;
    const int grid = p.tiles_m * p.tiles_n;
    hipStream_t st = as_stream(stream);
    if (p.npass == 2) gemm_pp_kernel<2, true, false, false, 1><<<grid, 512, PP_LDS_BYTES, st>>>(p);
    else gemm_pp_kernel<1, true, false, false, 1><<<grid, 512, PP_LDS_BYTES, st>>>(p);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ cross-attention query projection + cross-attention
extern "C" int didi_q_cross_attention_supported(int M, int H, int heads, int S, int C, int K, int precision) {
    return (precision == DIDI_PREC_FP16 || precision == DIDI_PREC_FP16X2) && S == 64 && C == 64 && heads > 0 && heads % 3 == 0 &&
           H == heads * 64 && M >= 2048 && M % 64 == 0 && K % 64 == 0 && K >= 128;
}

extern "C" int didi_q_cross_attention(const didi_gemm_args* a, const uint16_t* K_hi, const uint16_t* V_hi, int ldkv, const int32_t* key_mask,
                                      const int32_t* key_count, int heads, int S, int C, float scale, void* stream) {
    DIDI_CHECK_ARG(a && a->A_hi && a->W_hi && a->bias && a->C_hi && K_hi && V_hi && a->M >= 0 && a->N > 0 && a->K > 0 && scale >= 0.f);
    DIDI_CHECK_ARG(a->act == DIDI_ACT_NONE && !a->C_f32 && !a->C_lo && !a->R_f32 && !a->R_hi && !a->R_lo);
    if (!didi_q_cross_attention_supported(a->M, a->N, heads, S, C, a->K, a->precision)) return DIDI_ERR_UNSUPPORTED;
    DIDI_CHECK_ARG(a->lda % 8 == 0 && a->ldw % 8 == 0 && a->lda >= a->K && a->ldw >= a->K && a->ldc % 4 == 0 && a->ldc >= a->N);
    DIDI_CHECK_ARG(ldkv % 8 == 0 && ldkv >= a->N && al16(K_hi) && al16(V_hi));
    DIDI_CHECK_ARG(al16(a->A_hi) && al16(a->W_hi) && al16(a->W_lo) && al16(a->bias) && al8(a->C_hi));
    DIDI_CHECK_ARG(a->act_dither >= 0 && a->act_dither < 8192);
    if (a->M == 0) return DIDI_OK;
    if (!allow_big_lds()) return DIDI_ERR_LAUNCH;
    GemmParams p{};
    p.M = a->M; p.N = a->N; p.K = a->K;
    int rc = fill_operands(p, a->precision, a->A_hi, a->A_lo, a->W_hi, a->W_lo);
    if (rc != DIDI_OK) return rc;
    p.lda = a->lda; p.ldw = a->ldw; p.bias = a->bias; p.act = DIDI_ACT_NONE;
    p.C_hi = a->C_hi; p.ldc = a->ldc; p.fault = a->fault; p.dither = a->act_dither;
    p.attn_K = K_hi; p.attn_V = V_hi; p.attn_ldkv = ldkv; p.attn_B = a->M / 64;
    p.attn_mask = key_mask; p.attn_count = key_mask ? key_count : nullptr;
    p.attn_scale = 1.4426950408889634f * (scale > 0.f ? scale : 0.125f);
    p.attn_mask0 = 0.0f;
#ifdef DIDI_STAMPS
    p.stamps = g_stamp_buf;
#endif
    p.tiles_m = cdiv(p.M, 256); p.tiles_n = a->N / BBN;
    const int grid = p.tiles_m * p.tiles_n;
    hipStream_t st = as_stream(stream);
    if (p.npass == 2) gemm_pp_kernel<2, true, false, false, 2><<<grid, 512, XA_LDS_BYTES, st>>>(p);
    else gemm_pp_kernel<1, true, false, false, 2><<<grid, 512, XA_LDS_BYTES, st>>>(p);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ small batches: split-K GEMM + row epilogue
// Below ~2048 rows the 128x128 grid holds a few dozen workgroups, each walking all of K alone: 24 us per launch at
// M = 63 whatever the shape, with 90 % of the chip idle.  Here K is cut into chunks so that >= ~256 workgroups run, the
// raw partial sums go to a workspace (L2-resident: <= 64 KiB per workgroup), and ONE row kernel adds them up and applies
// bias, activation, residual and, for the output projections, LayerNorm -- so the separate LayerNorm pass goes away too.
// weight-gradient shapes with enough rows for the 256x192 tile: its staggered K-loop multiplies ~1.6x faster than the 128x128
// tile's, which pays for the extra splits (fewer, larger tiles -> more K-splits to fill the CUs -> more partial sums)
static bool small_uses_big_tile(int M, int N, int K) {
    return K >= 8192 && M >= 512 && N >= BBN && N % 8 == 0 && (long)cdiv(N, BBN) * BBN * 100 <= (long)N * 115 && cdiv(M, 256) * cdiv(N, BBN) < 128 &&
           didi_opt(DIDI_OPT_GEMM_BIG) != 0 && didi_opt(DIDI_OPT_GEMM_PP) != 0;
}
static int small_splits(int M, int N, int K, int precision) {
    const int max_m = didi_opt(DIDI_OPT_SMALL_MAX_M);  // tuning knob
    if (M <= 0 || N % 4 != 0 || N > 3072) return 0;
    // weight-gradient shapes (dW = dC^T A: a few thousand rows, K = the batch's B S rows) have the same problem at any M: a
    // few dozen output tiles, each walking a very long K.  The sampler's shapes (K <= 5120) never take this branch.
    if (M >= max_m && !(K >= 8192 && cdiv(M, BM) * cdiv(N, BN) < 256)) return 0;
    const int unit = 128;  // K columns per split are a multiple of this (two one-product K-tiles, four bf16x3 ones)
    if (K % unit != 0 || K < 2 * unit) return 0;
    if (precision != DIDI_PREC_BF16 && precision != DIDI_PREC_FP16 && precision != DIDI_PREC_BF16X3 && precision != DIDI_PREC_FP16X2 && precision != DIDI_PREC_FP16X3) return 0;
    // the 128x128 kernel runs two or more workgroups per CU (any count near 256 tiles x splits is one round); the 256x192 kernel
    // runs ONE, so its tiles x splits must not exceed the CUs: 264 workgroups are two rounds, the second one 3 % full
    const bool big = small_uses_big_tile(M, N, K);
    const int wgs = big ? cdiv(M, 256) * cdiv(N, BBN) : cdiv(M, BM) * cdiv(N, BN);
    int ks = std::min(big ? std::max(256 / wgs, 1) : cdiv(256, wgs), K / unit);
    if (ks < 2) return 0;
    const int kc = cdiv(K / unit, ks) * unit;
    return cdiv(K, kc);
}

extern "C" size_t didi_gemm_small_workspace_bytes(int M, int N, int K, int precision) {
    const int ks = small_splits(M, N, K, precision);
    return ks ? (size_t)ks * M * N * sizeof(float) : 0;
}

// One thread per 4 output columns: sum the K-split partials (independent loads, all in flight), bias, activation, residual.
// LN = false: a flat elementwise grid.  LN = true: one 256-thread workgroup per row (CH chunks per thread, N <= 1024 * CH),
// two-pass mean / variance through a wave + LDS reduction, as layernorm_kernel computes them.
struct RowsParams {
    const float* part; int ks, M, N;
    const float* bias; int act;
    const u16 *r_hi, *r_lo; const float* r_f32; int ldr;
    const float *gamma, *beta; float eps;
    u16 *y_hi, *y_lo; float* y_f32; int f16;
    uint32_t* fault;
};

__device__ __forceinline__ float4 rows_value(const RowsParams& p, int m, int c) {
    const size_t plane = (size_t)p.M * p.N;
    const float* src = p.part + (size_t)m * p.N + (size_t)c * 4;
    float4 a = *reinterpret_cast<const float4*>(src);
    float4 a2 = make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 1;
    for (; z + 1 < p.ks; z += 2) {  // two independent chains keep the loads in flight
        const float4 b = *reinterpret_cast<const float4*>(src + z * plane), d = *reinterpret_cast<const float4*>(src + (z + 1) * plane);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        a2.x += d.x; a2.y += d.y; a2.z += d.z; a2.w += d.w;
    }
    if (z < p.ks) {
        const float4 b = *reinterpret_cast<const float4*>(src + z * plane);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
    if (p.bias) {
        const float4 b = reinterpret_cast<const float4*>(p.bias)[c];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (p.act != DIDI_ACT_NONE) { a.x = act_apply(a.x, p.act); a.y = act_apply(a.y, p.act); a.z = act_apply(a.z, p.act); a.w = act_apply(a.w, p.act); }
    if (p.r_f32) {
        const float4 r = *reinterpret_cast<const float4*>(p.r_f32 + (size_t)m * p.ldr + c * 4);
        a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
    } else if (p.r_hi) {
        float r[4];
        unpack4(*reinterpret_cast<const uint2*>(p.r_hi + (size_t)m * p.ldr + c * 4), r, p.f16);
        a.x += r[0]; a.y += r[1]; a.z += r[2]; a.w += r[3];
        if (p.r_lo) {
            unpack4(*reinterpret_cast<const uint2*>(p.r_lo + (size_t)m * p.ldr + c * 4), r, p.f16);
            a.x += r[0]; a.y += r[1]; a.z += r[2]; a.w += r[3];
        }
    }
    return a;
}

__device__ __forceinline__ void rows_store(const RowsParams& p, int m, int c, const float (&y)[4]) {
    const size_t o = (size_t)m * p.N + (size_t)c * 4;
    if (p.y_f32) *reinterpret_cast<float4*>(p.y_f32 + o) = make_float4(y[0], y[1], y[2], y[3]);
    if (p.y_hi) {
        if (p.f16) raise_fault(p.fault, amax2(amax2(0.f, y[0], y[1]), y[2], y[3]) > DIDI_F16_MAX, DIDI_FAULT_FP16_RANGE);
        uint2 h, l;
        split4(y, h, l, p.f16);
        *reinterpret_cast<uint2*>(p.y_hi + o) = h;
        if (p.y_lo) *reinterpret_cast<uint2*>(p.y_lo + o) = l;
    }
}

__global__ void __launch_bounds__(256) splitk_sum_kernel(RowsParams p) {
    if (p.f16) f16_saturate_mode();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int nc = p.N / 4;
    if (i >= (size_t)p.M * nc) return;
    const int m = (int)(i / nc), c = (int)(i % nc);
    const float4 a = rows_value(p, m, c);
    const float y[4] = {a.x, a.y, a.z, a.w};
    rows_store(p, m, c, y);
}

__device__ __forceinline__ float block_sum256(float v, float* red) {  // red: 4 floats of LDS; all 256 threads call it
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();  // the previous use of red is over
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

template <int CH>
__global__ void __launch_bounds__(256) splitk_ln_kernel(RowsParams p) {
    __shared__ float red[4];
    if (p.f16) f16_saturate_mode();
    const int m = blockIdx.x, nc = p.N / 4;
    float v[CH][4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c = threadIdx.x + 256 * j;
        v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f;
        if (c < nc) {
            const float4 a = rows_value(p, m, c);
            v[j][0] = a.x; v[j][1] = a.y; v[j][2] = a.z; v[j][3] = a.w;
            s += (a.x + a.y) + (a.z + a.w);
        }
    }
    const float mean = block_sum256(s, red) / (float)p.N;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j)
        if (threadIdx.x + 256 * j < nc) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mean; q += d * d; }
        }
    const float rstd = 1.0f / sqrtf(block_sum256(q, red) / (float)p.N + p.eps);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < nc) {
            const float4 g = reinterpret_cast<const float4*>(p.gamma)[c], b = reinterpret_cast<const float4*>(p.beta)[c];
            const float y[4] = {(v[j][0] - mean) * rstd * g.x + b.x, (v[j][1] - mean) * rstd * g.y + b.y,
                                (v[j][2] - mean) * rstd * g.z + b.z, (v[j][3] - mean) * rstd * g.w + b.w};
            rows_store(p, m, c, y);
        }
    }
}

extern "C" int didi_gemm_small(const didi_gemm_args* a, const float* gamma, const float* beta, float eps, void* workspace,
                               void* stream) {
    DIDI_CHECK_ARG(a && a->A_hi && a->W_hi && workspace && (a->C_f32 || a->C_hi) && !(a->C_lo && !a->C_hi));
    DIDI_CHECK_ARG((gamma == nullptr) == (beta == nullptr));
    const int ks = small_splits(a->M, a->N, a->K, a->precision);
    if (!ks || a->ldc != a->N) return DIDI_ERR_UNSUPPORTED;
    DIDI_CHECK_ARG(a->lda % 8 == 0 && a->ldw % 8 == 0 && a->lda >= a->K && a->ldw >= a->K);
    DIDI_CHECK_ARG(al16(a->A_hi) && al16(a->A_lo) && al16(a->W_hi) && al16(a->W_lo) && al16(a->bias) && al16(a->R_f32) && al16(workspace));
    DIDI_CHECK_ARG(al8(a->R_hi) && al8(a->R_lo) && al16(a->C_f32) && al8(a->C_hi) && al8(a->C_lo) && al16(gamma) && al16(beta));
    if (a->R_f32 || a->R_hi) DIDI_CHECK_ARG(a->ldr % 4 == 0 && a->ldr >= a->N);
    DIDI_CHECK_ARG(a->act >= DIDI_ACT_NONE && a->act <= DIDI_ACT_GELU_NEW);
    GemmParams p{};
    p.M = a->M; p.N = a->N; p.K = a->K;
    int rc = fill_operands(p, a->precision, a->A_hi, a->A_lo, a->W_hi, a->W_lo);
    if (rc != DIDI_OK) return rc;
    p.lda = a->lda; p.ldw = a->ldw; p.act = DIDI_ACT_NONE;
    p.C_f32 = reinterpret_cast<float*>(workspace); p.ldc = a->N;
    p.k_split = cdiv(a->K / 128, ks) * 128;
    p.tiles_m = cdiv(a->M, BM); p.tiles_n = cdiv(a->N, BN);
    hipStream_t st = as_stream(stream);
    if (small_uses_big_tile(a->M, a->N, a->K) && allow_big_lds()) {  // raw partial sums out of the staggered 256x192 kernel
        p.tiles_m = cdiv(a->M, 256); p.tiles_n = cdiv(a->N, BBN);
        const dim3 grid_pp(p.tiles_m * p.tiles_n, ks);
        if (p.npass == 3 && p.f16) gemm_pp_kernel<3, true, false><<<grid_pp, 512, PP_LDS_BYTES, st>>>(p);
        else if (p.npass == 3) gemm_pp_kernel<3, false, false><<<grid_pp, 512, PP_LDS_BYTES, st>>>(p);
        else if (p.npass == 2) gemm_pp_kernel<2, true, false><<<grid_pp, 512, PP_LDS_BYTES, st>>>(p);
        else if (p.f16) gemm_pp_kernel<1, true, false><<<grid_pp, 512, PP_LDS_BYTES, st>>>(p);
        else gemm_pp_kernel<1, false, false><<<grid_pp, 512, PP_LDS_BYTES, st>>>(p);
    } else {
    const dim3 grid(p.tiles_m * p.tiles_n, ks);
    if (p.npass == 3 && p.f16) gemm_kernel<3, 2, true><<<grid, 256, SMEM_BYTES, st>>>(p);
    else if (p.npass == 3) gemm_kernel<3, 2, false><<<grid, 256, SMEM_BYTES, st>>>(p);
    else if (p.npass == 2) gemm_kernel<2, 2, true><<<grid, 256, SMEM_BYTES, st>>>(p);
    else if (p.f16) gemm_kernel<1, 2, true><<<grid, 256, SMEM_BYTES, st>>>(p);
    else gemm_kernel<1, 2, false><<<grid, 256, SMEM_BYTES, st>>>(p);
    }
    DIDI_LAUNCH_CHECK();
    RowsParams r{};
    r.part = reinterpret_cast<const float*>(workspace); r.ks = ks; r.M = a->M; r.N = a->N;
    r.bias = a->bias; r.act = a->act;
    r.r_hi = a->R_hi; r.r_lo = a->R_lo; r.r_f32 = a->R_f32; r.ldr = a->ldr;
    r.gamma = gamma; r.beta = beta; r.eps = eps;
    r.y_hi = a->C_hi; r.y_lo = a->C_lo; r.y_f32 = a->C_f32; r.f16 = p.f16;
    r.fault = a->fault;
    if (!gamma) splitk_sum_kernel<<<cdiv((long)a->M * (a->N / 4), 256), 256, 0, st>>>(r);
    else if (a->N <= 1024) splitk_ln_kernel<1><<<a->M, 256, 0, st>>>(r);
    else if (a->N <= 2048) splitk_ln_kernel<2><<<a->M, 256, 0, st>>>(r);
    else splitk_ln_kernel<3><<<a->M, 256, 0, st>>>(r);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ weight gradient from un-transposed operands
// dW [N, K] = dC^T A for dC [M, N], A [M, K] given as split planes in their natural row-major layout (gemm_pp_kernel's TN form):
// split-K over the M rows (a few dozen 256x192 output tiles, each walking a share of the batch's rows), partial sums to the
// workspace, one pass that adds them up.  Supported where the 256x192 split-K form is (small_uses_big_tile) in the two- and
// three-product precisions; 0 bytes / DIDI_ERR_UNSUPPORTED elsewhere (the caller then transposes and uses didi_gemm_small).
// rows per split (a multiple of the 32-row K-tile; 0 = not supported): as many splits as keep tiles x splits within the CUs (one
// workgroup per CU: everything runs in one round), rows spread evenly over them
static int wgrad_rows_per_split(int M, int N, int K, int precision) {
    if (precision != DIDI_PREC_BF16X3 && precision != DIDI_PREC_FP16X2 && precision != DIDI_PREC_FP16X3) return 0;
    if (N % 8 != 0 || K % 8 != 0 || M % 32 != 0 || M < 8192 || N < 512 || K < BBN) return 0;
    if ((long)cdiv(K, BBN) * BBN * 100 > (long)K * 115 || didi_opt(DIDI_OPT_GEMM_BIG) == 0 || didi_opt(DIDI_OPT_GEMM_PP) == 0) return 0;
    const DeviceInfo* d = device_info();
    const int tiles = cdiv(N, 256) * cdiv(K, BBN);
    if (!d || tiles >= 128 || tiles > d->ncu / 2) return 0;
    const int ks = d->ncu / tiles;
    return cdiv(M / 32, ks) * 32;
}
extern "C" size_t didi_gemm_wgrad_workspace_bytes(int M, int N, int K, int precision) {
    const int kc = wgrad_rows_per_split(M, N, K, precision);
    return kc ? (size_t)cdiv(M, kc) * N * K * sizeof(float) : 0;
}
extern "C" int didi_gemm_wgrad(const uint16_t* dC_hi, const uint16_t* dC_lo, int ldd, const uint16_t* A_hi, const uint16_t* A_lo, int lda, int M, int N,
                               int K, int precision, float* dW, void* workspace, void* stream) {
    DIDI_CHECK_ARG(dC_hi && A_hi && dW && workspace && M > 0 && N > 0 && K > 0);
    const int kc = wgrad_rows_per_split(M, N, K, precision);
    if (!kc) return DIDI_ERR_UNSUPPORTED;
    const int ks = cdiv(M, kc);
    DIDI_CHECK_ARG(ldd % 8 == 0 && lda % 8 == 0 && ldd >= N && lda >= K);
    DIDI_CHECK_ARG(al16(dC_hi) && al16(dC_lo) && al16(A_hi) && al16(A_lo) && al16(dW) && al16(workspace));
    GemmParams p{};
    p.M = N; p.N = K; p.K = M;  // the output's rows come from dC's columns, its columns from A's, the contraction runs over the M rows
    int rc = fill_operands(p, precision, dC_hi, dC_lo, A_hi, A_lo);
    if (rc != DIDI_OK) return rc;
    p.lda = ldd; p.ldw = lda; p.act = DIDI_ACT_NONE;
    p.C_f32 = reinterpret_cast<float*>(workspace); p.ldc = K;
    p.k_split = kc;
    p.tiles_m = cdiv(N, 256); p.tiles_n = cdiv(K, BBN);
    hipStream_t st = as_stream(stream);
    const dim3 grid(p.tiles_m * p.tiles_n, ks);
    if (p.npass == 3 && p.f16) gemm_pp_kernel<3, true, false, false, 0, 3><<<grid, 512, PP_LDS_BYTES, st>>>(p);
    else if (p.npass == 3) gemm_pp_kernel<3, false, false, false, 0, 3><<<grid, 512, PP_LDS_BYTES, st>>>(p);
    else gemm_pp_kernel<2, true, false, false, 0, 3><<<grid, 512, PP_LDS_BYTES, st>>>(p);
    DIDI_LAUNCH_CHECK();
    RowsParams r{};
    r.part = reinterpret_cast<const float*>(workspace); r.ks = ks; r.M = N; r.N = K;
    r.act = DIDI_ACT_NONE; r.y_f32 = dW; r.f16 = p.f16;
    splitk_sum_kernel<<<cdiv((long)N * (K / 4), 256), 256, 0, st>>>(r);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// dA [M, K] = dC [M, N] W [N, K] (+ residual): nn.Linear's input gradient with the weight planes as the forward holds them
// (gemm_pp_kernel's TN = 2 form: the weight side is read row-indexed, no transposed copy of W).
extern "C" int didi_gemm_dgrad_supported(int M, int N, int K, int precision) {
    if (precision != DIDI_PREC_BF16X3 && precision != DIDI_PREC_FP16X2 && precision != DIDI_PREC_FP16X3) return 0;
    return M >= 2048 && N >= 32 && N % 32 == 0 && K >= BBN && K % 8 == 0 && (long)cdiv(K, BBN) * BBN * 100 <= (long)K * 115 &&
           didi_opt(DIDI_OPT_GEMM_BIG) != 0 && didi_opt(DIDI_OPT_GEMM_PP) != 0 && allow_big_lds();
}
extern "C" int didi_gemm_dgrad(const uint16_t* dC_hi, const uint16_t* dC_lo, int ldd, const uint16_t* W_hi, const uint16_t* W_lo, int ldw, int M, int N,
                               int K, int precision, const float* R_f32, int ldr, float* dA, int lda, void* stream) {
    DIDI_CHECK_ARG(dC_hi && W_hi && dA && M > 0 && N > 0 && K > 0);
    if (!didi_gemm_dgrad_supported(M, N, K, precision)) return DIDI_ERR_UNSUPPORTED;
    DIDI_CHECK_ARG(ldd % 8 == 0 && ldw % 8 == 0 && lda % 4 == 0 && ldd >= N && ldw >= K && lda >= K && (!R_f32 || (ldr % 4 == 0 && ldr >= K)));
    DIDI_CHECK_ARG(al16(dC_hi) && al16(dC_lo) && al16(W_hi) && al16(W_lo) && al16(dA) && al16(R_f32));
    GemmParams p{};
    p.M = M; p.N = K; p.K = N;  // the output's columns are W's columns, the contraction runs over W's rows
    int rc = fill_operands(p, precision, dC_hi, dC_lo, W_hi, W_lo);
    if (rc != DIDI_OK) return rc;
    p.lda = ldd; p.ldw = ldw; p.act = DIDI_ACT_NONE;
    p.R_f32 = R_f32; p.ldr = ldr;
    p.C_f32 = dA; p.ldc = lda;
    p.tiles_m = cdiv(M, 256); p.tiles_n = cdiv(K, BBN);
    hipStream_t st = as_stream(stream);
    const int grid = p.tiles_m * p.tiles_n;
    if (p.npass == 3 && p.f16) gemm_pp_kernel<3, true, false, false, 0, 2><<<grid, 512, PP_LDS_BYTES, st>>>(p);
    else if (p.npass == 3) gemm_pp_kernel<3, false, false, false, 0, 2><<<grid, 512, PP_LDS_BYTES, st>>>(p);
    else gemm_pp_kernel<2, true, false, false, 0, 2><<<grid, 512, PP_LDS_BYTES, st>>>(p);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ output projection + residual + LayerNorm
extern "C" int didi_gemm_ln_supported(int M, int N, int K, int precision) {
    const bool prec_ok = precision == DIDI_PREC_BF16 || precision == DIDI_PREC_FP16 || precision == DIDI_PREC_BF16X3 ||
                         precision == DIDI_PREC_FP16X2 || precision == DIDI_PREC_FP16X3;
    return prec_ok && M >= 2048 && N == 4 * BBN && K % 64 == 0 && K >= 128;
}

extern "C" size_t didi_gemm_ln_stats_bytes(int M) { return (size_t)cdiv(M, 256) * 4 * 256 * 2 * sizeof(unsigned long long); }

static int gemm_ln_impl(const didi_gemm_args* a, const float* gamma, const float* beta, float eps, uint16_t* y_hi,
                        uint16_t* y_lo, float* y_f32, void* stats, uint32_t epoch, int lo_bytes, void* stream) {
    DIDI_CHECK_ARG(a && a->A_hi && a->W_hi && a->bias && gamma && beta && y_hi && stats && a->M >= 0);
    DIDI_CHECK_ARG(a->act == DIDI_ACT_NONE && !a->C_f32 && !a->C_hi && !a->C_lo && a->R_hi && !a->R_f32);
    if (!didi_gemm_ln_supported(a->M, a->N, a->K, a->precision)) return DIDI_ERR_UNSUPPORTED;
    DIDI_CHECK_ARG(a->lda % 8 == 0 && a->ldw % 8 == 0 && a->lda >= a->K && a->ldw >= a->K && a->ldr % 8 == 0 && a->ldr >= a->N);
    DIDI_CHECK_ARG(al16(a->A_hi) && al16(a->A_lo) && al16(a->W_hi) && al16(a->W_lo) && al16(a->bias) && al16(a->R_hi) && al16(a->R_lo));
    DIDI_CHECK_ARG(al16(gamma) && al16(beta) && al16(y_hi) && al16(y_lo) && al16(y_f32) && al16(stats));
    DIDI_CHECK_ARG(lo_bytes >= 0 && lo_bytes <= 3 && (!(lo_bytes & 1) || a->R_lo) && (!(lo_bytes & 2) || y_lo));
    if (lo_bytes && a->precision != DIDI_PREC_FP16 && a->precision != DIDI_PREC_FP16X2) return DIDI_ERR_UNSUPPORTED;
    if (a->M == 0) return DIDI_OK;
    const DeviceInfo* di = device_info();  // of the current device: opt-ins done, CU count known
    if (!di || di->ncu < 4) return DIDI_ERR_LAUNCH;
    const int ncu = di->ncu;
    GemmParams p{};
    p.N = a->N; p.K = a->K;
    int rc = fill_operands(p, a->precision, a->A_hi, a->A_lo, a->W_hi, a->W_lo);
    if (rc != DIDI_OK) return rc;
    p.lda = a->lda; p.ldw = a->ldw; p.bias = a->bias; p.act = DIDI_ACT_NONE; p.ldr = a->ldr; p.ldc = a->N;
    p.ln_gamma = gamma; p.ln_beta = beta; p.ln_eps = eps; p.tiles_n = 4; p.fault = a->fault;
    DIDI_CHECK_ARG(a->act_dither >= 0 && a->act_dither < 8192);
    p.dither = a->act_dither;
    p.lo8 = lo_bytes;
#ifdef DIDI_STAMPS
    p.stamps = g_stamp_buf;
#endif
    hipStream_t st = as_stream(stream);
    const int bands = cdiv(a->M, 256);
    if (epoch == 0) {  // stand-alone use: clear the granules, then tag 1
        if (hipMemsetAsync(stats, 0, didi_gemm_ln_stats_bytes(a->M), st) != hipSuccess) return DIDI_ERR_LAUNCH;
        epoch = 1;
    }
    p.ln_epoch = epoch;
    // row chunks of at most (CUs / 4) bands: every workgroup of a launch is resident, so band mates never wait for a CU
    const int bands_per_launch = ncu / 4;
    const size_t lds = PP_LDS_BYTES;
    for (int b0 = 0; b0 < bands; b0 += bands_per_launch) {
        const int nb = std::min(bands_per_launch, bands - b0);
        const size_t r0 = (size_t)b0 * 256;
        p.M = (int)std::min<size_t>((size_t)nb * 256, (size_t)a->M - r0);
        p.tiles_m = nb;
        p.A_hi = a->A_hi + r0 * a->lda; p.A_lo = (p.npass == 3 ? a->A_lo : a->A_hi) + r0 * a->lda;
        // byte planes: the same element offsets, one byte each
        p.R_hi = a->R_hi + r0 * a->ldr;
        p.R_lo = !a->R_lo ? nullptr : (lo_bytes & 1) ? reinterpret_cast<const u16*>(reinterpret_cast<const uint8_t*>(a->R_lo) + r0 * a->ldr) : a->R_lo + r0 * a->ldr;
        p.C_hi = y_hi + r0 * a->N;
        p.C_lo = !y_lo ? nullptr : (lo_bytes & 2) ? reinterpret_cast<u16*>(reinterpret_cast<uint8_t*>(y_lo) + r0 * a->N) : y_lo + r0 * a->N;
        p.C_f32 = y_f32 ? y_f32 + r0 * a->N : nullptr;
        p.ln_stats = reinterpret_cast<unsigned long long*>(stats) + (size_t)b0 * 4 * 512;
        const int grid = nb * 4;
        if (p.npass == 3 && p.f16) gemm_pp_kernel<3, true, true><<<grid, 512, lds, st>>>(p);
        else if (p.npass == 3) gemm_pp_kernel<3, false, true><<<grid, 512, lds, st>>>(p);
        else if (p.npass == 2) gemm_pp_kernel<2, true, true><<<grid, 512, lds, st>>>(p);
        else if (p.f16) gemm_pp_kernel<1, true, true><<<grid, 512, lds, st>>>(p);
        else gemm_pp_kernel<1, false, true><<<grid, 512, lds, st>>>(p);
        DIDI_LAUNCH_CHECK();
    }
    return DIDI_OK;
}

extern "C" int didi_gemm_ln(const didi_gemm_args* a, const float* gamma, const float* beta, float eps, uint16_t* y_hi,
                            uint16_t* y_lo, float* y_f32, void* stats, uint32_t epoch, void* stream) {
    return gemm_ln_impl(a, gamma, beta, eps, y_hi, y_lo, y_f32, stats, epoch, 0, stream);
}

extern "C" int didi_gemm_ln_lo8(const didi_gemm_args* a, const float* gamma, const float* beta, float eps, uint16_t* y_hi, void* y_lo,
                                float* y_f32, void* stats, uint32_t epoch, int lo_bytes, void* stream) {
    return gemm_ln_impl(a, gamma, beta, eps, y_hi, reinterpret_cast<uint16_t*>(y_lo), y_f32, stats, epoch, lo_bytes, stream);
}

// ------------------------------------------------------------------ rounding: classifier GEMM + argmax
// Each workgroup owns one 128-row tile and a chunk of vocabulary tiles; it keeps a per-lane running
// (max, index) and never writes logits.  Cross-lane / cross-wave reduction once per workgroup.
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

template <int MODE, int NSTAGE, bool F16, bool CE = false>
__global__ void __launch_bounds__(256, NSTAGE <= 2 ? 2 : 1) gemm_argmax_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int mt = blockIdx.x % p.tiles_m, chunk = blockIdx.x / p.tiles_m;
    const int m0 = mt * BM;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lg = lane >> 4;
    float best[4];
    int bidx[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) { best[mi] = -INFINITY; bidx[mi] = 0x7fffffff; }
    // CE variant: online log-sum-exp (running max rmax, sum of exp(v - rmax)) and the target's logit
    constexpr float LOG2E = 1.4426950408889634f;
    float rmax[4], rsum[4], tgt[4];
    int tg[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        rmax[mi] = -INFINITY; rsum[mi] = 0.f; tgt[mi] = -INFINITY; tg[mi] = -1;
        if constexpr (CE) {
            const int m = m0 + wm * 64 + mi * 16 + lr;
            if (m < p.M) {
                const int64_t t = p.targets[m];
                tg[mi] = (t >= 0 && t < p.N) ? (int)t : -1;
            }
        }
    }
    const int t_begin = chunk * p.tiles_per_chunk;
    const int t_end = min(t_begin + p.tiles_per_chunk, p.tiles_n);
    for (int tn = t_begin; tn < t_end; ++tn) {
        const int n0 = tn * BN;
        f32x4 acc[4][4];
        gemm_tile<MODE, NSTAGE, F16>(p, smem, m0, n0, acc);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn * 64 + ni * 16 + lg * 4 + r;
                    const float v = acc[ni][mi][r];
                    if (n < p.N && better(v, n, best[mi], bidx[mi])) { best[mi] = v; bidx[mi] = n; }
                }
        if constexpr (CE) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                float tmax = -INFINITY;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n0 + wn * 64 + ni * 16 + lg * 4 + r < p.N) tmax = fmaxf(tmax, acc[ni][mi][r]);
                if (tmax > -INFINITY) {
                    const float nm = fmaxf(rmax[mi], tmax);
                    float sacc = rsum[mi] * __builtin_amdgcn_exp2f((rmax[mi] - nm) * LOG2E);  // rmax = -inf: 0 * 0
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int n = n0 + wn * 64 + ni * 16 + lg * 4 + r;
                            if (n < p.N) {
                                sacc += __builtin_amdgcn_exp2f((acc[ni][mi][r] - nm) * LOG2E);
                                if (n == tg[mi]) tgt[mi] = acc[ni][mi][r];
                            }
                        }
                    rsum[mi] = sacc; rmax[mi] = nm;
                }
            }
        }
    }
    auto lse_merge = [&](float& m1, float& s1, float m2, float s2) {
        const float nm = fmaxf(m1, m2);
        if (nm == -INFINITY) { s1 = 0.f; return; }
        s1 = s1 * __builtin_amdgcn_exp2f((m1 - nm) * LOG2E) + s2 * __builtin_amdgcn_exp2f((m2 - nm) * LOG2E);
        m1 = nm;
    };
    // lanes lr, lr+16, lr+32, lr+48 hold the same rows
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float ov = __shfl_xor(best[mi], o, 64);
            const int oi = __shfl_xor(bidx[mi], o, 64);
            if (better(ov, oi, best[mi], bidx[mi])) { best[mi] = ov; bidx[mi] = oi; }
            if constexpr (CE) {
                const float om = __shfl_xor(rmax[mi], o, 64), os = __shfl_xor(rsum[mi], o, 64), ot = __shfl_xor(tgt[mi], o, 64);
                lse_merge(rmax[mi], rsum[mi], om, os);
                tgt[mi] = fmaxf(tgt[mi], ot);
            }
        }
    }
    __syncthreads();  // all waves are done with the staging buffers
    float* rv = reinterpret_cast<float*>(smem);       // [128]
    int* ri = reinterpret_cast<int*>(smem + 512);      // [128]
    float* rm = reinterpret_cast<float*>(smem + 1024), *rs = rm + 128, *rt = rs + 128;  // CE: [128] each
    if (wn == 1 && lg == 0) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int ml = wm * 64 + mi * 16 + lr;
            rv[ml] = best[mi]; ri[ml] = bidx[mi];
            if constexpr (CE) { rm[ml] = rmax[mi]; rs[ml] = rsum[mi]; rt[ml] = tgt[mi]; }
        }
    }
    __syncthreads();
    if (wn == 0 && lg == 0) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int ml = wm * 64 + mi * 16 + lr;
            const int m = m0 + ml;
            if (m < p.M) {
                float v = best[mi]; int i = bidx[mi];
                if (better(rv[ml], ri[ml], v, i)) { v = rv[ml]; i = ri[ml]; }
                p.partial[(size_t)m * p.nchunks + chunk] = make_float2(v, __int_as_float(i));
                if constexpr (CE) {  // the chunk max is v: the sum is relative to it
                    float cm = rmax[mi], cs = rsum[mi];
                    lse_merge(cm, cs, rm[ml], rs[ml]);
                    p.partial2[(size_t)m * p.nchunks + chunk] = make_float2(cs, fmaxf(tgt[mi], rt[ml]));
                }
            }
        }
    }
}

// one wave per row: the lanes read the row's chunk results 64 at a time (coalesced), then a butterfly picks the winner
__global__ void __launch_bounds__(256) argmax_reduce_kernel(const float2* partial, int M, int nchunks, int V, int64_t* ids64, int32_t* ids32) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int c = lane; c < nchunks; c += 64) {
        const float2 q = partial[(size_t)m * nchunks + c];
        const int i = __float_as_int(q.y);
        if (better(q.x, i, bv, bi)) { bv = q.x; bi = i; }
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) {
        if (bi < 0 || bi >= V) bi = 0;
        if (ids64) ids64[m] = bi;
        if (ids32) ids32[m] = bi;
    }
}

// CE variant: ids as above, nll[m] = logsumexp(logits[m, :]) - logits[m, target[m]] (+inf for a target outside [0, V))
__global__ void __launch_bounds__(256) ce_reduce_kernel(const float2* partial, const float2* partial2, int M, int nchunks, int V,
                                                        int64_t* ids64, float* nll) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int c = 0; c < nchunks; ++c) {
        const float2 q = partial[(size_t)m * nchunks + c];
        const int i = __float_as_int(q.y);
        if (better(q.x, i, bv, bi)) { bv = q.x; bi = i; }
    }
    float ssum = 0.f, t = -INFINITY;
    for (int c = 0; c < nchunks; ++c) {
        const float2 q = partial[(size_t)m * nchunks + c];
        const float2 r = partial2[(size_t)m * nchunks + c];
        ssum += r.x * expf(q.x - bv);
        t = fmaxf(t, r.y);
    }
    if (bi < 0 || bi >= V) bi = 0;
    if (ids64) ids64[m] = bi;
    nll[m] = (bv + logf(ssum)) - t;
}

extern "C" size_t didi_round_argmax_bytes(int M, int V) {
    if (M <= 0 || V <= 0) return 0;
    return (size_t)M * (size_t)cdiv(V, BN) * sizeof(float2);
}

extern "C" size_t didi_round_ce_bytes(int M, int V) { return 2 * didi_round_argmax_bytes(M, V); }

#define round_impl didi_round_impl  // shared with round_static.hip (declared in common.hpp)

extern "C" int didi_round_argmax(const uint16_t* X_hi, const uint16_t* X_lo, int ldx, const uint16_t* W_hi,
                                 const uint16_t* W_lo, int ldw, int M, int V, int K, int precision, void* partial,
                                 int64_t* ids64, int32_t* ids32, void* stream) {
    DIDI_CHECK_ARG(ids64 || ids32);
    return round_impl(X_hi, X_lo, ldx, W_hi, W_lo, ldw, M, V, K, precision, partial, ids64, ids32, nullptr, nullptr, stream);
}

extern "C" int didi_round_ce(const uint16_t* X_hi, const uint16_t* X_lo, int ldx, const uint16_t* W_hi, const uint16_t* W_lo,
                             int ldw, int M, int V, int K, int precision, const int64_t* targets, void* partial, int64_t* ids64,
                             float* nll, void* stream) {
    DIDI_CHECK_ARG(targets && nll);
    return round_impl(X_hi, X_lo, ldx, W_hi, W_lo, ldw, M, V, K, precision, partial, ids64, nullptr, targets, nll, stream);
}

int didi_round_impl(const uint16_t* X_hi, const uint16_t* X_lo, int ldx, const uint16_t* W_hi, const uint16_t* W_lo, int ldw,
                    int M, int V, int K, int precision, void* partial, int64_t* ids64, int32_t* ids32, const int64_t* targets,
                    float* nll, void* stream) {
    const bool ce = targets != nullptr;
    DIDI_CHECK_ARG(X_hi && W_hi && partial && M >= 0 && V > 0 && K > 0);
    if (K % 8 != 0) return DIDI_ERR_UNSUPPORTED;
    DIDI_CHECK_ARG(ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K);
    DIDI_CHECK_ARG(al16(X_hi) && al16(X_lo) && al16(W_hi) && al16(W_lo) && al8(partial));
    if (M == 0) return DIDI_OK;
    GemmParams p{};
    p.M = M; p.N = V; p.K = K;
    int rc = fill_operands(p, precision, X_hi, X_lo, W_hi, W_lo);
    if (rc != DIDI_OK) return rc;
    p.lda = ldx; p.ldw = ldw;
    hipStream_t st = as_stream(stream);
    if (!ce && M >= 2048 && K >= 128 && pick_mode(p) != 0 && gemm_pp_enabled() && gemm_big_enabled()) {
        // large batches: the 256x192 staggered loop with the argmax epilogue, one workgroup per (row band, 192 vocabulary columns)
        if (!allow_big_lds()) return DIDI_ERR_LAUNCH;
        p.tiles_m = cdiv(M, 256); p.tiles_n = cdiv(V, BBN); p.nchunks = p.tiles_n;
        p.partial = reinterpret_cast<float2*>(partial);
        const int grid = p.tiles_m * p.tiles_n;
        const size_t lds = PP_LDS_BYTES;
        if (p.npass == 3 && p.f16) gemm_pp_kernel<3, true, false, true><<<grid, 512, lds, st>>>(p);
        else if (p.npass == 3) gemm_pp_kernel<3, false, false, true><<<grid, 512, lds, st>>>(p);
        else if (p.npass == 2) gemm_pp_kernel<2, true, false, true><<<grid, 512, lds, st>>>(p);
        else if (p.f16) gemm_pp_kernel<1, true, false, true><<<grid, 512, lds, st>>>(p);
        else gemm_pp_kernel<1, false, false, true><<<grid, 512, lds, st>>>(p);
        DIDI_LAUNCH_CHECK();
        argmax_reduce_kernel<<<cdiv(M, 4), 256, 0, st>>>(p.partial, M, p.nchunks, V, ids64, ids32);
        DIDI_LAUNCH_CHECK();
        return DIDI_OK;
    }
    p.tiles_m = cdiv(M, BM); p.tiles_n = cdiv(V, BN);
    int nchunks = cdiv(1024, p.tiles_m);  // aim at >= 4 workgroups per CU
    if (nchunks > p.tiles_n) nchunks = p.tiles_n;
    if (nchunks < 1) nchunks = 1;
    p.tiles_per_chunk = cdiv(p.tiles_n, nchunks);
    p.nchunks = cdiv(p.tiles_n, p.tiles_per_chunk);
    p.partial = reinterpret_cast<float2*>(partial);
    p.partial2 = p.partial + (size_t)M * cdiv(V, BN);  // second half of didi_round_ce_bytes
    p.targets = targets;
    const int grid = p.tiles_m * p.nchunks;
#define DIDI_ROUND_LAUNCH(CE_)                                                                           \
    switch (pick_mode(p)) {                                                                              \
        case 3:                                                                                          \
            if (p.f16) gemm_argmax_kernel<3, 2, true, CE_><<<grid, 256, SMEM_BYTES, st>>>(p);            \
            else gemm_argmax_kernel<3, 2, false, CE_><<<grid, 256, SMEM_BYTES, st>>>(p);                 \
            break;                                                                                       \
        case 2: gemm_argmax_kernel<2, 2, true, CE_><<<grid, 256, SMEM_BYTES, st>>>(p); break;            \
        case 1:                                                                                          \
            if (p.f16) gemm_argmax_kernel<1, 2, true, CE_><<<grid, 256, SMEM_BYTES, st>>>(p);            \
            else gemm_argmax_kernel<1, 2, false, CE_><<<grid, 256, SMEM_BYTES, st>>>(p);                 \
            break;                                                                                       \
        default:                                                                                         \
            if (p.f16) gemm_argmax_kernel<0, 2, true, CE_><<<grid, 256, SMEM_BYTES, st>>>(p);            \
            else gemm_argmax_kernel<0, 2, false, CE_><<<grid, 256, SMEM_BYTES, st>>>(p);                 \
            break;                                                                                       \
    }
    if (ce) { DIDI_ROUND_LAUNCH(true) } else { DIDI_ROUND_LAUNCH(false) }
#undef DIDI_ROUND_LAUNCH
    DIDI_LAUNCH_CHECK();
    if (ce) ce_reduce_kernel<<<cdiv(M, 256), 256, 0, st>>>(p.partial, p.partial2, M, p.nchunks, V, ids64, nll);
    else argmax_reduce_kernel<<<cdiv(M, 4), 256, 0, st>>>(p.partial, M, p.nchunks, V, ids64, ids32);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}
