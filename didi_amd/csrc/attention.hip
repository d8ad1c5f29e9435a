// LDS-resident fused attention for short dialogs (Sq, Sk <= 96, head dim <= 64).
//
// One 256-thread workgroup per (dialog, head).  K and V of that pair are staged ROW-MAJOR in LDS
// once (bf16 planes, zero-padded to MFMA granularity, XOR-swizzled 16-byte chunks instead of row
// padding); Q never touches LDS.  Each wave owns 16-query blocks and runs them without any
// workgroup barrier:
//   scores, transposed   S^T = K Q^T    v_mfma_f32_16x16x32_bf16, A-operand = K rows (ds_read_b128),
//                                       B-operand = Q rows read straight from global memory
//                                       -> lane holds S[q = lane&15][key = 16*kt + 4*(lane>>4) + r]
//   softmax in registers                one query per lane: row max / row sum in-lane over the key
//                                       tiles, then two cross-lane steps (xor 16, 32)
//   O^T = V^T P^T                       the probabilities a lane holds ARE its B-operand elements
//                                       (k index = the lane's own keys), and the matching V^T
//                                       A-operand comes from the row-major V image through
//                                       ds_read_b64_tr_b16 (hardware transpose): no V^T staging,
//                                       no round trip of P through LDS
//                                       -> lane holds O[q = lane&15][d = 16*dt + 4*(lane>>4) + r]:
//                                       4 consecutive columns, one 8-byte store per plane
// fp32 everywhere except the MFMA operands.  In bf16x3 precision every product is three MFMAs
// (lo*hi, hi*lo, hi*hi) on split planes, P included.
//
// Masking follows HF's additive-mask semantics (see include/didi_hip.h): masked (causal or padded
// context) keys take the most negative finite float; LDS padding keys are excluded outright.
//
// ds_read_b64_tr_b16 needs EXEC all ones: every branch around it below is wave-uniform.
#include <stdlib.h>

#include "common.hpp"

struct AttnParams {
    int B, heads, Sq, Sk;
    int causal;
    float scale;  // 1/sqrt(D) * log2(e): the softmax runs in base 2
    const int32_t* key_mask;
    const int32_t* key_count;  // [B] or null: key_mask[b] is 1 for the first key_count[b] keys and 0 after (right padding)
    const float* pos_bias;  // [heads][Sq][Sk] additive score bias (T5 relative positions) or null
    const u16* Q[2]; int ldq;
    const u16* K[2]; int ldk;
    const u16* V[2]; int ldv;
    u16* O[2]; int ldo;
    int SP, CP;  // padded lengths: SP % 16 == 0, CP % 32 == 0
    int d;       // true head dim (multiple of 8, <= D); columns d..D-1 are zero padding in LDS
    int nh;      // heads per workgroup (divides heads)
    int dither;  // one-product fp16: rounding dither of the probability and output planes (common.hpp f16x2_dither), 0 = RNE
};

typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));

// Byte offset of 16-byte chunk `ch` of row `row` in a [rows][D] bf16 LDS image.  The XOR makes both
// the ds_read_b128 row reads of the 16x16x32 A-operand and the transposed reads conflict-free
// (same swizzles as the GEMM's 128-byte and 64-byte rows).
template <int D>
__device__ __forceinline__ int img_off(int row, int ch) {
    if constexpr (D == 64) return row * 128 + ((ch ^ (row & 7)) << 4);
    else return row * 64 + ((ch ^ ((-(row >> 2)) & 3)) << 4);
}

__device__ __forceinline__ v4s lds_tr_read(const char* ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)ptr);
}

__device__ __forceinline__ bf16x8 cat8(v4s a, v4s b) {
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

__device__ __forceinline__ bf16x8 ld_q(const u16* q, bool ok) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ok) v = *reinterpret_cast<const uint4*>(q);
    return __builtin_bit_cast(bf16x8, v);
}

// NJ = 16-byte chunks of one K (or V) plane image each thread stages: ceil(CP * D/8 / 256), 2 or 3
template <int D, int NPASS, bool F16, int NJ, bool MULTI>
__global__ void __launch_bounds__(256) attn_kernel(AttnParams p) {
    constexpr int NPL = NPASS == 3 ? 2 : 1;
    constexpr int CH = D / 8;   // 16-byte chunks per row
    constexpr int KS = D / 32;  // MFMA k-steps over the head dim
    constexpr float NEG = -3.4028234663852886e38f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y, head0 = blockIdx.x * p.nh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int CP = p.CP;
    const int plane = CP * D * 2;  // bytes per K / V plane image
    char* Ks = smem;
    char* Vs = Ks + NPL * plane;
    float* mb = reinterpret_cast<float*>(Vs + NPL * plane);  // [CP] additive key bias: 0, NEG (masked) or -inf (padding)

    const int nqb = p.SP / 16;
    // right-padded contexts (every CommonSense batch: mean length 24 of 64): keys from key_count[b] on are masked, their softmax
    // weight is exp2(NEG - max) = 0 exactly, so neither their K / V rows nor their 16-key tiles are touched.  One scalar load per
    // workgroup; a count of 0 (nothing visible: the reference's uniform-attention corner) keeps the full masked path.
    int kc = p.Sk;
    if (p.key_count) {
        const int c = __builtin_amdgcn_readfirstlane(p.key_count[b]);
        if (c > 0 && c < p.Sk) kc = c;
    }
    const int nkt_real = (kc + 15) / 16;

    // Q fragments of one 16-query block: global -> registers (B-operand rows)
    bf16x8 qh0, qh1, ql0, ql1, nh0, nh1, nl0, nl1;
    qh0 = qh1 = ql0 = ql1 = nh0 = nh1 = nl0 = nl1 = __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0));
#define DIDI_LOAD_Q(H0, H1, L0, L1, head_, qb_)                                                        \
    do {                                                                                               \
        const int q_ = (qb_) * 16 + lr;                                                                \
        const size_t ro_ = ((size_t)b * p.Sq + q_) * p.ldq + (head_) * p.d;                            \
        const bool ok0_ = q_ < p.Sq && lg * 8 < p.d;                                                   \
        const bool ok1_ = q_ < p.Sq && 32 + lg * 8 < p.d;                                              \
        H0 = ld_q(p.Q[0] + ro_ + lg * 8, ok0_);                                                        \
        if constexpr (KS == 2) H1 = ld_q(p.Q[0] + ro_ + 32 + lg * 8, ok1_);                            \
        if constexpr (NPL == 2) {                                                                      \
            L0 = ld_q(p.Q[1] + ro_ + lg * 8, ok0_);                                                    \
            if constexpr (KS == 2) L1 = ld_q(p.Q[1] + ro_ + 32 + lg * 8, ok1_);                        \
        }                                                                                              \
    } while (0)

    // K, V of one head: global -> registers (zero past Sk / past the true head dim), registers -> LDS
    uint4 kr[NJ][NPL], vr[NJ][NPL];
#define DIDI_LOAD_KV(head_)                                                                            \
    do {                                                                                               \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                               \
            const int i_ = tid + j * 256;                                                              \
            const int r_ = i_ / CH, c_ = i_ % CH;                                                      \
            const bool ok_ = r_ < kc && c_ * 8 < p.d;                                                  \
            _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) {                                       \
                kr[j][pl] = make_uint4(0, 0, 0, 0);                                                    \
                vr[j][pl] = make_uint4(0, 0, 0, 0);                                                    \
                if (ok_) {                                                                             \
                    kr[j][pl] = *reinterpret_cast<const uint4*>(p.K[pl] + ((size_t)b * p.Sk + r_) * p.ldk + (head_) * p.d + c_ * 8); \
                    vr[j][pl] = *reinterpret_cast<const uint4*>(p.V[pl] + ((size_t)b * p.Sk + r_) * p.ldv + (head_) * p.d + c_ * 8); \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
    } while (0)
#define DIDI_STORE_KV()                                                                                \
    do {                                                                                               \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                               \
            const int i_ = tid + j * 256;                                                              \
            const int r_ = i_ / CH, c_ = i_ % CH;                                                      \
            if (r_ < CP) {                                                                             \
                const int o_ = img_off<D>(r_, c_);                                                     \
                _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) {                                   \
                    *reinterpret_cast<uint4*>(Ks + pl * plane + o_) = kr[j][pl];                       \
                    *reinterpret_cast<uint4*>(Vs + pl * plane + o_) = vr[j][pl];                       \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
    } while (0)

    if (wave < nqb) DIDI_LOAD_Q(qh0, qh1, ql0, ql1, head0, wave);
    DIDI_LOAD_KV(head0);
    for (int c = tid; c < CP; c += 256) {
        float f = 0.f;
        if (c >= kc) f = -INFINITY;
        else if (p.key_mask && p.key_mask[(size_t)b * p.Sk + c] == 0) f = NEG;
        mb[c] = f;
    }
    DIDI_STORE_KV();
    __syncthreads();

    // a workgroup walks p.nh heads of its dialog: the next head's K, V and Q are in flight (registers)
    // while the current head is computed from LDS
    for (int hh = 0; hh < (MULTI ? p.nh : 1); ++hh) {
        const int head = head0 + hh;
        const bool more = MULTI && hh + 1 < p.nh;
        if (more) {
            DIDI_LOAD_KV(head + 1);
            if (wave < nqb) DIDI_LOAD_Q(nh0, nh1, nl0, nl1, head + 1, wave);
        }
        u16* oh = p.O[0] + (size_t)b * p.Sq * p.ldo + head * p.d;
        u16* ol = NPL == 2 ? p.O[1] + (size_t)b * p.Sq * p.ldo + head * p.d : nullptr;

        for (int qb = wave; qb < nqb; qb += 4) {
            if (qb != wave) DIDI_LOAD_Q(qh0, qh1, ql0, ql1, head, qb);
            const int nkt = p.causal ? min(qb + 1, nkt_real) : nkt_real;
            const int nks = (nkt + 1) / 2;
            const int q = qb * 16 + lr;
            float sc[6][4];
            float rmax = NEG, rsum = 0.f;
            // ---- scores: lane holds S[q][16*kt + 4*lg + r]
#pragma unroll
            for (int kt = 0; kt < 6; ++kt) {
                if (kt < nkt) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const int ko = img_off<D>(kt * 16 + lr, ks * 4 + lg);
                        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(Ks + ko);
                        const bf16x8 qh = ks ? qh1 : qh0;
                        if constexpr (NPASS == 3) {
                            const bf16x8 kl = *reinterpret_cast<const bf16x8*>(Ks + plane + ko);
                            const bf16x8 ql = ks ? ql1 : ql0;
                            acc = mma16<F16>(kl, qh, acc);
                            acc = mma16<F16>(kh, ql, acc);
                        }
                        acc = mma16<F16>(kh, qh, acc);
                    }
                    // masked keys: score + NEG rounds to NEG exactly, the reference's additive mask
                    const float4 m4 = *reinterpret_cast<const float4*>(mb + kt * 16 + lg * 4);
                    sc[kt][0] = fmaf(acc[0], p.scale, m4.x);
                    sc[kt][1] = fmaf(acc[1], p.scale, m4.y);
                    sc[kt][2] = fmaf(acc[2], p.scale, m4.z);
                    sc[kt][3] = fmaf(acc[3], p.scale, m4.w);
                    if (p.pos_bias && q < p.Sq) {  // the bias enters in natural-log units: scale by log2(e)
                        const float* pb = p.pos_bias + ((size_t)head * p.Sq + q) * p.Sk;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int kk = kt * 16 + lg * 4 + r;
                            if (kk < p.Sk) sc[kt][r] = fmaf(pb[kk], 1.4426950408889634f, sc[kt][r]);
                        }
                    }
                    if (p.causal && kt == qb) {  // only the diagonal tile holds future keys
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (kt * 16 + lg * 4 + r > q) sc[kt][r] = fminf(sc[kt][r], NEG);
                    }
                    rmax = fmaxf(rmax, fmaxf(fmaxf(sc[kt][0], sc[kt][1]), fmaxf(sc[kt][2], sc[kt][3])));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[kt][r] = -INFINITY;
                }
            }
            rmax = fmaxf(rmax, __shfl_xor(rmax, 16, 64));
            rmax = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
            // ---- probabilities (unnormalised; the row sum divides O at the end) -> B-operand fragments:
            //      k-step s, element e: key 32*s + 16*(e>>2) + 4*lg + (e&3).  Scores carry log2(e): one
            //      v_exp_f32 per probability; exp2(-inf) = 0 covers padding keys and absent tiles.
            v8s ph[3], pl[3];
            const int dith = NPASS == 1 && F16 ? p.dither : 0;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                ph[s] = v8s{0, 0, 0, 0, 0, 0, 0, 0};
                pl[s] = v8s{0, 0, 0, 0, 0, 0, 0, 0};
                if (s < nks) {
                    if (dith) {  // one-product fp16 call of a dithered loop: two probabilities per v_cvt_pkrtz
                        float pr[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            pr[e] = __builtin_amdgcn_exp2f(sc[2 * s + (e >> 2)][e & 3] - rmax);
                            rsum += pr[e];
                        }
                        const uint2 a0 = pack4_hi_dither(pr, dith), a1 = pack4_hi_dither(pr + 4, dith);
                        ph[s] = __builtin_bit_cast(v8s, make_uint4(a0.x, a0.y, a1.x, a1.y));
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float pr = __builtin_amdgcn_exp2f(sc[2 * s + (e >> 2)][e & 3] - rmax);
                            rsum += pr;
                            u16 h, l;
                            split2(pr, h, l, F16);
                            ph[s][e] = (short)h;
                            pl[s][e] = (short)l;
                        }
                    }
                }
            }
            rsum += __shfl_xor(rsum, 16, 64);
            rsum += __shfl_xor(rsum, 32, 64);
            const float rinv = 1.0f / rsum;
            // ---- O^T = V^T P^T; A-operand element e of lane (lr, lg) = V[key(s, lg, e)][16*dt + lr]
            const int trow = 4 * lg + (lr >> 2);  // row of the 4x16 block this lane addresses
            const int tsub = 8 * (lr & 1);
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const int cch = 2 * dt + ((lr & 3) >> 1);
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    if (s < nks) {
                        const int o0 = img_off<D>(32 * s + trow, cch) + tsub;
                        const int o1 = img_off<D>(32 * s + 16 + trow, cch) + tsub;
                        const bf16x8 vh = cat8(lds_tr_read(Vs + o0), lds_tr_read(Vs + o1));
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, ph[s]);
                        if constexpr (NPASS == 3) {
                            const bf16x8 vl = cat8(lds_tr_read(Vs + plane + o0), lds_tr_read(Vs + plane + o1));
                            const bf16x8 bl = __builtin_bit_cast(bf16x8, pl[s]);
                            acc = mma16<F16>(vl, bh, acc);
                            acc = mma16<F16>(vh, bl, acc);
                        }
                        acc = mma16<F16>(vh, bh, acc);
                    }
                }
                const int d0 = dt * 16 + lg * 4;
                if (q < p.Sq && d0 < p.d) {
                    // the fp32 products are made opaque before the 16-bit conversion: hipcc otherwise fuses product and conversion into one
                    // v_fma_mixlo_f16 (a single rounding) in SOME instantiations, and the fused QKV form and attn_kernel would differ by an ulp in 1 of 50 000 values
                    float o[4] = {acc[0] * rinv, acc[1] * rinv, acc[2] * rinv, acc[3] * rinv};
                    asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
                    const size_t off = (size_t)q * p.ldo + d0;
                    if (ol) {
                        uint2 h2, l2;
                        split4(o, h2, l2, F16);
                        *reinterpret_cast<uint2*>(oh + off) = h2;
                        *reinterpret_cast<uint2*>(ol + off) = l2;
                    } else {
                        *reinterpret_cast<uint2*>(oh + off) = dith ? pack4_hi_dither(o, dith) : pack4_hi(o, F16);
                    }
                }
            }
        }
        if (more) {
            __syncthreads();  // every wave is done with this head's K / V image
            DIDI_STORE_KV();
            qh0 = nh0; qh1 = nh1; ql0 = nl0; ql1 = nl1;
            __syncthreads();
        }
    }
#undef DIDI_LOAD_Q
#undef DIDI_LOAD_KV
#undef DIDI_STORE_KV
}

static size_t attn_lds_bytes(int D, int npl, int CP) {
    return (size_t)2 * npl * CP * D * 2 + CP * sizeof(float);
}

extern "C" int didi_attention(const didi_attn_args* a, void* stream) {
    DIDI_CHECK_ARG(a && a->Q_hi && a->K_hi && a->V_hi && a->O_hi);
    DIDI_CHECK_ARG(a->B >= 0 && a->heads > 0 && a->Sq > 0 && a->Sk > 0);
    if (a->D < 8 || a->D > 64 || a->D % 8 != 0 || a->Sq > 96 || a->Sk > 96) return DIDI_ERR_UNSUPPORTED;
    const int DP = a->D <= 32 ? 32 : 64;  // head dim padded to the MFMA K granularity
    if (a->causal && a->Sq != a->Sk) return DIDI_ERR_BAD_ARG;
    DIDI_CHECK_ARG(a->scale >= 0.f && (reinterpret_cast<uintptr_t>(a->pos_bias) & 3) == 0);
    DIDI_CHECK_ARG(a->ldq % 8 == 0 && a->ldk % 8 == 0 && a->ldv % 8 == 0 && a->ldo > 0 && a->ldo % 4 == 0);
    const bool x3 = a->precision == DIDI_PREC_BF16X3 || a->precision == DIDI_PREC_FP16X3;
    const bool f16 = a->precision == DIDI_PREC_FP16 || a->precision == DIDI_PREC_FP16X2 ||  // no weights here: x2 = one product
                     a->precision == DIDI_PREC_FP16X3;                                       // x3: all three products on fp16 planes
    DIDI_CHECK_ARG(x3 || f16 || a->precision == DIDI_PREC_BF16);
    if (x3) DIDI_CHECK_ARG(a->Q_lo && a->K_lo && a->V_lo && a->O_lo);
    const void* ptrs[6] = {a->Q_hi, a->Q_lo, a->K_hi, a->K_lo, a->V_hi, a->V_lo};
    for (const void* q : ptrs) DIDI_CHECK_ARG((reinterpret_cast<uintptr_t>(q) & 15) == 0);
    DIDI_CHECK_ARG((reinterpret_cast<uintptr_t>(a->O_hi) & 7) == 0 && (reinterpret_cast<uintptr_t>(a->O_lo) & 7) == 0);
    if (a->B == 0) return DIDI_OK;
    AttnParams p{};
    p.B = a->B; p.heads = a->heads; p.Sq = a->Sq; p.Sk = a->Sk;
    p.causal = a->causal; p.d = a->D; p.key_mask = a->key_mask; p.key_count = a->key_mask ? a->key_count : nullptr;
    p.scale = 1.4426950408889634f * (a->scale > 0.f ? a->scale : 1.0f / sqrtf((float)a->D));  // softmax scale * log2(e)
    p.pos_bias = a->pos_bias;
    p.Q[0] = a->Q_hi; p.Q[1] = a->Q_lo; p.ldq = a->ldq;
    p.K[0] = a->K_hi; p.K[1] = a->K_lo; p.ldk = a->ldk;
    p.V[0] = a->V_hi; p.V[1] = a->V_lo; p.ldv = a->ldv;
    p.O[0] = a->O_hi; p.O[1] = x3 ? a->O_lo : nullptr; p.ldo = a->ldo;
    DIDI_CHECK_ARG(a->act_dither >= 0 && a->act_dither < 8192);
    p.dither = a->act_dither;
    p.SP = (a->Sq + 15) / 16 * 16;
    p.CP = (a->Sk + 31) / 32 * 32;
    const size_t lds = attn_lds_bytes(DP, x3 ? 2 : 1, p.CP);  // <= 48.4 KiB: 3-4 workgroups per CU
    // heads per workgroup (the MULTI form of the kernel).  Measured at B=256, 12 heads, S=64 (tools/attn_bench.py): one head
    // 19.2 / 20.9 us (fp16 self / cross), two heads 23.8 / 24.3, three 24.2 / 24.0 -- the kernel is bound by its per-wave
    // dependency chains, so 8 waves per SIMD beat prefetching with 4: one head per workgroup it is.
    p.nh = 1;
    dim3 grid(a->heads / p.nh, a->B);
    hipStream_t st = as_stream(stream);
    const bool nj3 = p.CP * (DP / 8) > 512;
#define DIDI_ATTN(DD, NP, FF)                                                        \
    do {                                                                             \
        if (p.nh > 1) {                                                              \
            if (nj3) attn_kernel<DD, NP, FF, 3, true><<<grid, 256, lds, st>>>(p);    \
            else attn_kernel<DD, NP, FF, 2, true><<<grid, 256, lds, st>>>(p);        \
        } else {                                                                     \
            if (nj3) attn_kernel<DD, NP, FF, 3, false><<<grid, 256, lds, st>>>(p);   \
            else attn_kernel<DD, NP, FF, 2, false><<<grid, 256, lds, st>>>(p);       \
        }                                                                            \
    } while (0)
    if (DP == 64) {
        if (x3 && f16) DIDI_ATTN(64, 3, true);
        else if (x3) DIDI_ATTN(64, 3, false);
        else if (f16) DIDI_ATTN(64, 1, true);
        else DIDI_ATTN(64, 1, false);
    } else {
        if (x3 && f16) DIDI_ATTN(32, 3, true);
        else if (x3) DIDI_ATTN(32, 3, false);
        else if (f16) DIDI_ATTN(32, 1, true);
        else DIDI_ATTN(32, 1, false);
    }
#undef DIDI_ATTN
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ key counts of right-padded masks
// counts[b] = number of 1s of mask[b, :] when they form a prefix (1...1 0...0), else Sk (= no trimming): what didi_attention's
// key_count takes.  One pass over the [B, Sk] mask, once per batch.
__global__ void __launch_bounds__(256) key_counts_kernel(const int32_t* mask, int B, int Sk, int32_t* counts) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    int n = 0;
    bool prefix = true;
    for (int c = 0; c < Sk; ++c) {
        const bool on = mask[(size_t)b * Sk + c] != 0;
        if (on && n != c) prefix = false;
        n += on;
    }
    counts[b] = prefix ? n : Sk;
}

extern "C" int didi_key_counts(const int32_t* mask, int B, int Sk, int32_t* counts, void* stream) {
    DIDI_CHECK_ARG(mask && counts && B >= 0 && Sk > 0);
    if (B == 0) return DIDI_OK;
    key_counts_kernel<<<cdiv(B, 256), 256, 0, as_stream(stream)>>>(mask, B, Sk, counts);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}
