// LDS-resident fused attention for short dialogs (Sq, Sk <= 96, head dim 32/64).
//
// One 256-thread workgroup per (dialog, head).  Q, K and V^T of that pair are staged in LDS once
// (bf16 planes, zero-padded to MFMA granularity), then each wave owns 16-query blocks:
//   scores  S = Q K^T          v_mfma_f32_16x16x32_bf16, A-operand = Q rows, B-operand = K rows
//                              -> lane holds S[q = 4*(lane>>4)+r][key = 16*kt + (lane&15)]
//   softmax in registers       row max / row sum: in-lane over key tiles, then a 16-lane butterfly
//   P -> LDS (per-wave strip)  so that P can be re-read as the A-operand (8 consecutive keys/lane)
//   O = P V                    B-operand = rows of V^T (V is transposed while it is staged, so the
//                              8 keys a lane needs are one 16-byte LDS read)
// fp32 everywhere except the MFMA operands.  In bf16x3 precision every product is three MFMAs
// (lo*hi, hi*lo, hi*hi) on split planes, P included.
//
// Masking follows HF's additive-mask semantics (see include/didi_hip.h): masked (causal or padded
// context) keys take the most negative finite float; LDS padding keys are excluded outright.
// LDS rows are padded by 16 bytes to spread ds_read_b128 over the banks.
#include "common.hpp"

struct AttnParams {
    int B, heads, Sq, Sk;
    int causal;
    float scale;
    const int32_t* key_mask;
    const u16* Q[2]; int ldq;
    const u16* K[2]; int ldk;
    const u16* V[2]; int ldv;
    u16* O[2]; int ldo;
    int SP, CP;  // padded lengths: SP % 16 == 0, CP % 32 == 0
    int d;       // true head dim (multiple of 8, <= D); columns d..D-1 are zero padding in LDS
};

template <int D, int NPASS>
__global__ void __launch_bounds__(256) attn_kernel(AttnParams p) {
    constexpr int NPL = NPASS == 3 ? 2 : 1;
    constexpr int QS = D * 2 + 16;  // bytes per Q/K row in LDS
    constexpr float NEG = -3.4028234663852886e38f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int head = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int SP = p.SP, CP = p.CP;
    const int VS = CP * 2 + 16;  // bytes per V^T row / P row
    char* Qs = smem;
    char* Ks = Qs + NPL * SP * QS;
    char* Vt = Ks + NPL * CP * QS;
    char* Ps = Vt + NPL * D * VS;
    int* mk = reinterpret_cast<int*>(Ps + 4 * NPL * 16 * VS);  // [CP] 0 visible, 1 masked, 2 padding

    // ---- stage Q, K, V^T, mask
    constexpr int CH = D / 8;  // 16-byte chunks per row
    for (int pl = 0; pl < NPL; ++pl) {
        const u16* q = p.Q[pl] + (size_t)b * p.Sq * p.ldq + head * p.d;
        for (int i = tid; i < SP * CH; i += 256) {
            const int r = i / CH, c = i % CH;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < p.Sq && c * 8 < p.d) v = *reinterpret_cast<const uint4*>(q + (size_t)r * p.ldq + c * 8);
            *reinterpret_cast<uint4*>(Qs + pl * SP * QS + r * QS + c * 16) = v;
        }
        const u16* k = p.K[pl] + (size_t)b * p.Sk * p.ldk + head * p.d;
        const u16* vv = p.V[pl] + (size_t)b * p.Sk * p.ldv + head * p.d;
        for (int i = tid; i < CP * CH; i += 256) {
            const int r = i / CH, c = i % CH;
            uint4 kv = make_uint4(0, 0, 0, 0), v4 = make_uint4(0, 0, 0, 0);
            if (r < p.Sk && c * 8 < p.d) {
                kv = *reinterpret_cast<const uint4*>(k + (size_t)r * p.ldk + c * 8);
                v4 = *reinterpret_cast<const uint4*>(vv + (size_t)r * p.ldv + c * 8);
            }
            *reinterpret_cast<uint4*>(Ks + pl * CP * QS + r * QS + c * 16) = kv;
            const uint32_t w[4] = {v4.x, v4.y, v4.z, v4.w};
            char* vt = Vt + pl * D * VS + r * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const u16 x = (u16)(w[e >> 1] >> ((e & 1) * 16));
                *reinterpret_cast<u16*>(vt + (c * 8 + e) * VS) = x;
            }
        }
    }
    for (int c = tid; c < CP; c += 256) {
        int f = 0;
        if (c >= p.Sk) f = 2;
        else if (p.key_mask && p.key_mask[(size_t)b * p.Sk + c] == 0) f = 1;
        mk[c] = f;
    }
    __syncthreads();

    const int nqb = SP / 16;
    const int nkt_real = (p.Sk + 15) / 16;
    char* Pw = Ps + wave * NPL * 16 * VS;
    for (int it = 0; it * 4 < nqb; ++it) {
        const int qb = it * 4 + wave;
        const bool active = qb < nqb;
        const int nkt = active ? (p.causal ? min(qb + 1, nkt_real) : nkt_real) : 0;
        const int nks = (nkt + 1) / 2;
        float sc[6][4];
        float rmax[4] = {NEG, NEG, NEG, NEG}, rsum[4] = {0.f, 0.f, 0.f, 0.f};
        // ---- scores
#pragma unroll
        for (int kt = 0; kt < 6; ++kt) {
            if (kt < nkt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < D / 32; ++ks) {
                    const int qo = (qb * 16 + lr) * QS + (ks * 4 + lg) * 16;
                    const int ko = (kt * 16 + lr) * QS + (ks * 4 + lg) * 16;
                    const bf16x8 qh = *reinterpret_cast<const bf16x8*>(Qs + qo);
                    const bf16x8 kh = *reinterpret_cast<const bf16x8*>(Ks + ko);
                    if constexpr (NPASS == 3) {
                        const bf16x8 ql = *reinterpret_cast<const bf16x8*>(Qs + SP * QS + qo);
                        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(Ks + CP * QS + ko);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ql, kh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh, kl, acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh, kh, acc, 0, 0, 0);
                }
                const int key = kt * 16 + lr;
                const int f = mk[key];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qb * 16 + lg * 4 + r;
                    float s = acc[r] * p.scale;
                    if (f == 1 || (p.causal && key > q)) s = NEG;
                    if (f == 2) s = -INFINITY;
                    sc[kt][r] = s;
                    rmax[r] = fmaxf(rmax[r], s);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[kt][r] = -INFINITY;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) rmax[r] = fmaxf(rmax[r], __shfl_xor(rmax[r], o, 64));
        }
        // ---- probabilities -> LDS strip (unnormalised; the row sum divides O at the end)
#pragma unroll
        for (int kt = 0; kt < 6; ++kt) {
            if (kt < 2 * nks) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = (kt < nkt) ? expf(sc[kt][r] - rmax[r]) : 0.f;
                    rsum[r] += pr;
                    u16 h, l;
                    split2(pr, h, l);
                    const int po = (lg * 4 + r) * VS + (kt * 16 + lr) * 2;
                    *reinterpret_cast<u16*>(Pw + po) = h;
                    if constexpr (NPASS == 3) *reinterpret_cast<u16*>(Pw + 16 * VS + po) = l;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) rsum[r] += __shfl_xor(rsum[r], o, 64);
        }
        __syncthreads();
        // ---- O = P V
        if (active) {
            u16* oh = p.O[0] + (size_t)b * p.Sq * p.ldo + head * p.d;
            u16* ol = NPL == 2 ? p.O[1] + (size_t)b * p.Sq * p.ldo + head * p.d : nullptr;
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int ks = 0; ks < nks; ++ks) {
                    const int po = lr * VS + (ks * 4 + lg) * 16;
                    const int vo = (dt * 16 + lr) * VS + (ks * 4 + lg) * 16;
                    const bf16x8 ph = *reinterpret_cast<const bf16x8*>(Pw + po);
                    const bf16x8 vh = *reinterpret_cast<const bf16x8*>(Vt + vo);
                    if constexpr (NPASS == 3) {
                        const bf16x8 pl = *reinterpret_cast<const bf16x8*>(Pw + 16 * VS + po);
                        const bf16x8 vl = *reinterpret_cast<const bf16x8*>(Vt + D * VS + vo);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, vh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vl, acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vh, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qb * 16 + lg * 4 + r;
                    if (q < p.Sq && dt * 16 + lr < p.d) {
                        const float o = acc[r] / rsum[r];
                        u16 h, l;
                        split2(o, h, l);
                        const size_t off = (size_t)q * p.ldo + dt * 16 + lr;
                        oh[off] = h;
                        if (ol) ol[off] = l;
                    }
                }
            }
        }
        __syncthreads();
    }
}

static size_t attn_lds_bytes(int D, int npl, int SP, int CP) {
    const size_t QS = D * 2 + 16, VS = CP * 2 + 16;
    return npl * SP * QS + npl * CP * QS + npl * D * VS + 4 * npl * 16 * VS + CP * sizeof(int);
}

extern "C" int didi_attention(const didi_attn_args* a, void* stream) {
    DIDI_CHECK_ARG(a && a->Q_hi && a->K_hi && a->V_hi && a->O_hi);
    DIDI_CHECK_ARG(a->B >= 0 && a->heads > 0 && a->Sq > 0 && a->Sk > 0);
    if (a->D < 8 || a->D > 64 || a->D % 8 != 0 || a->Sq > 96 || a->Sk > 96) return DIDI_ERR_UNSUPPORTED;
    const int DP = a->D <= 32 ? 32 : 64;  // head dim padded to the MFMA K granularity
    if (a->causal && a->Sq != a->Sk) return DIDI_ERR_BAD_ARG;
    DIDI_CHECK_ARG(a->ldq % 8 == 0 && a->ldk % 8 == 0 && a->ldv % 8 == 0 && a->ldo > 0);
    const bool x3 = a->precision == DIDI_PREC_BF16X3;
    DIDI_CHECK_ARG(x3 || a->precision == DIDI_PREC_BF16);
    if (x3) DIDI_CHECK_ARG(a->Q_lo && a->K_lo && a->V_lo && a->O_lo);
    const void* ptrs[6] = {a->Q_hi, a->Q_lo, a->K_hi, a->K_lo, a->V_hi, a->V_lo};
    for (const void* q : ptrs) DIDI_CHECK_ARG((reinterpret_cast<uintptr_t>(q) & 15) == 0);
    if (a->B == 0) return DIDI_OK;
    AttnParams p{};
    p.B = a->B; p.heads = a->heads; p.Sq = a->Sq; p.Sk = a->Sk;
    p.causal = a->causal; p.scale = 1.0f / sqrtf((float)a->D); p.d = a->D; p.key_mask = a->key_mask;
    p.Q[0] = a->Q_hi; p.Q[1] = a->Q_lo; p.ldq = a->ldq;
    p.K[0] = a->K_hi; p.K[1] = a->K_lo; p.ldk = a->ldk;
    p.V[0] = a->V_hi; p.V[1] = a->V_lo; p.ldv = a->ldv;
    p.O[0] = a->O_hi; p.O[1] = x3 ? a->O_lo : nullptr; p.ldo = a->ldo;
    p.SP = (a->Sq + 15) / 16 * 16;
    p.CP = (a->Sk + 31) / 32 * 32;
    const size_t lds = attn_lds_bytes(DP, x3 ? 2 : 1, p.SP, p.CP);
    dim3 grid(a->heads, a->B);
    hipStream_t st = as_stream(stream);
#define DIDI_ATTN_LAUNCH(DD, NP)                                                                          \
    do {                                                                                                  \
        if (lds > 65536 &&                                                                                \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<DD, NP>),                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)      \
            return DIDI_ERR_LAUNCH;                                                                       \
        attn_kernel<DD, NP><<<grid, 256, lds, st>>>(p);                                                   \
    } while (0)
    if (DP == 64) { if (x3) DIDI_ATTN_LAUNCH(64, 3); else DIDI_ATTN_LAUNCH(64, 1); }
    else { if (x3) DIDI_ATTN_LAUNCH(32, 3); else DIDI_ATTN_LAUNCH(32, 1); }
#undef DIDI_ATTN_LAUNCH
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}
