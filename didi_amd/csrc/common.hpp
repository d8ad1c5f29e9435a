// Shared device/host helpers for libdidi_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/didi_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short u16;

#define DIDI_CHECK_ARG(cond)                 \
    do {                                     \
        if (!(cond)) return DIDI_ERR_BAD_ARG; \
    } while (0)

#define DIDI_LAUNCH_CHECK()                                      \
    do {                                                         \
        if (hipGetLastError() != hipSuccess) return DIDI_ERR_LAUNCH; \
    } while (0)

// ---- split 16-bit planes ----------------------------------------------------------------------
// An fp32 value x travels as two 16-bit planes: hi = rne16(x) and lo = rne16(x - hi).
//   bf16 planes (f16 == 0): hi + lo carries ~16 mantissa bits.  "bf16" GEMM mode multiplies hi planes
//     only; "bf16x3" adds the hi*lo and lo*hi cross terms into the same fp32 accumulator.
//   fp16 planes (f16 != 0): hi alone carries 11 bits (8x finer than bf16) and feeds the one-product
//     "fp16" mode; lo is kept only where a value is re-joined (residual adds): hi + lo ~ 22 bits.
__device__ __forceinline__ u16 bf16_bits(float x) { return __builtin_bit_cast(u16, (__bf16)x); }
__device__ __forceinline__ float bf16_to_f32(u16 b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }
__device__ __forceinline__ u16 f16_bits(float x) { return __builtin_bit_cast(u16, (_Float16)x); }
__device__ __forceinline__ float f16_to_f32(u16 b) { return (float)__builtin_bit_cast(_Float16, b); }

// fp16 planes saturate instead of overflowing.  MODE register bit 23 (FP16_OVFL): "an overflowed FP16 VALU result is clamped to
// +-MAX_FP16 regardless of round mode, while still preserving true INF values" -- so v_cvt_f16_f32 of |x| > 65504 gives +-65504
// at no cost per value (an explicit compare + select per value cost the issue-bound GEMM epilogues 7-11 % of the whole
// sampler, tools/ab_bench.sh); true infinities and NaNs pass through.  Every kernel that writes fp16 planes sets the bit once per
// wave; GEMM epilogues also track max |value| (one v_max3_f32 per two values) and report DIDI_FAULT_FP16_RANGE at the end.
#define DIDI_F16_MAX 65504.0f
__device__ __forceinline__ void f16_saturate_mode() { __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1 /* hwreg(HW_REG_MODE, 23, 1) */, 1); }
__device__ __forceinline__ float amax2(float m, float a, float b) { return fmaxf(fmaxf(m, fabsf(a)), fabsf(b)); }  // NaNs are ignored

__device__ __forceinline__ void split2(float x, u16& hi, u16& lo, int f16 = 0) {
    if (f16) {
        hi = f16_bits(x);
        lo = f16_bits(x - f16_to_f32(hi));
    } else {
        hi = bf16_bits(x);
        lo = bf16_bits(x - bf16_to_f32(hi));
    }
}
__device__ __forceinline__ float join2(u16 hi, u16 lo, int f16 = 0) {
    return f16 ? f16_to_f32(hi) + f16_to_f32(lo) : bf16_to_f32(hi) + bf16_to_f32(lo);
}

// ---- ordered rounding dither of fp16 ACTIVATION planes (didi_*_args.act_dither) -------------------------------------------
// Plain RNE is a fixed function of the value: in the late calls of a sampling loop, where consecutive denoiser inputs are nearly
// identical, every call rounds every activation the same way and the errors add up coherently over the loop (DESIGN.md
// section 2), exactly as a once-rounded weight does.  With d != 0 the hi plane is instead TRUNCATED after adding d / 8192 of an
// fp16 ulp to the magnitude (d in units of the fp32 mantissa's last place: the 13 bits fp16 drops); a loop passes
// d = 1024, 3072, 5120, 7168 in turn, so over four consecutive calls a static value is rounded down 1..4 times in proportion
// to its remainder and the MEAN of its four roundings is within ulp / 8 of it (plain RNE: ulp / 2, always the same side).
// One integer add per value; the conversion is v_cvt_pkrtz_f16_f32 (two values per instruction, overflow saturates to
// +-65504 by construction).  Zero stays zero (0 + d is an fp32 subnormal), NaN stays NaN; +-inf would become NaN: values
// reach these conversions through the range-checked epilogues only.  The remainder plane stays lo = rne16(x - hi).
__device__ __forceinline__ uint32_t f16x2_dither(float a, float b, int d) {
    const float a_ = __builtin_bit_cast(float, __builtin_bit_cast(int, a) + d), b_ = __builtin_bit_cast(float, __builtin_bit_cast(int, b) + d);
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a_, b_));
}
__device__ __forceinline__ uint2 pack4_hi_dither(const float v[4], int d) {
    return make_uint2(f16x2_dither(v[0], v[1], d), f16x2_dither(v[2], v[3], d));
}
// DIDI_RNE_AS_DITHER: the offset that makes the truncating conversion round to nearest (ties away from zero instead of to even:
// one value in 2^13 differs from RNE, by an ulp's rounding direction).  The 256x192 GEMM's fp16 epilogues convert through
// f16x2_dither always -- 1.5 VALU instructions per value against 2.5 for two v_cvt_f16_f32 and a v_perm_b32.
#define DIDI_RNE_AS_DITHER 4096
__device__ __forceinline__ void split4_dither(const float v[4], uint2& hi, uint2& lo, int d) {
    hi = pack4_hi_dither(v, d);
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 a = __builtin_bit_cast(h2, hi.x), b = __builtin_bit_cast(h2, hi.y);
    const float r[4] = {v[0] - (float)a[0], v[1] - (float)a[1], v[2] - (float)b[0], v[3] - (float)b[1]};
    lo = pack4_hi_dither(r, DIDI_RNE_AS_DITHER);
}

// The remainder plane as BYTES ("lo8"): byte = top byte (e5m2) of fp16(256 * (v - hi)), rounded half away from zero.  hi + lo8
// carries 14 bits and more (absolute error <= 2^-25 below |v| = 2^-10): what the residual stream of a one- or two-product call
// needs -- no GEMM of such a call multiplies an activation's remainder, it is only re-joined by the next residual add -- at 3
// instead of 4 bytes per value and direction in the launches whose epilogues are bound by exactly those bytes.  The factor 256
// keeps the remainders of small values out of fp16's subnormal range.
#define DIDI_LO8_SCALE 256.0f
__device__ __forceinline__ void split4_dither_lo8(const float v[4], uint2& hi, uint32_t& lo8, int d) {
    hi = pack4_hi_dither(v, d);
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 a = __builtin_bit_cast(h2, hi.x), b = __builtin_bit_cast(h2, hi.y);
    const float r[4] = {(v[0] - (float)a[0]) * DIDI_LO8_SCALE, (v[1] - (float)a[1]) * DIDI_LO8_SCALE, (v[2] - (float)b[0]) * DIDI_LO8_SCALE,
                        (v[3] - (float)b[1]) * DIDI_LO8_SCALE};
    const uint2 l = pack4_hi_dither(r, DIDI_RNE_AS_DITHER);
    lo8 = __builtin_amdgcn_perm(l.y + 0x00800080u, l.x + 0x00800080u, 0x07050301u);  // bytes 1 and 3 of both words
}
// four lo8 bytes -> four fp16 bit patterns (values 256 x the remainders)
__device__ __forceinline__ uint2 lo8_unpack4(uint32_t w) {
    return make_uint2(__builtin_amdgcn_perm(0u, w, 0x010c000cu), __builtin_amdgcn_perm(0u, w, 0x030c020cu));
}

// 4 consecutive values -> two 8-byte packs (hi plane, lo plane)
__device__ __forceinline__ void split4(const float v[4], uint2& hi, uint2& lo, int f16 = 0) {
    u16 h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split2(v[i], h[i], l[i], f16);
    hi = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    lo = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
}
// hi plane only (callers that do not keep the remainder)
__device__ __forceinline__ uint2 pack4_hi(const float v[4], int f16 = 0) {
    u16 h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = f16 ? f16_bits(v[i]) : bf16_bits(v[i]);
    return make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
}
__device__ __forceinline__ void unpack4(uint2 p, float v[4], int f16 = 0) {
    const u16 b[4] = {(u16)(p.x & 0xffff), (u16)(p.x >> 16), (u16)(p.y & 0xffff), (u16)(p.y >> 16)};
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = f16 ? f16_to_f32(b[i]) : bf16_to_f32(b[i]);
}

// one MFMA on 8 x 16-bit operand fragments held as raw bit patterns
template <bool F16>
__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// raises `bit` in the device fault word: one atomic per wave that saw the condition (rare: normally none)
__device__ __forceinline__ void raise_fault(uint32_t* fault, bool cond, uint32_t bit) {
    if (fault && cond) {  // the first active lane that saw it writes
        const unsigned long long act = __ballot(1);
        if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) atomicOr(fault, bit);
    }
}

// classifier GEMM + argmax (+ cross-entropy terms when targets != null) with caller-provided partials: gemm.hip
int didi_round_impl(const uint16_t* X_hi, const uint16_t* X_lo, int ldx, const uint16_t* W_hi, const uint16_t* W_lo, int ldw,
                    int M, int V, int K, int precision, void* partial, int64_t* ids64, int32_t* ids32, const int64_t* targets,
                    float* nll, void* stream);

// process-wide tuning switches (didi_set_option; engine.hip holds the table)
int didi_opt(int key);

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
