// Row / elementwise kernels of the sampler path: HBM-bound, one pass over the [M, H] state each.
//
// Layout: one 64-lane wave owns one row of H fp32 values, lane l holding the 16-byte chunks
// c = l + 64*j (so a wave-instruction reads 1 KiB contiguous); four rows per 256-thread block.
// LayerNorm statistics are two-pass in registers (mean, then sum of squared deviations) with a
// wave butterfly -- no LDS, no second trip to memory.  Algorithmic bytes per row are listed at
// each kernel; DESIGN.md section "Kernels" carries the roofline arithmetic.
#include "common.hpp"

#include "philox.hpp"

__global__ void __launch_bounds__(256) philox_normal_kernel(float* out, size_t n4, uint64_t seed, uint64_t offset) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float z[4];
        philox_normal4(offset + i, seed, z);
        reinterpret_cast<float4*>(out)[i] = make_float4(z[0], z[1], z[2], z[3]);
    }
}

extern "C" int didi_philox_normal(float* out, size_t n, uint64_t seed, uint64_t offset, void* stream) {
    DIDI_CHECK_ARG(out && n % 4 == 0);
    if (n == 0) return DIDI_OK;
    const size_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    philox_normal_kernel<<<grid, 256, 0, as_stream(stream)>>>(out, n4, seed, offset);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ fp32 -> split planes
// bytes per element: 4 read + 4 written.
__global__ void __launch_bounds__(256) split_kernel(const float* src, u16* hi, u16* lo, size_t n4, int f16) {
    if (f16) f16_saturate_mode();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        const float f[4] = {v.x, v.y, v.z, v.w};
        uint2 h, l;
        split4(f, h, l, f16);
        reinterpret_cast<uint2*>(hi)[i] = h;
        if (lo) reinterpret_cast<uint2*>(lo)[i] = l;
    }
}

static int split_planes(const float* src, uint16_t* hi, uint16_t* lo, size_t n, int f16, void* stream) {
    DIDI_CHECK_ARG(src && hi && n % 4 == 0);
    if (n == 0) return DIDI_OK;
    const size_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    split_kernel<<<grid, 256, 0, as_stream(stream)>>>(src, hi, lo, n4, f16);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}
extern "C" int didi_split_bf16(const float* src, uint16_t* hi, uint16_t* lo, size_t n, void* stream) {
    return split_planes(src, hi, lo, n, 0, stream);
}
extern "C" int didi_split_fp16(const float* src, uint16_t* hi, uint16_t* lo, size_t n, void* stream) {
    return split_planes(src, hi, lo, n, 1, stream);
}

// ------------------------------------------------------------------ LayerNorm tail shared by both row kernels
template <int MAXV>
__device__ __forceinline__ void ln_row_store(float (&v)[MAXV][4], int lane, int H, size_t row_off, const float* gamma,
                                             const float* beta, float eps, u16* y_hi, u16* y_lo, float* y_f32, int f16, int dith = 0) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane + 64 * j;
        if (c * 4 < H) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane + 64 * j;
        if (c * 4 < H) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[j][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane + 64 * j;
        if (c * 4 < H) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[c];
            const float4 b = reinterpret_cast<const float4*>(beta)[c];
            float y[4];
            y[0] = (v[j][0] - mean) * rstd * g.x + b.x;
            y[1] = (v[j][1] - mean) * rstd * g.y + b.y;
            y[2] = (v[j][2] - mean) * rstd * g.z + b.z;
            y[3] = (v[j][3] - mean) * rstd * g.w + b.w;
            const size_t o = row_off + (size_t)c * 4;
            if (y_f32) *reinterpret_cast<float4*>(y_f32 + o) = make_float4(y[0], y[1], y[2], y[3]);
            if (y_hi) {
                uint2 h, l;
                if (dith && f16) split4_dither(y, h, l, dith);
                else split4(y, h, l, f16);
                *reinterpret_cast<uint2*>(y_hi + o) = h;
                if (y_lo) *reinterpret_cast<uint2*>(y_lo + o) = l;
            }
        }
    }
}

// ------------------------------------------------------------------ LayerNorm
// bytes per row: 4H read + (2H | 4H) planes [+ 4H fp32] written.
template <int MAXV>
__global__ void __launch_bounds__(256) layernorm_kernel(const float* x, const float* gamma, const float* beta, float eps,
                                                        int M, int H, u16* y_hi, u16* y_lo, float* y_f32, int f16) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const size_t row_off = (size_t)m * H;
    float v[MAXV][4];
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane + 64 * j;
        if (c * 4 < H) {
            const float4 t = reinterpret_cast<const float4*>(x + row_off)[c];
            v[j][0] = t.x; v[j][1] = t.y; v[j][2] = t.z; v[j][3] = t.w;
        } else {
            v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f;
        }
    }
    ln_row_store<MAXV>(v, lane, H, row_off, gamma, beta, eps, y_hi, y_lo, y_f32, f16);
}

extern "C" int didi_layernorm(const float* x, const float* gamma, const float* beta, float eps, int M, int H,
                              uint16_t* y_hi, uint16_t* y_lo, float* y_f32, int plane_format, void* stream) {
    DIDI_CHECK_ARG(x && gamma && beta && (y_hi || y_f32) && M >= 0 && H > 0);
    DIDI_CHECK_ARG(plane_format == DIDI_FMT_BF16 || plane_format == DIDI_FMT_FP16);
    const int f16 = plane_format == DIDI_FMT_FP16;
    DIDI_CHECK_ARG(!(y_lo && !y_hi));
    if (H % 4 != 0 || H > 4096) return DIDI_ERR_UNSUPPORTED;  // 4096: the widest context encoder the reference names (T5 v1.1 xxl)
    if (M == 0) return DIDI_OK;
    const int grid = cdiv(M, 4);
    hipStream_t st = as_stream(stream);
    if (H <= 256) layernorm_kernel<1><<<grid, 256, 0, st>>>(x, gamma, beta, eps, M, H, y_hi, y_lo, y_f32, f16);
    else if (H <= 768) layernorm_kernel<3><<<grid, 256, 0, st>>>(x, gamma, beta, eps, M, H, y_hi, y_lo, y_f32, f16);
    else if (H <= 1280) layernorm_kernel<5><<<grid, 256, 0, st>>>(x, gamma, beta, eps, M, H, y_hi, y_lo, y_f32, f16);
    else if (H <= 2048) layernorm_kernel<8><<<grid, 256, 0, st>>>(x, gamma, beta, eps, M, H, y_hi, y_lo, y_f32, f16);
    else layernorm_kernel<16><<<grid, 256, 0, st>>>(x, gamma, beta, eps, M, H, y_hi, y_lo, y_f32, f16);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// Training forward of BertSelfOutput / BertOutput with hidden dropout on: s = residual + dropout(x) (didi_dropout_add's mask and
// arithmetic), s kept for the backward, LayerNorm(s) out -- one read of x and of the residual instead of a dropout pass that writes s
// and a LayerNorm pass that reads it back.
template <int MAXV>
__global__ void __launch_bounds__(256) dropout_add_layernorm_kernel(const float* x, const float* residual, float p, uint64_t seed, uint64_t offset,
                                                                    float* sum_out, const float* gamma, const float* beta, float eps, int M, int H,
                                                                    u16* y_hi, u16* y_lo, float* y_f32, int f16) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const size_t row_off = (size_t)m * H;
    const uint32_t thr = (uint32_t)fminf(p * 4294967296.0f, 4294967040.0f);
    const float inv_keep = 1.0f / (1.0f - p);
    float v[MAXV][4];
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane + 64 * j;
        if (c * 4 < H) {
            float4 t = reinterpret_cast<const float4*>(x + row_off)[c];
            const float4 r = reinterpret_cast<const float4*>(residual + row_off)[c];
            uint32_t w[4];
            philox4x32_10(offset + row_off / 4 + c, seed, w);
            t.x = w[0] < thr ? 0.f : t.x * inv_keep;
            t.y = w[1] < thr ? 0.f : t.y * inv_keep;
            t.z = w[2] < thr ? 0.f : t.z * inv_keep;
            t.w = w[3] < thr ? 0.f : t.w * inv_keep;
            t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
            reinterpret_cast<float4*>(sum_out + row_off)[c] = t;
            v[j][0] = t.x; v[j][1] = t.y; v[j][2] = t.z; v[j][3] = t.w;
        } else {
            v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f;
        }
    }
    ln_row_store<MAXV>(v, lane, H, row_off, gamma, beta, eps, y_hi, y_lo, y_f32, f16);
}

extern "C" int didi_dropout_add_layernorm(const float* x, const float* residual, float p, uint64_t seed, uint64_t offset, float* sum_out,
                                          const float* gamma, const float* beta, float eps, int M, int H, uint16_t* y_hi, uint16_t* y_lo,
                                          float* y_f32, int plane_format, void* stream) {
    DIDI_CHECK_ARG(x && residual && sum_out && gamma && beta && (y_hi || y_f32) && M >= 0 && H > 0 && p >= 0.f && p < 1.f);
    DIDI_CHECK_ARG(plane_format == DIDI_FMT_BF16 || plane_format == DIDI_FMT_FP16);
    const int f16 = plane_format == DIDI_FMT_FP16;
    DIDI_CHECK_ARG(!(y_lo && !y_hi));
    if (H % 4 != 0 || H > 1280) return DIDI_ERR_UNSUPPORTED;
    if (M == 0) return DIDI_OK;
    const int grid = cdiv(M, 4);
    hipStream_t st = as_stream(stream);
    if (H <= 256) dropout_add_layernorm_kernel<1><<<grid, 256, 0, st>>>(x, residual, p, seed, offset, sum_out, gamma, beta, eps, M, H, y_hi, y_lo, y_f32, f16);
    else if (H <= 768) dropout_add_layernorm_kernel<3><<<grid, 256, 0, st>>>(x, residual, p, seed, offset, sum_out, gamma, beta, eps, M, H, y_hi, y_lo, y_f32, f16);
    else dropout_add_layernorm_kernel<5><<<grid, 256, 0, st>>>(x, residual, p, seed, offset, sum_out, gamma, beta, eps, M, H, y_hi, y_lo, y_f32, f16);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ T5LayerNorm (RMS norm) and the gated-FFN product
template <int MAXV>
__global__ void __launch_bounds__(256) rmsnorm_kernel(const float* x, const float* gamma, float eps, int M, int H, u16* y_hi,
                                                      u16* y_lo, float* y_f32, int f16) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const size_t row_off = (size_t)m * H;
    float v[MAXV][4];
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane + 64 * j;
        v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f;
        if (c * 4 < H) {
            const float4 t = reinterpret_cast<const float4*>(x + row_off)[c];
            v[j][0] = t.x; v[j][1] = t.y; v[j][2] = t.z; v[j][3] = t.w;
            q += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane + 64 * j;
        if (c * 4 < H) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[c];
            const float y[4] = {v[j][0] * rstd * g.x, v[j][1] * rstd * g.y, v[j][2] * rstd * g.z, v[j][3] * rstd * g.w};
            const size_t o = row_off + (size_t)c * 4;
            if (y_f32) *reinterpret_cast<float4*>(y_f32 + o) = make_float4(y[0], y[1], y[2], y[3]);
            if (y_hi) {
                uint2 h, l;
                split4(y, h, l, f16);
                *reinterpret_cast<uint2*>(y_hi + o) = h;
                if (y_lo) *reinterpret_cast<uint2*>(y_lo + o) = l;
            }
        }
    }
}

extern "C" int didi_rmsnorm(const float* x, const float* gamma, float eps, int M, int H, uint16_t* y_hi, uint16_t* y_lo,
                            float* y_f32, int plane_format, void* stream) {
    DIDI_CHECK_ARG(x && gamma && (y_hi || y_f32) && M >= 0 && H > 0);
    DIDI_CHECK_ARG(plane_format == DIDI_FMT_BF16 || plane_format == DIDI_FMT_FP16);
    const int f16 = plane_format == DIDI_FMT_FP16;
    DIDI_CHECK_ARG(!(y_lo && !y_hi));
    if (H % 4 != 0 || H > 4096) return DIDI_ERR_UNSUPPORTED;
    if (M == 0) return DIDI_OK;
    const int grid = cdiv(M, 4);
    hipStream_t st = as_stream(stream);
    if (H <= 256) rmsnorm_kernel<1><<<grid, 256, 0, st>>>(x, gamma, eps, M, H, y_hi, y_lo, y_f32, f16);
    else if (H <= 768) rmsnorm_kernel<3><<<grid, 256, 0, st>>>(x, gamma, eps, M, H, y_hi, y_lo, y_f32, f16);
    else if (H <= 1280) rmsnorm_kernel<5><<<grid, 256, 0, st>>>(x, gamma, eps, M, H, y_hi, y_lo, y_f32, f16);
    else if (H <= 2048) rmsnorm_kernel<8><<<grid, 256, 0, st>>>(x, gamma, eps, M, H, y_hi, y_lo, y_f32, f16);
    else rmsnorm_kernel<16><<<grid, 256, 0, st>>>(x, gamma, eps, M, H, y_hi, y_lo, y_f32, f16);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

__global__ void __launch_bounds__(256) mul_split_kernel(const float* a, const float* b, size_t n4, u16* y_hi, u16* y_lo, int f16) {
    if (f16) f16_saturate_mode();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 u = reinterpret_cast<const float4*>(a)[i], w = reinterpret_cast<const float4*>(b)[i];
    const float y[4] = {u.x * w.x, u.y * w.y, u.z * w.z, u.w * w.w};
    uint2 h, l;
    split4(y, h, l, f16);
    reinterpret_cast<uint2*>(y_hi)[i] = h;
    if (y_lo) reinterpret_cast<uint2*>(y_lo)[i] = l;
}

extern "C" int didi_mul_split(const float* a, const float* b, size_t n, uint16_t* y_hi, uint16_t* y_lo, int plane_format,
                              void* stream) {
    DIDI_CHECK_ARG(a && b && y_hi && n % 4 == 0);
    DIDI_CHECK_ARG(plane_format == DIDI_FMT_BF16 || plane_format == DIDI_FMT_FP16);
    if (n == 0) return DIDI_OK;
    mul_split_kernel<<<(unsigned)cdiv((long)(n / 4), 256), 256, 0, as_stream(stream)>>>(a, b, n / 4, y_hi, y_lo, plane_format == DIDI_FMT_FP16);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ sampler step + decoder prologue
// bytes per row (H fp32 = 4H): DDPM reads x0 (4H) [+ eps 4H when injected], writes planes (4H)
// [+ x_t 4H when traced]; the reference makes 3 elementwise passes + gather + add + LN passes.
// MODE is a template parameter and every load of a row is issued before the first use: with the mode as a run-time branch per
// 16-byte chunk hipcc placed the x0 load, the Philox draws, the time / token-type / position loads of each chunk one after the
// other with s_waitcnt vmcnt(0) between them -- six memory latencies in a row per wave (28.6 us per call at 16384 x 768).
template <int MAXV, int MODE>
__global__ void __launch_bounds__(256) step_embed_ln_kernel(didi_step_args a) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= a.M) return;
    const int H = a.H;
    const size_t row_off = (size_t)m * H;
    const int s = m % a.S;
    const int t = a.time_idx ? a.time_idx[m / a.S] : a.t;
    constexpr bool RD_X = MODE == DIDI_STEP_EULER || MODE == DIDI_STEP_NONE, RD_X0 = MODE == DIDI_STEP_DDPM || MODE == DIDI_STEP_EULER;
    const bool inj = a.E != nullptr && MODE != DIDI_STEP_NONE, emb = a.h_hi != nullptr;
    const bool draw = !inj && (a.ce != 0.f) && MODE != DIDI_STEP_NONE;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 qx[MAXV], qx0[MAXV], qe[MAXV], qte[MAXV], qtk[MAXV], qps[MAXV];
    int cc[MAXV];  // chunks past the row read chunk 0 instead (never used): no branch between the loads
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        cc[j] = (lane + 64 * j) * 4 < H ? lane + 64 * j : 0;
        qx[j] = qx0[j] = qe[j] = qte[j] = qtk[j] = qps[j] = z4;
        if constexpr (RD_X) qx[j] = reinterpret_cast<const float4*>(a.X + row_off)[cc[j]];
        if constexpr (RD_X0) qx0[j] = reinterpret_cast<const float4*>(a.X0 + row_off)[cc[j]];
    }
    if (inj) {
#pragma unroll
        for (int j = 0; j < MAXV; ++j) qe[j] = reinterpret_cast<const float4*>(a.E + row_off)[cc[j]];
    }
    if (emb) {
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            qte[j] = reinterpret_cast<const float4*>(a.time_embeds + (size_t)t * H)[cc[j]];
            qtk[j] = reinterpret_cast<const float4*>(a.tok_type0)[cc[j]];
            qps[j] = reinterpret_cast<const float4*>(a.pos_emb + (size_t)s * H)[cc[j]];
        }
    }
    float v[MAXV][4];
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane + 64 * j;
        v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f;
        if (c * 4 < H) {
            float x[4] = {qx[j].x, qx[j].y, qx[j].z, qx[j].w};
            const float x0[4] = {qx0[j].x, qx0[j].y, qx0[j].z, qx0[j].w};
            float e[4] = {qe[j].x, qe[j].y, qe[j].z, qe[j].w};
            if (draw) philox_normal4(a.offset + (row_off / 4 + c), a.seed, e);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float r;
                if constexpr (MODE == DIDI_STEP_INIT) r = a.ce * e[k];
                else if constexpr (MODE == DIDI_STEP_DDPM) r = a.cx0 * x0[k] + a.ce * e[k];
                else if constexpr (MODE == DIDI_STEP_EULER) r = (x[k] + (x[k] - x0[k]) * a.ratio) + a.ce * e[k];
                else r = x[k];
                x[k] = r;
            }
            if (a.x_out) *reinterpret_cast<float4*>(a.x_out + row_off + (size_t)c * 4) = make_float4(x[0], x[1], x[2], x[3]);
            if (a.xin_out)
                *reinterpret_cast<float4*>(a.xin_out + row_off + (size_t)c * 4) =
                    make_float4(a.cin * x[0], a.cin * x[1], a.cin * x[2], a.cin * x[3]);
            // (x + time) + token_type + position: the reference's association order
            v[j][0] = ((a.cin * x[0] + qte[j].x) + qtk[j].x) + qps[j].x;
            v[j][1] = ((a.cin * x[1] + qte[j].y) + qtk[j].y) + qps[j].y;
            v[j][2] = ((a.cin * x[2] + qte[j].z) + qtk[j].z) + qps[j].z;
            v[j][3] = ((a.cin * x[3] + qte[j].w) + qtk[j].w) + qps[j].w;
        }
    }
    if (emb) ln_row_store<MAXV>(v, lane, H, row_off, a.ln_gamma, a.ln_beta, a.ln_eps, a.h_hi, a.h_lo, nullptr,
                               a.plane_format == DIDI_FMT_FP16, a.act_dither);
}

template <int MAXV>
static void launch_step(const didi_step_args* a, int grid, hipStream_t st) {
    switch (a->mode) {
        case DIDI_STEP_INIT: step_embed_ln_kernel<MAXV, DIDI_STEP_INIT><<<grid, 256, 0, st>>>(*a); break;
        case DIDI_STEP_DDPM: step_embed_ln_kernel<MAXV, DIDI_STEP_DDPM><<<grid, 256, 0, st>>>(*a); break;
        case DIDI_STEP_EULER: step_embed_ln_kernel<MAXV, DIDI_STEP_EULER><<<grid, 256, 0, st>>>(*a); break;
        default: step_embed_ln_kernel<MAXV, DIDI_STEP_NONE><<<grid, 256, 0, st>>>(*a); break;
    }
}

extern "C" int didi_step_embed_ln(const didi_step_args* a, void* stream) {
    DIDI_CHECK_ARG(a && a->M >= 0 && a->S > 0 && a->H > 0 && a->M % a->S == 0);
    DIDI_CHECK_ARG(a->mode >= DIDI_STEP_NONE && a->mode <= DIDI_STEP_EULER);
    if (a->mode == DIDI_STEP_EULER || a->mode == DIDI_STEP_NONE) DIDI_CHECK_ARG(a->X);
    if (a->mode == DIDI_STEP_DDPM || a->mode == DIDI_STEP_EULER) DIDI_CHECK_ARG(a->X0);
    DIDI_CHECK_ARG(a->h_hi || a->x_out || a->xin_out);
    DIDI_CHECK_ARG(a->plane_format == DIDI_FMT_BF16 || a->plane_format == DIDI_FMT_FP16);
    DIDI_CHECK_ARG(a->act_dither >= 0 && a->act_dither < 8192);
    if (a->h_hi) DIDI_CHECK_ARG(a->h_lo && a->time_embeds && a->pos_emb && a->tok_type0 && a->ln_gamma && a->ln_beta);
    if (a->H % 4 != 0 || a->H > 2048) return DIDI_ERR_UNSUPPORTED;
    if (a->M == 0) return DIDI_OK;
    const int grid = cdiv(a->M, 4);
    hipStream_t st = as_stream(stream);
    if (a->H <= 256) launch_step<1>(a, grid, st);
    else if (a->H <= 768) launch_step<3>(a, grid, st);
    else if (a->H <= 1280) launch_step<5>(a, grid, st);
    else launch_step<8>(a, grid, st);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ token-embedding gather (encoder input)
__global__ void __launch_bounds__(256) embed_gather_kernel(const int64_t* ids, const float* table, float scale,
                                                           const float* pos, int M, int S, int H, float* out) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* src = table + (size_t)ids[m] * H;
    const float* ps = pos ? pos + (size_t)(m % S) * H : nullptr;
    for (int c = lane; c * 4 < H; c += 64) {
        float4 v = reinterpret_cast<const float4*>(src)[c];
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        if (ps) {
            const float4 q = reinterpret_cast<const float4*>(ps)[c];
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        reinterpret_cast<float4*>(out + (size_t)m * H)[c] = v;
    }
}

extern "C" int didi_embed_gather(const int64_t* ids, const float* table, float scale, const float* pos, int M, int S, int H,
                                 float* out, void* stream) {
    DIDI_CHECK_ARG(ids && table && out && M >= 0 && S > 0 && H > 0 && H % 4 == 0);
    if (M == 0) return DIDI_OK;
    embed_gather_kernel<<<cdiv(M, 4), 256, 0, as_stream(stream)>>>(ids, table, scale, pos, M, S, H, out);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ eos truncation lengths
__global__ void __launch_bounds__(256) first_eos_kernel(const int64_t* ids, int B, int S, int64_t eos, int32_t* lengths) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    int len = S;
    for (int s = 0; s < S; ++s) {
        if (ids[(size_t)b * S + s] == eos) { len = s + 1; break; }
    }
    lengths[b] = len;
}

extern "C" int didi_first_eos(const int64_t* ids, int B, int S, int64_t eos_id, int32_t* lengths, void* stream) {
    DIDI_CHECK_ARG(ids && lengths && B >= 0 && S >= 0);
    if (B == 0) return DIDI_OK;
    first_eos_kernel<<<cdiv(B, 256), 256, 0, as_stream(stream)>>>(ids, B, S, eos_id, lengths);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ misc
extern "C" int didi_version(void) { return DIDI_ABI_VERSION; }

extern "C" const char* didi_error_string(int code) {
    switch (code) {
        case DIDI_OK: return "ok";
        case DIDI_ERR_BAD_ARG: return "bad argument (null pointer, size or alignment constraint)";
        case DIDI_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
        case DIDI_ERR_LAUNCH: return "kernel launch failed";
        case DIDI_ERR_WORKSPACE: return "workspace too small";
        case DIDI_ERR_FAULT: return "device-side fault word is set (didi_fault_string names the bits)";
        default: return "unknown error";
    }
}

extern "C" const char* didi_fault_string(uint32_t bits) {
    switch (bits & (DIDI_FAULT_LN_EXCHANGE | DIDI_FAULT_FP16_RANGE)) {
        case 0: return bits ? "unknown fault bits" : "no fault";
        case DIDI_FAULT_LN_EXCHANGE:
            return "LayerNorm row-statistics exchange timed out (didi_gemm_ln launch not co-resident): rows are NaN";
        case DIDI_FAULT_FP16_RANGE:
            return "a GEMM output exceeded the fp16 range and was saturated to +-65504: use precision bf16x3";
        default:
            return "LayerNorm row-statistics exchange timed out AND a GEMM output exceeded the fp16 range";
    }
}

// host-side helper; the one call that waits for the stream
extern "C" int didi_fault_read(const uint32_t* fault, uint32_t* bits_host, void* stream) {
    DIDI_CHECK_ARG(fault);
    uint32_t bits = 0;
    if (hipMemcpyAsync(&bits, fault, sizeof(bits), hipMemcpyDeviceToHost, as_stream(stream)) != hipSuccess) return DIDI_ERR_LAUNCH;
    if (hipStreamSynchronize(as_stream(stream)) != hipSuccess) return DIDI_ERR_LAUNCH;
    if (bits_host) *bits_host = bits;
    return bits ? DIDI_ERR_FAULT : DIDI_OK;
}
