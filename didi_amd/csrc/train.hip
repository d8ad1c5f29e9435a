// Training-step kernels (reference src/diffusion/model.py:200-243): the noising of the target
// embeddings, the masked loss terms and their gradient, LayerNorm backward, column sums (bias gradients) and the
// transposing split that turns dC / A / W into the operand planes of the dgrad and wgrad GEMMs (which are didi_gemm launches:
// dA = dC W  is  didi_gemm(A' = dC [M,N], W' = W^T [K,N]);  dW = dC^T A  is  didi_gemm(A' = dC^T [N,M], W' = A^T [K,M])).
// All HBM-bound, one pass each; layouts as elementwise.hip (a wave owns a row, lane l holds 16-byte chunks l + 64 j).
#include <algorithm>

#include "common.hpp"
#include "philox.hpp"

// ------------------------------------------------------------------ noising: x_0 = emb + std_0 n0 ; x_t = (x_0 + s_b n1) c_b
// reference: get_x0 (src/diffusion/utils.py:24-27) and get_diffusion_variables (:34-40) with a per-dialog t:
// s_b = sigma_{t_b} * noise_factor^2, c_b = 1 / sqrt(sigma_{t_b}^2 + 1), both computed on the host from t.
// bytes per element: emb 4 + (n0, n1 4 + 4 when injected) in, x_0 4 + x_t 4 out.
__global__ void __launch_bounds__(256) train_noise_kernel(const float* emb, const float* n0, const float* n1, float std0,
                                                          const float* s_b, const float* c_b, int S, int H, size_t n4,
                                                          uint64_t seed, uint64_t offset, float* x0, float* xt) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int b = (int)(i / ((size_t)S * H / 4));
    const float4 e = reinterpret_cast<const float4*>(emb)[i];
    float a[4], z[4];
    if (n0) { const float4 q = reinterpret_cast<const float4*>(n0)[i]; a[0] = q.x; a[1] = q.y; a[2] = q.z; a[3] = q.w; }
    else philox_normal4(offset + i, seed, a);
    if (n1) { const float4 q = reinterpret_cast<const float4*>(n1)[i]; z[0] = q.x; z[1] = q.y; z[2] = q.z; z[3] = q.w; }
    else philox_normal4(offset + n4 + i, seed, z);
    const float s = s_b[b], c = c_b[b];
    const float x[4] = {e.x + std0 * a[0], e.y + std0 * a[1], e.z + std0 * a[2], e.w + std0 * a[3]};
    reinterpret_cast<float4*>(x0)[i] = make_float4(x[0], x[1], x[2], x[3]);
    reinterpret_cast<float4*>(xt)[i] = make_float4((x[0] + s * z[0]) * c, (x[1] + s * z[1]) * c, (x[2] + s * z[2]) * c, (x[3] + s * z[3]) * c);
}

extern "C" int didi_train_noise(const float* emb, const float* n0, const float* n1, float std0, const float* s_b, const float* c_b,
                                int B, int S, int H, uint64_t seed, uint64_t offset, float* x0, float* xt, void* stream) {
    DIDI_CHECK_ARG(emb && s_b && c_b && x0 && xt && B >= 0 && S > 0 && H > 0 && H % 4 == 0);
    const size_t n4 = (size_t)B * S * H / 4;
    if (n4 == 0) return DIDI_OK;
    train_noise_kernel<<<(unsigned)cdiv((long)n4, 256), 256, 0, as_stream(stream)>>>(emb, n0, n1, std0, s_b, c_b, S, H, n4, seed, offset, x0, xt);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ masked loss terms (model.py:222-231)
// One 256-thread workgroup per dialog: sums over its S*H elements of  m (x0hat - emb)^2,  m (x0hat - x0)^2,  m x0^2
// (m = the target's non-pad mask of the row).  out[b] = {sum_emb, sum_x0, sum_sq}; the means, the t == 1 mix (a [B, B]
// broadcast in the reference) and the sigma_T scale are scalar arithmetic on B values (didi_amd/training.py).
__global__ void __launch_bounds__(256) train_loss_terms_kernel(const float* x0hat, const float* emb, const float* x0,
                                                               const int64_t* mask, int S, int H, float* out) {
    __shared__ float red[3][4];
    const int b = blockIdx.x, h4 = H / 4;
    const size_t base = (size_t)b * S * H;
    float a = 0.f, c = 0.f, q = 0.f;
    for (int i = threadIdx.x; i < S * h4; i += 256) {
        const int s = i / h4;
        if (mask[(size_t)b * S + s] == 0) continue;
        const float4 y = reinterpret_cast<const float4*>(x0hat + base)[i];
        const float4 e = reinterpret_cast<const float4*>(emb + base)[i];
        const float4 x = reinterpret_cast<const float4*>(x0 + base)[i];
        a += (y.x - e.x) * (y.x - e.x) + (y.y - e.y) * (y.y - e.y) + (y.z - e.z) * (y.z - e.z) + (y.w - e.w) * (y.w - e.w);
        c += (y.x - x.x) * (y.x - x.x) + (y.y - x.y) * (y.y - x.y) + (y.z - x.z) * (y.z - x.z) + (y.w - x.w) * (y.w - x.w);
        q += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    }
    a = wave_sum(a); c = wave_sum(c); q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = c; red[2][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x < 3) out[b * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

extern "C" int didi_train_loss_terms(const float* x0hat, const float* emb, const float* x0, const int64_t* mask, int B, int S,
                                     int H, float* out, void* stream) {
    DIDI_CHECK_ARG(x0hat && emb && x0 && mask && out && B >= 0 && S > 0 && H > 0 && H % 4 == 0);
    if (B == 0) return DIDI_OK;
    train_loss_terms_kernel<<<B, 256, 0, as_stream(stream)>>>(x0hat, emb, x0, mask, S, H, out);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// d mse / d x0hat = m * (w_emb (x0hat - emb) + w_x0 (x0hat - x0)),  w_emb = 2 p / (B S H),  w_x0 = 2 (1 - p) / (B S H)
__global__ void __launch_bounds__(256) train_mse_backward_kernel(const float* x0hat, const float* emb, const float* x0,
                                                                 const int64_t* mask, float w_emb, float w_x0, int H, size_t n4, float* dx) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const size_t row = i / (H / 4);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mask[row] != 0) {
        const float4 y = reinterpret_cast<const float4*>(x0hat)[i], e = reinterpret_cast<const float4*>(emb)[i], x = reinterpret_cast<const float4*>(x0)[i];
        g = make_float4(w_emb * (y.x - e.x) + w_x0 * (y.x - x.x), w_emb * (y.y - e.y) + w_x0 * (y.y - x.y),
                        w_emb * (y.z - e.z) + w_x0 * (y.z - x.z), w_emb * (y.w - e.w) + w_x0 * (y.w - x.w));
    }
    reinterpret_cast<float4*>(dx)[i] = g;
}

extern "C" int didi_train_mse_backward(const float* x0hat, const float* emb, const float* x0, const int64_t* mask, float w_emb,
                                       float w_x0, int M, int H, float* dx, void* stream) {
    DIDI_CHECK_ARG(x0hat && emb && x0 && mask && dx && M >= 0 && H > 0 && H % 4 == 0);
    const size_t n4 = (size_t)M * H / 4;
    if (n4 == 0) return DIDI_OK;
    train_mse_backward_kernel<<<(unsigned)cdiv((long)n4, 256), 256, 0, as_stream(stream)>>>(x0hat, emb, x0, mask, w_emb, w_x0, H, n4, dx);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ LayerNorm backward
// y = (x - mean) rstd gamma + beta over the last dim.  With xh = (x - mean) rstd and g = dy gamma:
//   dx = rstd (g - mean_H(g) - xh mean_H(g xh)),  dgamma = sum_rows dy xh,  dbeta = sum_rows dy.
// One wave per row (statistics recomputed from x: two-pass in registers, as the forward); every 256-thread block
// accumulates its rows' dgamma / dbeta contributions in registers across a grid-stride loop and writes ONE partial row
// [2][H] per block; a second kernel sums the partials in a fixed order: deterministic, no atomics.
template <int MAXV>
__global__ void __launch_bounds__(256) layernorm_backward_kernel(const float* dy, const float* x, const float* gamma, float eps, int M,
                                                                 int H, float* dx, float* partial) {
    __shared__ float pg[4][MAXV * 256], pb[4][MAXV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ag[MAXV][4], ab[MAXV][4];
#pragma unroll
    for (int j = 0; j < MAXV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) ag[j][e] = ab[j][e] = 0.f;
    for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
        const size_t ro = (size_t)m * H;
        float v[MAXV][4], d[MAXV][4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int c = lane + 64 * j;
            v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f;
            d[j][0] = d[j][1] = d[j][2] = d[j][3] = 0.f;
            if (c * 4 < H) {
                const float4 t = reinterpret_cast<const float4*>(x + ro)[c], u = reinterpret_cast<const float4*>(dy + ro)[c];
                v[j][0] = t.x; v[j][1] = t.y; v[j][2] = t.z; v[j][3] = t.w;
                d[j][0] = u.x; d[j][1] = u.y; d[j][2] = u.z; d[j][3] = u.w;
                s += (t.x + t.y) + (t.z + t.w);
            }
        }
        const float mean = wave_sum(s) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < MAXV; ++j)
            if ((lane + 64 * j) * 4 < H) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float t = v[j][e] - mean; q += t * t; }
            }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int c = lane + 64 * j;
            if (c * 4 < H) {
                const float4 gm = reinterpret_cast<const float4*>(gamma)[c];
                const float gg[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (v[j][e] - mean) * rstd;
                    ag[j][e] += d[j][e] * xh;
                    ab[j][e] += d[j][e];
                    v[j][e] = xh;            // keep xh
                    d[j][e] = d[j][e] * gg[e];  // g
                    sg += d[j][e];
                    sgx += d[j][e] * xh;
                }
            }
        }
        const float mg = wave_sum(sg) / (float)H, mgx = wave_sum(sgx) / (float)H;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int c = lane + 64 * j;
            if (c * 4 < H)
                reinterpret_cast<float4*>(dx + ro)[c] = make_float4(rstd * (d[j][0] - mg - v[j][0] * mgx), rstd * (d[j][1] - mg - v[j][1] * mgx),
                                                                    rstd * (d[j][2] - mg - v[j][2] * mgx), rstd * (d[j][3] - mg - v[j][3] * mgx));
        }
    }
    // the block's four waves -> one partial row
#pragma unroll
    for (int j = 0; j < MAXV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pg[wave][(lane + 64 * j) * 4 + e] = ag[j][e];
            pb[wave][(lane + 64 * j) * 4 + e] = ab[j][e];
        }
    __syncthreads();
    for (int n = threadIdx.x; n < H; n += 256) {
        partial[((size_t)blockIdx.x * 2 + 0) * H + n] = (pg[0][n] + pg[1][n]) + (pg[2][n] + pg[3][n]);
        partial[((size_t)blockIdx.x * 2 + 1) * H + n] = (pb[0][n] + pb[1][n]) + (pb[2][n] + pb[3][n]);
    }
}

__global__ void __launch_bounds__(256) partial_rows_sum_kernel(const float* partial, int nblk, int stride, int W, float* out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= W) return;
    float a = 0.f;
    for (int b = 0; b < nblk; ++b) a += partial[(size_t)b * stride + n];
    out[n] = a;
}

static int ln_bwd_blocks(int M) { return std::min(1024, cdiv(M, 4)); }

extern "C" size_t didi_layernorm_backward_bytes(int M, int H) { return M > 0 && H > 0 ? (size_t)ln_bwd_blocks(M) * 2 * H * sizeof(float) : 0; }

extern "C" int didi_layernorm_backward(const float* dy, const float* x, const float* gamma, float eps, int M, int H, float* dx,
                                       float* dgamma, float* dbeta, void* workspace, void* stream) {
    DIDI_CHECK_ARG(dy && x && gamma && dx && dgamma && dbeta && workspace && M >= 0 && H > 0);
    if (H % 4 != 0 || H > 1280) return DIDI_ERR_UNSUPPORTED;
    if (M == 0) return DIDI_OK;
    const int nblk = ln_bwd_blocks(M);
    float* partial = reinterpret_cast<float*>(workspace);
    hipStream_t st = as_stream(stream);
    if (H <= 256) layernorm_backward_kernel<1><<<nblk, 256, 0, st>>>(dy, x, gamma, eps, M, H, dx, partial);
    else if (H <= 768) layernorm_backward_kernel<3><<<nblk, 256, 0, st>>>(dy, x, gamma, eps, M, H, dx, partial);
    else layernorm_backward_kernel<5><<<nblk, 256, 0, st>>>(dy, x, gamma, eps, M, H, dx, partial);
    DIDI_LAUNCH_CHECK();
    partial_rows_sum_kernel<<<cdiv(H, 256), 256, 0, st>>>(partial, nblk, 2 * H, H, dgamma);
    partial_rows_sum_kernel<<<cdiv(H, 256), 256, 0, st>>>(partial + H, nblk, 2 * H, H, dbeta);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ column sums (bias gradient): out[n] = sum_m x[m, n]
__global__ void __launch_bounds__(256) colsum_partial_kernel(const float* x, int M, int N, int rows_per_block, float* partial) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
    float a = 0.f;
    for (int m = m0; m < m1; ++m) a += x[(size_t)m * N + n];
    partial[(size_t)blockIdx.y * N + n] = a;
}

static int colsum_blocks(int M) { return std::min(256, std::max(1, M / 16)); }

extern "C" size_t didi_colsum_bytes(int M, int N) { return M > 0 && N > 0 ? (size_t)colsum_blocks(M) * N * sizeof(float) : 0; }

extern "C" int didi_colsum(const float* x, int M, int N, float* out, void* workspace, void* stream) {
    DIDI_CHECK_ARG(x && out && workspace && M > 0 && N > 0);
    const int nb = colsum_blocks(M), rpb = cdiv(M, nb);
    hipStream_t st = as_stream(stream);
    colsum_partial_kernel<<<dim3(cdiv(N, 256), nb), 256, 0, st>>>(x, M, N, rpb, reinterpret_cast<float*>(workspace));
    DIDI_LAUNCH_CHECK();
    partial_rows_sum_kernel<<<cdiv(N, 256), 256, 0, st>>>(reinterpret_cast<const float*>(workspace), nb, N, N, out);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ transposing split: fp32 [R, C] -> planes [C, R]
// 64 x 64 tiles through LDS (padded rows: conflict-free both ways); reads and writes are row-contiguous.
__global__ void __launch_bounds__(256) transpose_split_kernel(const float* src, int R, int C, int ldo, u16* hi, u16* lo, int f16) {
    __shared__ float tile[64][65];
    if (f16) f16_saturate_mode();
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;  // output row c, output column r
        if (c < C && r < ldo) {
            u16 h, l;
            split2(tile[tx][i], h, l, f16);  // columns R..ldo-1 come out as zeros (tile rows past R are zero)
            hi[(size_t)c * ldo + r] = h;
            if (lo) lo[(size_t)c * ldo + r] = l;
        }
    }
}

extern "C" int didi_transpose_split(const float* src, int R, int C, int ldo, uint16_t* hi, uint16_t* lo, int plane_format, void* stream) {
    DIDI_CHECK_ARG(src && hi && R > 0 && C > 0 && ldo >= R);
    DIDI_CHECK_ARG(plane_format == DIDI_FMT_BF16 || plane_format == DIDI_FMT_FP16);
    transpose_split_kernel<<<dim3(cdiv(C, 64), cdiv(ldo, 64)), 256, 0, as_stream(stream)>>>(src, R, C, ldo, hi, lo, plane_format == DIDI_FMT_FP16);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}
