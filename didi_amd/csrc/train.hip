// Training-step kernels (reference src/diffusion/model.py:200-243): the noising of the target
// embeddings, the masked loss terms and their gradient, LayerNorm backward, column sums (bias gradients) and the
// transposing split that turns dC / A / W into the operand planes of the dgrad and wgrad GEMMs (which are didi_gemm launches:
// dA = dC W  is  didi_gemm(A' = dC [M,N], W' = W^T [K,N]);  dW = dC^T A  is  didi_gemm(A' = dC^T [N,M], W' = A^T [K,M])).
// All HBM-bound, one pass each; layouts as elementwise.hip (a wave owns a row, lane l holds 16-byte chunks l + 64 j).
#include <algorithm>
#include <mutex>

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

// out[n] = sum_b partial[b, n] for n < W (and out2[n - W] for W <= n < 2W when out2 is given): 64 columns per workgroup, the
// rows dealt over its four waves with four independent accumulators each (a serial 1024-term chain per column cost 140 us).
__global__ void __launch_bounds__(256) partial_rows_sum_kernel(const float* partial, int nblk, int stride, int W, float* out, float* out2) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane, total = out2 ? 2 * W : W;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (n < total) {
        int b = wave;
        for (; b + 60 < nblk; b += 64) {  // 16 independent loads in flight: the kernel is a chain of memory latencies (12 workgroups at N = 768)
            float t_[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t_[u] = partial[(size_t)(b + 4 * u) * stride + n];
#pragma unroll
            for (int u = 0; u < 16; u += 4) { a0 += t_[u]; a1 += t_[u + 1]; a2 += t_[u + 2]; a3 += t_[u + 3]; }
        }
        for (; b + 12 < nblk; b += 16) {
            a0 += partial[(size_t)b * stride + n];
            a1 += partial[(size_t)(b + 4) * stride + n];
            a2 += partial[(size_t)(b + 8) * stride + n];
            a3 += partial[(size_t)(b + 12) * stride + n];
        }
        for (; b < nblk; b += 4) a0 += partial[(size_t)b * stride + n];
    }
    red[wave][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wave == 0 && n < total) {
        const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (n < W) out[n] = v;
        else out2[n - W] = v;
    }
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
    partial_rows_sum_kernel<<<cdiv(2 * H, 64), 256, 0, st>>>(partial, nblk, 2 * H, H, dgamma, dbeta);
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
    partial_rows_sum_kernel<<<cdiv(N, 64), 256, 0, st>>>(reinterpret_cast<const float*>(workspace), nb, N, N, out, nullptr);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ transposing split: fp32 [R, C] -> planes [C, R]
// 64 x 64 tiles through LDS (padded rows: conflict-free both ways); reads and writes are row-contiguous.
// `partial` (nullable): [gridDim.y][C] column sums of each 64-row block -- the bias gradient as a by-product of the pass that
// prepares dC^T for the weight-gradient GEMM (a separate column-sum pass re-read the whole tensor: 3 ms per training step).
__global__ void __launch_bounds__(256) transpose_split_kernel(const float* src, int R, int C, int ldo, u16* hi, u16* lo, int f16,
                                                              float* partial) {
    __shared__ float tile[64][65];
    __shared__ float red[4][64];
    if (f16) f16_saturate_mode();
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    float cs = 0.f;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        const float val = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
        tile[i][tx] = val;
        cs += val;
    }
    if (partial) red[ty][tx] = cs;
    __syncthreads();
    if (partial && ty == 0 && c0 + tx < C) partial[(size_t)blockIdx.y * C + c0 + tx] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
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

// d/dx of the exact-erf GELU: Phi(x) + x phi(x)
__device__ __forceinline__ float gelu_grad(float v) {
    return 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v);
}
// The same pass with 16-byte loads and 8-byte stores (C % 4 == 0, which every caller's shapes satisfy; the kernel above stays as the
// general form): thread t loads columns 4 (t & 15) .. + 3 of rows (t >> 4) + 16 j, and stores rows-of-the-source 4 (t & 15) .. + 3 as
// four consecutive columns of output row (t >> 4) + 16 j.  2.9 -> ~4 TB/s of the pass's 8 bytes per element.
// drop_p > 0: src is a gradient arriving at a hidden-dropout site; its mask (dropout_add_kernel's: word e & 3 of
// Philox4x32-10(offset + e / 4, seed) for element e = r C + c) is applied as the values are read, so the backward needs no pass of its own
__global__ void __launch_bounds__(256) transpose_split_kernel_v4(const float* src, int R, int C, int ldo, u16* hi, u16* lo, int f16,
                                                                 float* partial, u16* plain_hi = nullptr, u16* plain_lo = nullptr,
                                                                 float drop_p = 0.f, uint64_t drop_seed = 0, uint64_t drop_offset = 0,
                                                                 const float* gelu_u = nullptr) {
    __shared__ float tile[64][65];
    __shared__ float red[16][64];
    if (f16) f16_saturate_mode();
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tq = threadIdx.x & 15, tr = threadIdx.x >> 4;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = tr + 16 * j, r = r0 + i, c = c0 + 4 * tq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R && c < C) {
            v = *reinterpret_cast<const float4*>(src + (size_t)r * C + c);  // C % 4 == 0: c + 3 < C too
            if (gelu_u) {  // src is the gradient behind a GELU whose pre-activation is gelu_u: gelu_backward_kernel's factor, on the read
                const float4 u = *reinterpret_cast<const float4*>(gelu_u + (size_t)r * C + c);
                v.x *= gelu_grad(u.x); v.y *= gelu_grad(u.y); v.z *= gelu_grad(u.z); v.w *= gelu_grad(u.w);
            }
            if (drop_p > 0.f) {
                const uint32_t thr = (uint32_t)fminf(drop_p * 4294967296.0f, 4294967040.0f);
                const float inv_keep = 1.0f / (1.0f - drop_p);
                uint32_t w[4];
                philox4x32_10(drop_offset + ((size_t)r * C + c) / 4, drop_seed, w);
                v.x = w[0] < thr ? 0.f : v.x * inv_keep;
                v.y = w[1] < thr ? 0.f : v.y * inv_keep;
                v.z = w[2] < thr ? 0.f : v.z * inv_keep;
                v.w = w[3] < thr ? 0.f : v.w * inv_keep;
            }
            if (plain_hi) {  // the un-transposed planes from the same read (the dgrad operand of linear_backward)
                const float pv[4] = {v.x, v.y, v.z, v.w};
                uint2 ph, pl;
                split4(pv, ph, pl, f16);
                *reinterpret_cast<uint2*>(plain_hi + (size_t)r * C + c) = ph;
                if (plain_lo) *reinterpret_cast<uint2*>(plain_lo + (size_t)r * C + c) = pl;
            }
        }
        tile[i][4 * tq] = v.x; tile[i][4 * tq + 1] = v.y; tile[i][4 * tq + 2] = v.z; tile[i][4 * tq + 3] = v.w;
        cs[0] += v.x; cs[1] += v.y; cs[2] += v.z; cs[3] += v.w;
    }
    if (partial) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red[tr][4 * tq + k] = cs[k];
    }
    __syncthreads();
    if (partial && threadIdx.x < 64 && c0 + threadIdx.x < C) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][threadIdx.x];
        partial[(size_t)blockIdx.y * C + c0 + threadIdx.x] = s;
    }
    if (!hi) return;  // didi_split_both without transposed planes
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + tr + 16 * j, r = r0 + 4 * tq;  // output row c, output columns r .. r + 3
        if (c < C && r < ldo) {                           // ldo % 8 == 0: r + 3 < ldo too
            const float v[4] = {tile[4 * tq][tr + 16 * j], tile[4 * tq + 1][tr + 16 * j], tile[4 * tq + 2][tr + 16 * j], tile[4 * tq + 3][tr + 16 * j]};
            uint2 h, l;
            split4(v, h, l, f16);  // columns R .. ldo - 1 come out as zeros (tile rows past R are zero)
            *reinterpret_cast<uint2*>(hi + (size_t)c * ldo + r) = h;
            if (lo) *reinterpret_cast<uint2*>(lo + (size_t)c * ldo + r) = l;
        }
    }
}
static bool transpose_v4_ok(const float* src, int C, int ldo, const uint16_t* hi, const uint16_t* lo) {
    return C % 4 == 0 && ldo % 4 == 0 && reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(hi) % 8 == 0 &&
           reinterpret_cast<uintptr_t>(lo) % 8 == 0;
}

extern "C" int didi_transpose_split(const float* src, int R, int C, int ldo, uint16_t* hi, uint16_t* lo, int plane_format, void* stream) {
    DIDI_CHECK_ARG(src && hi && R > 0 && C > 0 && ldo >= R);
    DIDI_CHECK_ARG(plane_format == DIDI_FMT_BF16 || plane_format == DIDI_FMT_FP16);
    if (transpose_v4_ok(src, C, ldo, hi, lo))
        transpose_split_kernel_v4<<<dim3(cdiv(C, 64), cdiv(ldo, 64)), 256, 0, as_stream(stream)>>>(src, R, C, ldo, hi, lo, plane_format == DIDI_FMT_FP16, nullptr);
    else
        transpose_split_kernel<<<dim3(cdiv(C, 64), cdiv(ldo, 64)), 256, 0, as_stream(stream)>>>(src, R, C, ldo, hi, lo, plane_format == DIDI_FMT_FP16, nullptr);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

extern "C" size_t didi_transpose_split_colsum_bytes(int R, int C) { return R > 0 && C > 0 ? (size_t)cdiv((R + 7) / 8 * 8, 64) * C * sizeof(float) : 0; }

extern "C" int didi_transpose_split_colsum(const float* src, int R, int C, int ldo, uint16_t* hi, uint16_t* lo, int plane_format, float* colsum,
                                           void* workspace, void* stream) {
    DIDI_CHECK_ARG(src && hi && colsum && workspace && R > 0 && C > 0 && ldo >= R && ldo == (R + 7) / 8 * 8);
    DIDI_CHECK_ARG(plane_format == DIDI_FMT_BF16 || plane_format == DIDI_FMT_FP16);
    const int nb = cdiv(ldo, 64);
    hipStream_t st = as_stream(stream);
    if (transpose_v4_ok(src, C, ldo, hi, lo))
        transpose_split_kernel_v4<<<dim3(cdiv(C, 64), nb), 256, 0, st>>>(src, R, C, ldo, hi, lo, plane_format == DIDI_FMT_FP16, reinterpret_cast<float*>(workspace));
    else
        transpose_split_kernel<<<dim3(cdiv(C, 64), nb), 256, 0, st>>>(src, R, C, ldo, hi, lo, plane_format == DIDI_FMT_FP16, reinterpret_cast<float*>(workspace));
    DIDI_LAUNCH_CHECK();
    partial_rows_sum_kernel<<<cdiv(C, 64), 256, 0, st>>>(reinterpret_cast<const float*>(workspace), nb, C, C, colsum, nullptr);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// transposed planes + un-transposed planes (+ column sums when colsum != NULL) of one fp32 matrix in ONE pass: what linear_backward
// needs of dC (dgrad reads dC, wgrad dC^T, the bias gradient its column sums).  C % 4 == 0.
static int split_both_impl(const float* src, int R, int C, int ldo, uint16_t* t_hi, uint16_t* t_lo, uint16_t* p_hi, uint16_t* p_lo,
                           int plane_format, float* colsum, void* workspace, const float* gelu_u, float drop_p, uint64_t drop_seed, uint64_t drop_offset,
                           void* stream) {
    DIDI_CHECK_ARG(src && p_hi && R > 0 && C > 0 && ldo >= R && ldo == (R + 7) / 8 * 8 && (!colsum || workspace) && (t_hi || !t_lo));
    DIDI_CHECK_ARG(plane_format == DIDI_FMT_BF16 || plane_format == DIDI_FMT_FP16);
    if (!transpose_v4_ok(src, C, ldo, t_hi, t_lo) || reinterpret_cast<uintptr_t>(p_hi) % 8 != 0 || reinterpret_cast<uintptr_t>(p_lo) % 8 != 0)
        return DIDI_ERR_UNSUPPORTED;
    const int nb = cdiv(ldo, 64);
    hipStream_t st = as_stream(stream);
    transpose_split_kernel_v4<<<dim3(cdiv(C, 64), nb), 256, 0, st>>>(src, R, C, ldo, t_hi, t_lo, plane_format == DIDI_FMT_FP16,
                                                                      colsum ? reinterpret_cast<float*>(workspace) : nullptr, p_hi, p_lo, drop_p, drop_seed,
                                                                      drop_offset, gelu_u);
    DIDI_LAUNCH_CHECK();
    if (colsum) {
        partial_rows_sum_kernel<<<cdiv(C, 64), 256, 0, st>>>(reinterpret_cast<const float*>(workspace), nb, C, C, colsum, nullptr);
        DIDI_LAUNCH_CHECK();
    }
    return DIDI_OK;
}

extern "C" int didi_split_both(const float* src, int R, int C, int ldo, uint16_t* t_hi, uint16_t* t_lo, uint16_t* p_hi, uint16_t* p_lo,
                               int plane_format, float* colsum, void* workspace, void* stream) {
    return split_both_impl(src, R, C, ldo, t_hi, t_lo, p_hi, p_lo, plane_format, colsum, workspace, nullptr, 0.f, 0, 0, stream);
}
extern "C" int didi_split_both_ex(const float* src, int R, int C, int ldo, uint16_t* t_hi, uint16_t* t_lo, uint16_t* p_hi, uint16_t* p_lo,
                                  int plane_format, float* colsum, void* workspace, const float* gelu_u, float drop_p, uint64_t seed, uint64_t offset,
                                  void* stream) {
    DIDI_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && reinterpret_cast<uintptr_t>(gelu_u) % 16 == 0);
    return split_both_impl(src, R, C, ldo, t_hi, t_lo, p_hi, p_lo, plane_format, colsum, workspace, gelu_u, drop_p, seed, offset, stream);
}

// ------------------------------------------------------------------ elementwise backward pieces
// GELU (exact erf form, hidden_act="gelu"): du = df * (Phi(u) + u phi(u));  tanh: dz = dy * (1 - y^2) from the OUTPUT y.
__global__ void __launch_bounds__(256) gelu_backward_kernel(const float* df, const float* u, size_t n4, float* du) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 g = reinterpret_cast<const float4*>(df)[i], x = reinterpret_cast<const float4*>(u)[i];
    reinterpret_cast<float4*>(du)[i] = make_float4(g.x * gelu_grad(x.x), g.y * gelu_grad(x.y), g.z * gelu_grad(x.z), g.w * gelu_grad(x.w));
}
__global__ void __launch_bounds__(256) tanh_backward_kernel(const float* dy, const float* y, size_t n4, float* dz) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 g = reinterpret_cast<const float4*>(dy)[i], t = reinterpret_cast<const float4*>(y)[i];
    reinterpret_cast<float4*>(dz)[i] = make_float4(g.x * (1.f - t.x * t.x), g.y * (1.f - t.y * t.y), g.z * (1.f - t.z * t.z), g.w * (1.f - t.w * t.w));
}
// y = act(x) as fp32 + split planes (the training forward keeps the pre-activation, so the activation is its own pass there)
template <bool F16>
__global__ void __launch_bounds__(256) act_forward_kernel(int act, const float* x, size_t n4, float* y, uint16_t* hi, uint16_t* lo) {
    if (F16) f16_saturate_mode();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 u = reinterpret_cast<const float4*>(x)[i];
    auto f = [act](float v) { return act == DIDI_ACT_GELU ? 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)) : tanhf(v); };
    const float r[4] = {f(u.x), f(u.y), f(u.z), f(u.w)};
    if (y) reinterpret_cast<float4*>(y)[i] = make_float4(r[0], r[1], r[2], r[3]);
    if (hi) {
        uint2 h, l;
        split4(r, h, l, F16);
        reinterpret_cast<uint2*>(hi)[i] = h;
        if (lo) reinterpret_cast<uint2*>(lo)[i] = l;
    }
}
extern "C" int didi_act_forward(int act, const float* x, size_t n, float* y, void* hi, void* lo, int plane_format, void* stream) {
    DIDI_CHECK_ARG(x && (y || hi) && n % 4 == 0 && (act == DIDI_ACT_GELU || act == DIDI_ACT_TANH));
    DIDI_CHECK_ARG(plane_format == DIDI_FMT_BF16 || plane_format == DIDI_FMT_FP16);
    if (n == 0) return DIDI_OK;
    const unsigned grid = (unsigned)cdiv((long)(n / 4), 256);
    if (plane_format == DIDI_FMT_FP16)
        act_forward_kernel<true><<<grid, 256, 0, as_stream(stream)>>>(act, x, n / 4, y, (uint16_t*)hi, (uint16_t*)lo);
    else
        act_forward_kernel<false><<<grid, 256, 0, as_stream(stream)>>>(act, x, n / 4, y, (uint16_t*)hi, (uint16_t*)lo);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

extern "C" int didi_act_backward(int act, const float* dy, const float* saved, size_t n, float* dx, void* stream) {
    DIDI_CHECK_ARG(dy && saved && dx && n % 4 == 0 && (act == DIDI_ACT_GELU || act == DIDI_ACT_TANH));
    if (n == 0) return DIDI_OK;
    const unsigned grid = (unsigned)cdiv((long)(n / 4), 256);
    if (act == DIDI_ACT_GELU) gelu_backward_kernel<<<grid, 256, 0, as_stream(stream)>>>(dy, saved, n / 4, dx);
    else tanh_backward_kernel<<<grid, 256, 0, as_stream(stream)>>>(dy, saved, n / 4, dx);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// y[i] = a x[i] + b y[i]: gradient accumulation where two paths meet (residual connections, x_0's several consumers)
__global__ void __launch_bounds__(256) axpby_kernel(float a, const float* x, float b, float* y, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 u = reinterpret_cast<const float4*>(x)[i];
    float4 v = reinterpret_cast<float4*>(y)[i];
    v = make_float4(a * u.x + b * v.x, a * u.y + b * v.y, a * u.z + b * v.z, a * u.w + b * v.w);
    reinterpret_cast<float4*>(y)[i] = v;
}
extern "C" int didi_axpby(float a, const float* x, float b, float* y, size_t n, void* stream) {
    DIDI_CHECK_ARG(x && y && n % 4 == 0);
    if (n == 0) return DIDI_OK;
    axpby_kernel<<<(unsigned)cdiv((long)(n / 4), 256), 256, 0, as_stream(stream)>>>(a, x, b, y, n / 4);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ cross-entropy backward on materialised logits
// calculate_batch_ce (src/metrics.py:91-93) = mean over non-pad tokens of -log softmax(logits)[target]:
// dlogits[m, :] = w_m (softmax(logits[m, :]) - onehot(target_m)), w_m = scale for kept tokens, 0 for padding.  In place.
// One 256-thread workgroup per row; V up to the vocabulary (50 000: 196 values per thread in three passes over L2-resident data).
__global__ void __launch_bounds__(256) ce_backward_kernel(float* logits, const int64_t* targets, const int64_t* mask, int V, float scale,
                                                          float* nll) {
    __shared__ float red[4];
    float* row = logits + (size_t)blockIdx.x * V;
    const bool keep = mask[blockIdx.x] != 0;
    float mx = -INFINITY;
    if (keep)
        for (int n = threadIdx.x; n < V; n += 256) mx = fmaxf(mx, row[n]);
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    if (keep)
        for (int n = threadIdx.x; n < V; n += 256) s += __expf(row[n] - mx);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = keep ? scale / s : 0.f;
    const int64_t tg = targets[blockIdx.x];
    // a target outside [0, V) (vocabulary mismatch, an ignore_index such as -100 on a row the mask keeps): the row gets +inf as its
    // term -- the step's loss shows it, as didi_round_ce defines the case -- and no gradient; nothing is read or written out of bounds
    const bool tg_ok = tg >= 0 && tg < V;
    if (nll && threadIdx.x == 0) nll[blockIdx.x] = keep ? (tg_ok ? __logf(s) + mx - row[tg] : INFINITY) : 0.f;
    __syncthreads();  // the target's logit is read before any lane overwrites it
    for (int n = threadIdx.x; n < V; n += 256) {
        float g = keep && tg_ok ? __expf(row[n] - mx) * inv : 0.f;
        if (keep && n == tg) g -= scale;
        row[n] = g;
    }
}
extern "C" int didi_ce_backward(float* logits, const int64_t* targets, const int64_t* mask, int M, int V, float scale, float* nll,
                                void* stream) {
    DIDI_CHECK_ARG(logits && targets && mask && M >= 0 && V > 0);
    if (M == 0) return DIDI_OK;
    ce_backward_kernel<<<M, 256, 0, as_stream(stream)>>>(logits, targets, mask, V, scale, nll);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ embedding gradients: table_grad[ids[m], :] += scale * dy[m, :]
// nn.Embedding backward (float atomics: rows repeat within a batch); rows equal to `skip_id` (padding_idx, model.py:104) get none.
__global__ void __launch_bounds__(256) embedding_backward_kernel(const int64_t* ids, const float* dy, int M, int H, int64_t V, float scale,
                                                                 int64_t skip_id, float* grad) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int64_t id = ids[m];
    if (id == skip_id || id < 0 || id >= V) return;  // ids outside the table get no gradient (and write nothing)
    for (int c = lane; c < H; c += 64) atomicAdd(grad + (size_t)id * H + c, scale * dy[(size_t)m * H + c]);
}
extern "C" int didi_embedding_backward(const int64_t* ids, const float* dy, int M, int H, int64_t V, float scale, int64_t skip_id, float* grad,
                                       void* stream) {
    DIDI_CHECK_ARG(ids && dy && grad && M >= 0 && H > 0 && V > 0);
    if (M == 0) return DIDI_OK;
    embedding_backward_kernel<<<cdiv(M, 4), 256, 0, as_stream(stream)>>>(ids, dy, M, H, V, scale, skip_id, grad);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ attention backward (fp32, one workgroup per (dialog, head))
// O = softmax(Q K^T s + mask) V.  With P the probabilities and D_i = sum_j P_ij dP_ij:
//   dV = P^T dO,  dP = dO V^T,  dS = P o (dP - D) s,  dQ = dS K,  dK = dS^T Q.
// One head lives in LDS as fp32 (Sq, Sk <= 96, D <= 64); P is recomputed from Q and K, not saved by the forward.  The five
// products are register-tiled fp32 FMA loops (16 x 16 threads, 4 x 4 or 4 x NJ outputs each) fed by 16-byte LDS reads: rows are
// padded to D + 4 (and Sk + 4) floats, so the 16 rows a wave reads at once fall into distinct banks.  HBM traffic is the
// operands once and the three gradients once; the work is 5 x 2 Sq Sk D FLOPs per head on the vector ALUs -- fp32 throughout,
// because the gradient fixtures are checked at 1e-5-class tolerances (MFMA would need the three-product split of the forward).
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ void fma4(float (&acc)[4], float a, const float4& b) {
    acc[0] += a * b.x; acc[1] += a * b.y; acc[2] += a * b.z; acc[3] += a * b.w;
}
__device__ __forceinline__ const float4& ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// C[m][n] = sum_r A[r][m] B[r][n] (m contiguous in A, n in B), m < Mv, n < Nv (both multiples of 4), stored as float4 rows
template <class Store>
__device__ __forceinline__ void mm_tn(const float* A, int lda, const float* B, int ldb, int Mv, int Nv, int R, int ty, int tx, Store store) {
    for (int m0 = 0; m0 < Mv; m0 += 64)
        for (int n0 = 0; n0 < Nv; n0 += 64) {
            const int mo = min(m0 + ty * 4, Mv - 4), no = min(n0 + tx * 4, Nv - 4);
            float acc[4][4] = {};
            for (int r = 0; r < R; ++r) {
                const float4 a = ld4(A + r * lda + mo), b = ld4(B + r * ldb + no);
                fma4(acc[0], a.x, b); fma4(acc[1], a.y, b); fma4(acc[2], a.z, b); fma4(acc[3], a.w, b);
            }
            if (mo == m0 + ty * 4 && no == n0 + tx * 4)
#pragma unroll
                for (int i = 0; i < 4; ++i) store(mo + i, no, make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]));
        }
}

// fp32 MFMA forms of the same products (MF = true: D = 64, Sk = 16 NJ = 64 or 96, Sq a multiple of 16: the training shapes).  v_mfma_f32_16x16x4_f32 multiplies fp32 operands
// exactly and accumulates in fp32, so the 1e-5-class gradient fixtures hold as they do for the FMA loops, at ~4x fewer LDS bytes
// per FLOP and without the FMA issue slots: lane (lr, lg) supplies A[lr][k] and B[k][lr] for one k = f(lg) per instruction and
// receives D[4 lg + r][lr], r = 0..3.  Three operand arrangements, named by how each operand lies in LDS:
//   mf_nt: A[i][k] = X[i0 + i][k], B[k][j] = Y[j0 + j][k]        (both along their rows: 16-byte reads, 4 k-steps each)
//   mf_nn: A[i][k] = X[i0 + i][k], B[k][j] = Y[k][j0 + j]        (A 16-byte reads; B one word per k-step)
//   mf_tn: A[i][k] = X[k][i0 + i], B[k][j] = Y[k][j0 + j]        (both one word per k-step)
// FA: applied to every A element (fmaxf(., 0) where the dropped probabilities, stored negated, must read as zero).
typedef float mf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mf4 mf_mma(float a, float b, mf4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// All NT_ column tiles are computed, without branches: the operands of one k-group are read first, then the MFMAs go out
// interleaved over the tiles, so that consecutive instructions never depend on each other.
template <int NT_, bool RELU_A>
__device__ __forceinline__ void mf_nt(const float* X, int ldx, int i0, const float* Y, int ldy, int K, int lr, int lg, mf4 (&acc)[NT_]) {
    for (int kk = 0; kk < K; kk += 16) {
        float4 a = ld4(X + (i0 + lr) * ldx + kk + 4 * lg);
        if (RELU_A) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
        float4 b[NT_];
#pragma unroll
        for (int j = 0; j < NT_; ++j) b[j] = ld4(Y + (16 * j + lr) * ldy + kk + 4 * lg);
#pragma unroll
        for (int j = 0; j < NT_; ++j) acc[j] = mf_mma(a.x, b[j].x, acc[j]);
#pragma unroll
        for (int j = 0; j < NT_; ++j) acc[j] = mf_mma(a.y, b[j].y, acc[j]);
#pragma unroll
        for (int j = 0; j < NT_; ++j) acc[j] = mf_mma(a.z, b[j].z, acc[j]);
#pragma unroll
        for (int j = 0; j < NT_; ++j) acc[j] = mf_mma(a.w, b[j].w, acc[j]);
    }
}
template <int NT_, bool RELU_A>
__device__ __forceinline__ void mf_nn(const float* X, int ldx, int i0, const float* Y, int ldy, int K, int lr, int lg, mf4 (&acc)[NT_]) {
    for (int kk = 0; kk < K; kk += 16) {
        float4 a = ld4(X + (i0 + lr) * ldx + kk + 4 * lg);
        if (RELU_A) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
        const float av[4] = {a.x, a.y, a.z, a.w};
        const float* yk = Y + (kk + 4 * lg) * ldy + lr;
        float b[4][NT_];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < NT_; ++j) b[s][j] = yk[s * ldy + 16 * j];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < NT_; ++j) acc[j] = mf_mma(av[s], b[s][j], acc[j]);
    }
}
template <int NT_, bool RELU_A>
__device__ __forceinline__ void mf_tn(const float* X, int ldx, int i0, const float* Y, int ldy, int K, int lr, int lg, mf4 (&acc)[NT_]) {
    for (int kk = 0; kk < K; kk += 16) {
        float a[4], b[4][NT_];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a[s] = X[(kk + 4 * s + lg) * ldx + i0 + lr];
            if (RELU_A) a[s] = fmaxf(a[s], 0.f);
#pragma unroll
            for (int j = 0; j < NT_; ++j) b[s][j] = Y[(kk + 4 * s + lg) * ldy + 16 * j + lr];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < NT_; ++j) acc[j] = mf_mma(a[s], b[s][j], acc[j]);
    }
}

// The training forward runs the same fp32 arrangement (FWD: Q K V in LDS, O = P V out), so that attention-probability dropout
// (HF:bert:129, attention_probs_dropout_prob) is ONE mask definition shared by forward and backward: element (b, h, i, j) keeps
// its probability iff word (j & 3) of Philox4x32-10(counter = offset + (((b heads + h) Sq + i) Sk4 + j) / 4, key = seed) is
// >= p 2^32, Sk4 = Sk rounded up to 4.  In LDS a dropped probability is stored NEGATED (probabilities are >= 0, -0.0 included):
// products that want the dropped matrix clamp at zero, the softmax backward takes |P|; kept entries are scaled by 1 / (1 - p).
// diagnostic build (-DDIDI_STAMPS): the backward, which has no use for p.O, writes s_memtime of wave 0 at its phase boundaries there
// (tools/attn_train_stamps.py)
#ifdef DIDI_STAMPS
#define AT_STAMP(i)                                                                                              \
    do {                                                                                                         \
        if (BWD && p.O && threadIdx.x == 0)                                                                      \
            reinterpret_cast<unsigned long long*>(p.O)[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define AT_STAMP(i) do {} while (0)
#endif
template <int NJ, bool BWD, bool MF = false>
__global__ void __launch_bounds__(256) attention_train_kernel(didi_attn_train_args p) {
    extern __shared__ float sm[];
    const int Sq = p.Sq, Sk = p.Sk, D = p.D;
    const int b = blockIdx.y, h = blockIdx.x, tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int LD = D + 4, Sk4 = (Sk + 3) & ~3, LP = Sk4 + 4, L4 = LD / 4, D_4 = D / 4;
    // The MFMA backward keeps TWO operand slots, not four: Q and K are needed by the scores and again only by the last two products
    // (dK, dQ), dO and V only by dV and dP in between.  So dO and V wait in registers (4 float4 each per thread) until the scores are
    // done, then take Q's and K's slots; Q and K are fetched a second time (32 KB, in flight under dP) for dQ / dK.  52 KB instead
    // of 87: THREE workgroups per CU, whose memory and compute phases overlap (240 -> 181 us per launch; with three slots, two
    // workgroups per CU: 197).
    // The MFMA forward at Sk = 64 (LP == LD) keeps no slot for P at all: a wave reads only ITS 16-row strip of Q for the scores and
    // writes the same rows of P, in program order, so P lies on Q: 52 KB, three workgroups per CU.
    constexpr bool OV = MF && BWD, OVF = MF && !BWD && NJ == 4;
    float* q = sm;                                      // [Sq][LD]  (OV: max(Sq, Sk) rows; later V, then Q again)
    float* k = q + (OV ? max(Sq, Sk) : Sq) * LD;        // [Sk][LD]  (OV: max(Sq, Sk) rows; V after the scores, K again for dQ)
    float* v = OV ? k : k + Sk * LD;                    // [Sk][LD]
    float* g = OV ? q : v + Sk * LD;                    // dO [Sq][LD] (BWD only; OV: in Q's slot after the scores, Q again for dK)
    float* P = OVF ? q : OV ? k + max(Sq, Sk) * LD : BWD ? g + Sq * LD : g;  // [Sq][LP]: scores -> probabilities (sign = dropout) -> dS
    float4 rv[OV ? NJ : 1], rgk[OV ? 6 : 1];            // OV: V and dO between their load and the end of the scores
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    AT_STAMP(0);
    if constexpr (MF) {
        // D = 64: a row is 16 float4; every load of the workgroup's four operands is issued before the first LDS store, so the
        // head's 64 KB arrive under ONE memory latency (the generic loop below waits per iteration: 16 us of a 30 us workgroup,
        // tools/attn_train_stamps.py; with the loads batched the phase is the launch's bandwidth share, 7.6 us).  Tried on top and
        // dropped: a workgroup walking three heads of its dialog with the next head's operands prefetched into registers
        // (backward 236 -> 262 us, forward 81 -> 113 us: a third of the workgroups, and the compiler keeps the loads next to their use).
        const int c = tid & 15, rb = tid >> 4;
        float4 rq[6], rg[6], rk[NJ], rv_[OV ? 1 : NJ];
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (rb + 16 * j < Sq) {
                rq[j] = ld4(p.Q + ((size_t)b * Sq + rb + 16 * j) * p.ldq + h * 64 + c * 4);
                if (BWD) (OV ? rgk[OV ? j : 0] : rg[j]) = ld4(p.dO + ((size_t)b * Sq + rb + 16 * j) * p.ldo + h * 64 + c * 4);
            }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            rk[j] = ld4(p.K + ((size_t)b * Sk + rb + 16 * j) * p.ldk + h * 64 + c * 4);
            (OV ? rv[j] : rv_[OV ? 0 : j]) = ld4(p.V + ((size_t)b * Sk + rb + 16 * j) * p.ldv + h * 64 + c * 4);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (rb + 16 * j < Sq) {
                *reinterpret_cast<float4*>(q + (rb + 16 * j) * LD + c * 4) = rq[j];
                if (BWD && !OV) *reinterpret_cast<float4*>(g + (rb + 16 * j) * LD + c * 4) = rg[j];
            }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            *reinterpret_cast<float4*>(k + (rb + 16 * j) * LD + c * 4) = rk[j];
            if constexpr (!OV) *reinterpret_cast<float4*>(v + (rb + 16 * j) * LD + c * 4) = rv_[OV ? 0 : j];
        }
    } else {
    for (int i = tid; i < Sq * L4; i += 256) {
        const int r = i / L4, c = i % L4;
        reinterpret_cast<float4*>(q)[i] = c < D_4 ? ld4(p.Q + ((size_t)b * Sq + r) * p.ldq + h * D + c * 4) : z4;
        if (BWD) reinterpret_cast<float4*>(g)[i] = c < D_4 ? ld4(p.dO + ((size_t)b * Sq + r) * p.ldo + h * D + c * 4) : z4;
    }
    for (int i = tid; i < Sk * L4; i += 256) {
        const int r = i / L4, c = i % L4;
        reinterpret_cast<float4*>(k)[i] = c < D_4 ? ld4(p.K + ((size_t)b * Sk + r) * p.ldk + h * D + c * 4) : z4;
        reinterpret_cast<float4*>(v)[i] = c < D_4 ? ld4(p.V + ((size_t)b * Sk + r) * p.ldv + h * D + c * 4) : z4;
    }
    }
    for (int i = tid; i < Sq * (LP - Sk); i += 256) P[(i / (LP - Sk)) * LP + Sk + i % (LP - Sk)] = 0.f;  // padding columns stay zero
    __syncthreads();
    AT_STAMP(1);  // operands in LDS
    constexpr float NEG = -3.4028234663852886e38f;
    const float scale = p.scale;
    const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
    const mf4 mz = {0.f, 0.f, 0.f, 0.f};
    // ---- scores: P[r][c] = scale q_r . k_c (+ masks); thread = rows ty + 16 i, columns tx + 16 j (MF: wave = 16-row strips)
    if constexpr (MF) {
        for (int i0 = 16 * wave; i0 < Sq; i0 += 64) {
            mf4 acc[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = mz;
            mf_nt<NJ, false>(q, LD, i0, k, LD, 64, lr, lg, acc);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                    const int c = 16 * j + lr;
                    const bool km = p.key_mask && p.key_mask[(size_t)b * Sk + c] == 0;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int r = i0 + 4 * lg + r4;
                        float sc = acc[j][r4] * scale;
                        if (p.pos_bias) sc += p.pos_bias[((size_t)h * Sq + r) * Sk + c];
                        if (km || (p.causal && c > r)) sc = fminf(sc + NEG, NEG);
                        P[r * LP + c] = sc;
                    }
                }
        }
    } else
    for (int r0 = 0; r0 < Sq; r0 += 64) {
        float acc[4][NJ] = {};
        int ra[4], cb[NJ];
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = min(r0 + ty + 16 * i, Sq - 1) * LD;
#pragma unroll
        for (int j = 0; j < NJ; ++j) cb[j] = min(tx + 16 * j, Sk - 1) * LD;
        for (int kk = 0; kk < D; kk += 4) {
            float4 a[4], bb[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = ld4(q + ra[i] + kk);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bb[j] = ld4(k + cb[j] + kk);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] += dot4(a[i], bb[j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int r = r0 + ty + 16 * i, c = tx + 16 * j;
                if (r < Sq && c < Sk) {
                    float sc = acc[i][j] * scale;
                    if (p.pos_bias) sc += p.pos_bias[((size_t)h * Sq + r) * Sk + c];  // T5's relative position bias
                    if ((p.key_mask && p.key_mask[(size_t)b * Sk + c] == 0) || (p.causal && c > r)) sc = fminf(sc + NEG, NEG);
                    P[r * LP + c] = sc;
                }
            }
    }
    __syncthreads();
    AT_STAMP(2);  // scores
    if constexpr (OV) {  // Q's and K's slots are free until dK / dQ: dO and V move in (visible after the softmax's barrier)
        const int c = tid & 15, rb = tid >> 4;
#pragma unroll
        for (int j = 0; j < NJ; ++j) *reinterpret_cast<float4*>(v + (rb + 16 * j) * LD + c * 4) = rv[j];
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (rb + 16 * j < Sq) *reinterpret_cast<float4*>(g + (rb + 16 * j) * LD + c * 4) = rgk[OV ? j : 0];
    }
    // ---- row softmax: four lanes per row
    for (int r = tid >> 2; r < Sq; r += 64) {
        float* row = P + r * LP;
        float mx = -INFINITY, sum = 0.f;
        for (int c = tid & 3; c < Sk; c += 4) mx = fmaxf(mx, row[c]);
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        for (int c = tid & 3; c < Sk; c += 4) { const float e = __expf(row[c] - mx); row[c] = e; sum += e; }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        const float inv = 1.0f / sum;
        for (int c = tid & 3; c < Sk; c += 4) row[c] *= inv;
    }
    __syncthreads();
    AT_STAMP(3);  // softmax
    // ---- dropout: negate the dropped probabilities
    const bool drop = p.drop_p > 0.f;
    const float inv_keep = drop ? 1.0f / (1.0f - p.drop_p) : 1.0f;
    if (drop) {
        const uint32_t thr = (uint32_t)fminf(p.drop_p * 4294967296.0f, 4294967040.0f);
        const int G4 = Sk4 / 4;
        for (int i = tid; i < Sq * G4; i += 256) {
            const int r = i / G4, c4 = i % G4;
            uint32_t w[4];
            philox4x32_10(p.offset + (((uint64_t)b * gridDim.x + h) * Sq + r) * G4 + c4, p.seed, w);
            float4* cell = reinterpret_cast<float4*>(P + r * LP + c4 * 4);
            float4 t = *cell;
            if (w[0] < thr) t.x = -t.x;
            if (w[1] < thr) t.y = -t.y;
            if (w[2] < thr) t.z = -t.z;
            if (w[3] < thr) t.w = -t.w;
            *cell = t;
        }
        __syncthreads();
    }
    if constexpr (!BWD) {
        // ---- O = dropout(P) V: rows ty + 16 i, columns n0 + 4 tx .. + 3 (n0 walks blocks of 64 columns: head dims up to 128)
        if constexpr (MF) {
            for (int i0 = 16 * wave; i0 < Sq; i0 += 64) {
                mf4 acc[4] = {mz, mz, mz, mz};
                mf_nn<4, true>(P, LP, i0, v, LD, 16 * NJ, lr, lg, acc);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
                        p.O[((size_t)b * Sq + i0 + 4 * lg + r4) * p.ldo + h * 64 + 16 * j + lr] = acc[j][r4] * inv_keep;
            }
            return;
        }
        for (int r0 = 0; r0 < Sq; r0 += 64)
            for (int n0 = 0; n0 < D; n0 += 64) {
                const int no = min(n0 + tx * 4, D - 4);
                float acc[4][4] = {};
                int ra[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) ra[i] = min(r0 + ty + 16 * i, Sq - 1) * LP;
                for (int kk = 0; kk < Sk4; kk += 4) {
                    float4 a[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a[i] = ld4(P + ra[i] + kk);
                        a[i].x = fmaxf(a[i].x, 0.f); a[i].y = fmaxf(a[i].y, 0.f); a[i].z = fmaxf(a[i].z, 0.f); a[i].w = fmaxf(a[i].w, 0.f);
                    }
                    const float4 b0 = ld4(v + min(kk, Sk - 1) * LD + no), b1 = ld4(v + min(kk + 1, Sk - 1) * LD + no),
                                 b2 = ld4(v + min(kk + 2, Sk - 1) * LD + no), b3 = ld4(v + min(kk + 3, Sk - 1) * LD + no);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { fma4(acc[i], a[i].x, b0); fma4(acc[i], a[i].y, b1); fma4(acc[i], a[i].z, b2); fma4(acc[i], a[i].w, b3); }
                }
                if (no == n0 + tx * 4)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = r0 + ty + 16 * i;
                        if (r < Sq)
                            *reinterpret_cast<float4*>(p.O + ((size_t)b * Sq + r) * p.ldo + h * D + no) =
                                make_float4(acc[i][0] * inv_keep, acc[i][1] * inv_keep, acc[i][2] * inv_keep, acc[i][3] * inv_keep);
                    }
            }
        return;
    } else if constexpr (MF) {
        AT_STAMP(4);  // dropout
        // ---- dV = dropout(P)^T dO: strips of 16 keys
        for (int j0 = 16 * wave; j0 < Sk; j0 += 64) {
            mf4 acc[4] = {mz, mz, mz, mz};
            mf_tn<4, true>(P, LP, j0, g, LD, Sq, lr, lg, acc);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
                        p.dV[((size_t)b * Sk + j0 + 4 * lg + r4) * p.ldv + h * D + 16 * j + lr] = acc[j][r4] * inv_keep;
        }
        __syncthreads();
        AT_STAMP(5);  // dV
        // Q and K a second time, for dK and dQ: in flight under dP
        float4 rq2[6], rk2[NJ];
        {
            const int c = tid & 15, rb = tid >> 4;
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (rb + 16 * j < Sq) rq2[j] = ld4(p.Q + ((size_t)b * Sq + rb + 16 * j) * p.ldq + h * 64 + c * 4);
#pragma unroll
            for (int j = 0; j < NJ; ++j) rk2[j] = ld4(p.K + ((size_t)b * Sk + rb + 16 * j) * p.ldk + h * 64 + c * 4);
        }
        // ---- dP = (dO V^T) o mask / keep; D_i = sum_c |P| dP (a wave owns whole rows: in-lane over the column tiles, then over the
        // 16 lanes of a row group); dS = |P| (dP - D) scale, back into P
        for (int i0 = 16 * wave; i0 < Sq; i0 += 64) {
            mf4 acc[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = mz;
            mf_nt<NJ, false>(g, LD, i0, v, LD, 64, lr, lg, acc);
            float pv[NJ][4], dr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const float sp = P[(i0 + 4 * lg + r4) * LP + 16 * j + lr];
                        pv[j][r4] = fabsf(sp);
                        acc[j][r4] = __builtin_signbitf(sp) ? 0.f : acc[j][r4] * inv_keep;
                        dr[r4] += pv[j][r4] * acc[j][r4];
                    }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                dr[r4] += __shfl_xor(dr[r4], 1, 64);
                dr[r4] += __shfl_xor(dr[r4], 2, 64);
                dr[r4] += __shfl_xor(dr[r4], 4, 64);
                dr[r4] += __shfl_xor(dr[r4], 8, 64);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) P[(i0 + 4 * lg + r4) * LP + 16 * j + lr] = pv[j][r4] * (acc[j][r4] - dr[r4]) * scale;
        }
        __syncthreads();
        AT_STAMP(6);  // dP -> dS
        {  // dO and V are done with: Q and K back into their slots
            const int c = tid & 15, rb = tid >> 4;
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (rb + 16 * j < Sq) *reinterpret_cast<float4*>(q + (rb + 16 * j) * LD + c * 4) = rq2[j];
#pragma unroll
            for (int j = 0; j < NJ; ++j) *reinterpret_cast<float4*>(k + (rb + 16 * j) * LD + c * 4) = rk2[j];
        }
        __syncthreads();
        // ---- dQ = dS K
        for (int i0 = 16 * wave; i0 < Sq; i0 += 64) {
            mf4 acc[4] = {mz, mz, mz, mz};
            mf_nn<4, false>(P, LP, i0, k, LD, 16 * NJ, lr, lg, acc);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
                        p.dQ[((size_t)b * Sq + i0 + 4 * lg + r4) * p.ldq + h * D + 16 * j + lr] = acc[j][r4];
        }
        AT_STAMP(7);  // dQ
        // ---- dK = dS^T Q
        for (int j0 = 16 * wave; j0 < Sk; j0 += 64) {
            mf4 acc[4] = {mz, mz, mz, mz};
            mf_tn<4, false>(P, LP, j0, q, LD, Sq, lr, lg, acc);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
                        p.dK[((size_t)b * Sk + j0 + 4 * lg + r4) * p.ldk + h * D + 16 * j + lr] = acc[j][r4];
        }
        AT_STAMP(8);  // dK issued
#ifdef DIDI_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        AT_STAMP(9);  // drained
    } else {
        // ---- dV = dropout(P)^T dO
        for (int m0 = 0; m0 < Sk4; m0 += 64) {
            const int mo = min(m0 + ty * 4, Sk4 - 4), no = min(tx * 4, D - 4);
            float acc[4][4] = {};
            for (int r = 0; r < Sq; ++r) {
                float4 a = ld4(P + r * LP + mo);
                const float4 bb = ld4(g + r * LD + no);
                a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
                fma4(acc[0], a.x, bb); fma4(acc[1], a.y, bb); fma4(acc[2], a.z, bb); fma4(acc[3], a.w, bb);
            }
            if (mo == m0 + ty * 4 && no == tx * 4)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (mo + i < Sk)
                        *reinterpret_cast<float4*>(p.dV + ((size_t)b * Sk + mo + i) * p.ldv + h * D + no) =
                            make_float4(acc[i][0] * inv_keep, acc[i][1] * inv_keep, acc[i][2] * inv_keep, acc[i][3] * inv_keep);
        }
        __syncthreads();
        // ---- dP = (dO V^T) o mask / keep in registers; D_i = sum_c |P| dP over the 16 lanes sharing a row; dS = |P| (dP - D) scale
        for (int r0 = 0; r0 < Sq; r0 += 64) {
            float acc[4][NJ] = {};
            int ra[4], cb[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = min(r0 + ty + 16 * i, Sq - 1) * LD;
#pragma unroll
            for (int j = 0; j < NJ; ++j) cb[j] = min(tx + 16 * j, Sk - 1) * LD;
            for (int kk = 0; kk < D; kk += 4) {
                float4 a[4], bb[NJ];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = ld4(g + ra[i] + kk);
#pragma unroll
                for (int j = 0; j < NJ; ++j) bb[j] = ld4(v + cb[j] + kk);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] += dot4(a[i], bb[j]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = min(r0 + ty + 16 * i, Sq - 1);
                float pv[NJ], dr = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int c = tx + 16 * j;
                    const float sp = c < Sk ? P[r * LP + c] : 0.f;
                    pv[j] = fabsf(sp);
                    acc[i][j] = __builtin_signbitf(sp) ? 0.f : acc[i][j] * inv_keep;
                    dr += pv[j] * acc[i][j];
                }
                dr += __shfl_xor(dr, 1, 64);
                dr += __shfl_xor(dr, 2, 64);
                dr += __shfl_xor(dr, 4, 64);
                dr += __shfl_xor(dr, 8, 64);
                if (r0 + ty + 16 * i < Sq)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int c = tx + 16 * j;
                        if (c < Sk) P[r * LP + c] = pv[j] * (acc[i][j] - dr) * scale;
                    }
            }
        }
        __syncthreads();
        // ---- dQ = dS K: rows ty + 16 i, columns 4 tx .. 4 tx + 3
        for (int r0 = 0; r0 < Sq; r0 += 64) {
            const int no = min(tx * 4, D - 4);
            float acc[4][4] = {};
            int ra[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = min(r0 + ty + 16 * i, Sq - 1) * LP;
            for (int kk = 0; kk < Sk4; kk += 4) {
                float4 a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = ld4(P + ra[i] + kk);
                const float4 b0 = ld4(k + min(kk, Sk - 1) * LD + no), b1 = ld4(k + min(kk + 1, Sk - 1) * LD + no),
                             b2 = ld4(k + min(kk + 2, Sk - 1) * LD + no), b3 = ld4(k + min(kk + 3, Sk - 1) * LD + no);
#pragma unroll
                for (int i = 0; i < 4; ++i) { fma4(acc[i], a[i].x, b0); fma4(acc[i], a[i].y, b1); fma4(acc[i], a[i].z, b2); fma4(acc[i], a[i].w, b3); }
            }
            if (no == tx * 4)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = r0 + ty + 16 * i;
                    if (r < Sq)
                        *reinterpret_cast<float4*>(p.dQ + ((size_t)b * Sq + r) * p.ldq + h * D + no) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
                }
        }
        // ---- dK = dS^T Q
        float* dK = p.dK;
        const int ldk = p.ldk;
        mm_tn(P, LP, q, LD, Sk4, D, Sq, ty, tx, [&](int m, int n, const float4& val) {
            if (m < Sk) *reinterpret_cast<float4*>(dK + ((size_t)b * Sk + m) * ldk + h * D + n) = val;
        });
    }
}

static int attention_train_launch(const didi_attn_train_args* a, bool bwd, void* stream) {
    DIDI_CHECK_ARG(a && a->Q && a->K && a->V && a->B >= 0 && a->heads > 0 && a->Sq > 0 && a->Sk > 0 && a->D > 0);
    DIDI_CHECK_ARG(a->drop_p >= 0.f && a->drop_p < 1.f);
    if (bwd) DIDI_CHECK_ARG(a->dO && a->dQ && a->dK && a->dV && !a->pos_bias);  // the bias is a forward-only feature (frozen T5 encoders)
    else DIDI_CHECK_ARG(a->O);
    // the forward takes head dims up to 128 (context encoders with 80- or 128-wide heads: Blenderbot-3B, T5-3B) as long as one
    // head's operands fit the LDS (checked below); the backward's dQ / dV stages are written for D <= 64
    if (a->Sq > 96 || a->Sk > 96 || a->D > (bwd ? 64 : 128) || a->D % 4 != 0) return DIDI_ERR_UNSUPPORTED;
    if (a->causal && a->Sq != a->Sk) return DIDI_ERR_BAD_ARG;
    // 16-byte loads and stores of head slices: leading dimensions in multiples of 4 floats, bases 16-byte aligned
    const int hd = a->heads * a->D;
    DIDI_CHECK_ARG(a->ldq % 4 == 0 && a->ldk % 4 == 0 && a->ldv % 4 == 0 && a->ldo % 4 == 0 && a->ldq >= hd && a->ldk >= hd && a->ldv >= hd && a->ldo >= hd);
    for (const void* ptr : {(const void*)a->Q, (const void*)a->K, (const void*)a->V, (const void*)a->dO, (const void*)a->O, (const void*)a->dQ,
                            (const void*)a->dK, (const void*)a->dV})
        DIDI_CHECK_ARG(reinterpret_cast<uintptr_t>(ptr) % 16 == 0);
    if (a->B == 0) return DIDI_OK;
    const int LD = a->D + 4, LP = ((a->Sk + 3) & ~3) + 4;
    const bool mf = a->Sq % 16 == 0 && (a->Sk == 64 || a->Sk == 96) && a->D == 64;  // the fp32-MFMA form: the training shapes
    const size_t lds = mf && !bwd && a->Sk == 64  // P lies on Q
                           ? sizeof(float) * ((size_t)a->Sq + 2 * a->Sk) * LD
                       : mf && bwd  // two operand slots (Q / dO, K / V), P
                           ? sizeof(float) * ((size_t)2 * std::max(a->Sq, a->Sk) * LD + (size_t)a->Sq * LP)
                           : sizeof(float) * ((size_t)(bwd ? 2 : 1) * a->Sq * LD + (size_t)2 * a->Sk * LD + (size_t)a->Sq * LP);
    if (lds > 160 * 1024 - 256) return DIDI_ERR_UNSUPPORTED;
    static std::mutex mu;
    {
        std::lock_guard<std::mutex> lock(mu);
        for (const void* fn : {reinterpret_cast<const void*>(&attention_train_kernel<4, true>), reinterpret_cast<const void*>(&attention_train_kernel<6, true>),
                               reinterpret_cast<const void*>(&attention_train_kernel<4, false>), reinterpret_cast<const void*>(&attention_train_kernel<6, false>),
                               reinterpret_cast<const void*>(&attention_train_kernel<4, true, true>), reinterpret_cast<const void*>(&attention_train_kernel<4, false, true>),
                               reinterpret_cast<const void*>(&attention_train_kernel<6, true, true>), reinterpret_cast<const void*>(&attention_train_kernel<6, false, true>)})
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess) return DIDI_ERR_LAUNCH;
    }
    didi_attn_train_args p = *a;
    if (!(p.scale > 0.f)) p.scale = 1.0f / sqrtf((float)a->D);
    const dim3 grid(a->heads, a->B);
    hipStream_t st = as_stream(stream);
    if (mf && bwd) {
        if (a->Sk == 64) attention_train_kernel<4, true, true><<<grid, 256, lds, st>>>(p);
        else attention_train_kernel<6, true, true><<<grid, 256, lds, st>>>(p);
    } else if (mf) {
        if (a->Sk == 64) attention_train_kernel<4, false, true><<<grid, 256, lds, st>>>(p);
        else attention_train_kernel<6, false, true><<<grid, 256, lds, st>>>(p);
    } else if (bwd) {
        if (a->Sk <= 64) attention_train_kernel<4, true><<<grid, 256, lds, st>>>(p);
        else attention_train_kernel<6, true><<<grid, 256, lds, st>>>(p);
    } else {
        if (a->Sk <= 64) attention_train_kernel<4, false><<<grid, 256, lds, st>>>(p);
        else attention_train_kernel<6, false><<<grid, 256, lds, st>>>(p);
    }
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

extern "C" int didi_attention_train_forward(const didi_attn_train_args* a, void* stream) { return attention_train_launch(a, false, stream); }
extern "C" int didi_attention_train_backward(const didi_attn_train_args* a, void* stream) { return attention_train_launch(a, true, stream); }

extern "C" int didi_attention_backward(const float* Q, const float* K, const float* V, const float* dO, const int32_t* key_mask, int B, int heads,
                                       int Sq, int Sk, int D, int ldq, int ldk, int ldv, int ldo, int causal, float scale, float* dQ, float* dK,
                                       float* dV, void* stream) {
    didi_attn_train_args a{};
    a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.key_mask = key_mask;
    a.B = B; a.heads = heads; a.Sq = Sq; a.Sk = Sk; a.D = D;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.causal = causal; a.scale = scale;
    a.dQ = dQ; a.dK = dK; a.dV = dV;
    return attention_train_launch(&a, true, stream);
}

// ------------------------------------------------------------------ hidden dropout (BertEmbeddings / BertSelfOutput / BertOutput)
// out = x o mask / (1 - p) + residual; mask of flat element e: word (e & 3) of Philox4x32-10(offset + e / 4, seed) >= p 2^32.
// The backward is the same call on the gradient (residual = NULL) with the same (seed, offset).
__global__ void __launch_bounds__(256) dropout_add_kernel(const float* x, const float* residual, size_t n4, float p, uint64_t seed,
                                                          uint64_t offset, float* out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const uint32_t thr = (uint32_t)fminf(p * 4294967296.0f, 4294967040.0f);
    const float inv_keep = 1.0f / (1.0f - p);
    uint32_t w[4];
    philox4x32_10(offset + i, seed, w);
    float4 t = reinterpret_cast<const float4*>(x)[i];
    t.x = w[0] < thr ? 0.f : t.x * inv_keep;
    t.y = w[1] < thr ? 0.f : t.y * inv_keep;
    t.z = w[2] < thr ? 0.f : t.z * inv_keep;
    t.w = w[3] < thr ? 0.f : t.w * inv_keep;
    if (residual) {
        const float4 r = reinterpret_cast<const float4*>(residual)[i];
        t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
    }
    reinterpret_cast<float4*>(out)[i] = t;
}
extern "C" int didi_dropout_add(const float* x, const float* residual, size_t n, float p, uint64_t seed, uint64_t offset, float* out, void* stream) {
    DIDI_CHECK_ARG(x && out && n % 4 == 0 && p >= 0.f && p < 1.f);
    if (n == 0) return DIDI_OK;
    dropout_add_kernel<<<(unsigned)cdiv((long)(n / 4), 256), 256, 0, as_stream(stream)>>>(x, residual, n / 4, p, seed, offset, out);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ AdamW (torch.optim.AdamW semantics, model.py:136-158)
// p -= lr wd p ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr (m / bc1) / (sqrt(v / bc2) + eps), with g pre-scaled by
// `gscale` (the gradient-clipping coefficient, configs/train.yaml:45).  bc1 = 1 - b1^t, bc2 = 1 - b2^t from the caller.
__global__ void __launch_bounds__(256) adamw_kernel(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2, const float* gscale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gs = gscale ? *gscale : 1.0f;
    const float gi = g[i] * gs;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    pi -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
}
extern "C" int didi_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int step, const float* gscale, void* stream) {
    DIDI_CHECK_ARG(p && g && m && v && step >= 1);
    if (n == 0) return DIDI_OK;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    adamw_kernel<<<(unsigned)cdiv((long)n, 256), 256, 0, as_stream(stream)>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, gscale);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// sum of squares of x (for the global gradient norm): out[0] += sum x^2 (float atomic per block; out zeroed by the caller)
__global__ void __launch_bounds__(256) sumsq_kernel(const float* x, size_t n, float* out) {
    __shared__ float red[4];
    float a = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += x[i] * x[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}
extern "C" int didi_sumsq_accumulate(const float* x, size_t n, float* out, void* stream) {
    DIDI_CHECK_ARG(x && out);
    if (n == 0) return DIDI_OK;
    const unsigned grid = (unsigned)std::min<long>(1024, cdiv((long)n, 256));
    sumsq_kernel<<<grid, 256, 0, as_stream(stream)>>>(x, n, out);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ decoder input of the training forward, kept in fp32
// x_in[b, s, :] = x_t[b, s, :] + time_embeds[t_b, :] + pos[s, :] + tok0   (model.py:190-191 + BertEmbeddings with inputs_embeds,
// HF:models/bert/modeling_bert.py:95-115).  The sampler fuses this into its LayerNorm kernel; training keeps the LayerNorm INPUT
// because the backward needs it.
__global__ void __launch_bounds__(256) train_embed_sum_kernel(const float* xt, const float* te, const int64_t* t, const float* pos,
                                                              const float* tok0, int S, int H4, int T_rows, size_t n4, float* out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)(i % H4);
    const size_t row = i / H4;
    const int s = (int)(row % S);
    const int64_t tb = t[row / S];
    const float nan_ = __builtin_nanf("");  // a timestep outside the table poisons its dialog instead of reading out of bounds
    const float4 a = reinterpret_cast<const float4*>(xt)[i],
                 b = tb >= 0 && tb < T_rows ? reinterpret_cast<const float4*>(te)[(size_t)tb * H4 + c] : make_float4(nan_, nan_, nan_, nan_),
                 p = reinterpret_cast<const float4*>(pos)[(size_t)s * H4 + c], k = reinterpret_cast<const float4*>(tok0)[c];
    reinterpret_cast<float4*>(out)[i] = make_float4(a.x + b.x + p.x + k.x, a.y + b.y + p.y + k.y, a.z + b.z + p.z + k.z, a.w + b.w + p.w + k.w);
}
extern "C" int didi_train_embed_sum(const float* x_t, const float* time_embeds, const int64_t* t, const float* pos_emb, const float* tok0,
                                    int B, int S, int H, int T_rows, float* out, void* stream) {
    DIDI_CHECK_ARG(x_t && time_embeds && t && pos_emb && tok0 && out && B >= 0 && S > 0 && H > 0 && H % 4 == 0 && T_rows > 0);
    const size_t n4 = (size_t)B * S * (H / 4);
    if (n4 == 0) return DIDI_OK;
    train_embed_sum_kernel<<<(unsigned)cdiv((long)n4, 256), 256, 0, as_stream(stream)>>>(x_t, time_embeds, t, pos_emb, tok0, S, H / 4, T_rows, n4, out);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// out[b, c] = sum_s x[b, s, c]: the per-dialog sums that scatter into the time-embedding rows
__global__ void __launch_bounds__(256) seq_sum_kernel(const float* x, int S, int H, float* out) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < H; c += 256) {
        float a = 0.f;
        for (int s = 0; s < S; ++s) a += x[((size_t)b * S + s) * H + c];
        out[(size_t)b * H + c] = a;
    }
}
extern "C" int didi_seq_sum(const float* x, int B, int S, int H, float* out, void* stream) {
    DIDI_CHECK_ARG(x && out && B >= 0 && S > 0 && H > 0);
    if (B == 0) return DIDI_OK;
    seq_sum_kernel<<<B, 256, 0, as_stream(stream)>>>(x, S, H, out);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ gradient arriving at the target embeddings (model.py:202-235)
// emb feeds the loss four ways: the mse's t == 1 branch directly, and through x_0 = emb + std_0 n0 the mse's other branch,
// t_T = mean(x_0^2 m) / (sigma_T^2 + 1), x_t = (x_0 + s n1) c (weight c_b of the decoder-input gradient) and, without the anchor
// loss, the cross-entropy on classifier(x_0) (`extra`).  One pass:
//   out = m (-w_emb (x0hat - emb) - w_x0 (x0hat - x0) + w_T x0) + c_b dxt + extra
__global__ void __launch_bounds__(256) train_target_backward_kernel(const float* x0hat, const float* emb, const float* x0, const int64_t* mask,
                                                                    float w_emb, float w_x0, float w_T, const float* c_b, const float* dxt,
                                                                    const float* extra, int S, int H4, size_t n4, float* out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const size_t row = i / H4;
    const float m = mask[row] != 0 ? 1.f : 0.f, cb = c_b[row / S];
    const float4 h = reinterpret_cast<const float4*>(x0hat)[i], e = reinterpret_cast<const float4*>(emb)[i],
                 z = reinterpret_cast<const float4*>(x0)[i], g = reinterpret_cast<const float4*>(dxt)[i];
    float4 x = extra ? reinterpret_cast<const float4*>(extra)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    auto f = [&](float hh, float ee, float zz, float gg, float xx) { return m * (-w_emb * (hh - ee) - w_x0 * (hh - zz) + w_T * zz) + cb * gg + xx; };
    reinterpret_cast<float4*>(out)[i] = make_float4(f(h.x, e.x, z.x, g.x, x.x), f(h.y, e.y, z.y, g.y, x.y), f(h.z, e.z, z.z, g.z, x.z), f(h.w, e.w, z.w, g.w, x.w));
}
extern "C" int didi_train_target_backward(const float* x0hat, const float* emb, const float* x0, const int64_t* mask, float w_emb, float w_x0,
                                          float w_T, const float* c_b, const float* dxt, const float* extra, int B, int S, int H, float* out,
                                          void* stream) {
    DIDI_CHECK_ARG(x0hat && emb && x0 && mask && c_b && dxt && out && B >= 0 && S > 0 && H > 0 && H % 4 == 0);
    const size_t n4 = (size_t)B * S * (H / 4);
    if (n4 == 0) return DIDI_OK;
    train_target_backward_kernel<<<(unsigned)cdiv((long)n4, 256), 256, 0, as_stream(stream)>>>(x0hat, emb, x0, mask, w_emb, w_x0, w_T, c_b, dxt, extra, S,
                                                                                           H / 4, n4, out);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}
