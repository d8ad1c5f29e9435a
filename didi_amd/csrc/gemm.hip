// MFMA bf16 GEMM for the decoder's Linear layers and the rounding (classifier + argmax) step.
//
//   C[M,N] = epilogue( sum over passes p of A_p[M,K] * W_p[N,K]^T )      fp32 accumulate
//
// Tile: 128(M) x 128(N) x 64(K) per 256-thread workgroup; 4 waves as 2(M) x 2(N), each wave a
// 64x64 sub-tile = 4x4 v_mfma_f32_16x16x32_bf16 accumulators.  The MFMA is issued "swapped"
// (A-operand = weight rows, B-operand = activation rows) so that every lane ends up holding FOUR
// CONSECUTIVE OUTPUT COLUMNS n of one output row m: the epilogue then loads bias as one float4 and
// stores 8-byte bf16 packs / 16-byte fp32 vectors instead of scalars.
//
// LDS image (same for both operands): [128 rows][64 bf16] = 128-byte rows, 16-byte chunk c of row
// r stored at chunk position c ^ (r & 7).  With that XOR the 16-lane groups in which the hardware
// services ds_read_b128 hit 16 distinct 16-byte slots of the 256-byte bank row: conflict-free
// fragment reads (cdna_hip_programming.md T2).  Two LDS stages (64 KiB), register prefetch of the
// next K-tile issued before the current tile's MFMAs, one barrier per K-tile.
//
// "bf16x3": the three products hi*hi, hi*lo, lo*hi are three extra trips of the SAME K loop over
// different plane pointers into the same accumulators -- the kernel sees a GEMM with K' = 3K.
//
// Grid: 1-D, XCD-aware: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs,
// so id -> (id % 8) * chunk + id / 8 gives each XCD (private 4 MiB L2) a contiguous band of output
// row-tiles: a band's activation rows are fetched from HBM once per XCD while the small weight
// matrix (<= 4.7 MB) stays L2-resident.
#include "common.hpp"

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (128 * 128)  // one operand tile in LDS

struct GemmParams {
    int M, N, K;
    int npass;
    const u16* A[3];
    const u16* W[3];
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
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// Accumulate one 128x128 output tile at (m0, n0) into acc.  All 256 threads participate.
__device__ __forceinline__ void gemm_tile(const GemmParams& p, char* smem, int m0, int n0, f32x4 (&acc)[4][4]) {
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
    const int so##I = swz(c##I >> 3, c##I & 7);                              \
    const size_t a_off##I = (size_t)min(m0 + (c##I >> 3), p.M - 1) * p.lda + (c##I & 7) * 8; \
    const size_t w_off##I = (size_t)min(n0 + (c##I >> 3), p.N - 1) * p.ldw + (c##I & 7) * 8;
    GEMM_STAGE(0) GEMM_STAGE(1) GEMM_STAGE(2) GEMM_STAGE(3)
#undef GEMM_STAGE
    const int kt_per_pass = (p.K + BK - 1) / BK;  // last K-tile may be partial: chunks past K load as zero
    const int nt = kt_per_pass * p.npass;

    // (no lambdas / runtime-indexed arrays here: either leaves the prefetch registers in scratch)
#define GEMM_GLOAD(T)                                                                   \
    {                                                                                   \
        const int pass_ = (T) / kt_per_pass;                                            \
        const int k0_ = ((T) - pass_ * kt_per_pass) * BK;                               \
        const u16* Ab_ = (pass_ == 0 ? p.A[0] : pass_ == 1 ? p.A[1] : p.A[2]) + k0_;    \
        const u16* Wb_ = (pass_ == 0 ? p.W[0] : pass_ == 1 ? p.W[1] : p.W[2]) + k0_;    \
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
                const int nl = wn * 64 + i * 16 + lr;
                const int ml = wm * 64 + i * 16 + lr;
                wf[i] = *reinterpret_cast<const bf16x8*>(Ws + swz(nl, chunk));
                af[i] = *reinterpret_cast<const bf16x8*>(As + swz(ml, chunk));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
        if (more) GEMM_LSTORE(buf ^ 1);
        __syncthreads();
    }
#undef GEMM_GLOAD
#undef GEMM_LSTORE
}

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == DIDI_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == DIDI_ACT_TANH) return tanhf(v);
    return v;
}

// bijective XCD remap of a 1-D grid (cdna_hip_programming.md, "XCD swizzle must be bijective")
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = id & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

__global__ void __launch_bounds__(256, 2) gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    f32x4 acc[4][4];
    gemm_tile(p, smem, m0, n0, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + lr;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + lg * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
            if (p.bias) {
                const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if (p.act != DIDI_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], p.act);
            }
            if (p.R_f32) {
                const float4 r = *reinterpret_cast<const float4*>(p.R_f32 + (size_t)m * p.ldr + n);
                v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
            } else if (p.R_hi) {
                float rh[4], rl[4] = {0.f, 0.f, 0.f, 0.f};
                unpack4(*reinterpret_cast<const uint2*>(p.R_hi + (size_t)m * p.ldr + n), rh);
                if (p.R_lo) unpack4(*reinterpret_cast<const uint2*>(p.R_lo + (size_t)m * p.ldr + n), rl);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += rh[e] + rl[e];
            }
            const size_t o = (size_t)m * p.ldc + n;
            if (p.C_f32) *reinterpret_cast<float4*>(p.C_f32 + o) = make_float4(v[0], v[1], v[2], v[3]);
            if (p.C_hi) {
                uint2 h, l;
                split4(v, h, l);
                *reinterpret_cast<uint2*>(p.C_hi + o) = h;
                if (p.C_lo) *reinterpret_cast<uint2*>(p.C_lo + o) = l;
            }
        }
    }
}

static int fill_passes(GemmParams& p, int precision, const u16* A_hi, const u16* A_lo, const u16* W_hi, const u16* W_lo) {
    if (precision == DIDI_PREC_BF16) {
        p.npass = 1;
        p.A[0] = A_hi; p.W[0] = W_hi;
        p.A[1] = p.A[2] = A_hi; p.W[1] = p.W[2] = W_hi;
        return DIDI_OK;
    }
    if (precision == DIDI_PREC_BF16X3) {
        if (!A_lo || !W_lo) return DIDI_ERR_BAD_ARG;
        p.npass = 3;  // small cross terms first, the dominant hi*hi product last
        p.A[0] = A_lo; p.W[0] = W_hi;
        p.A[1] = A_hi; p.W[1] = W_lo;
        p.A[2] = A_hi; p.W[2] = W_hi;
        return DIDI_OK;
    }
    return DIDI_ERR_BAD_ARG;
}

static inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

extern "C" int didi_gemm_bf16(const didi_gemm_args* a, void* stream) {
    DIDI_CHECK_ARG(a && a->A_hi && a->W_hi && a->M >= 0 && a->N > 0 && a->K > 0);
    DIDI_CHECK_ARG(a->C_f32 || a->C_hi);
    DIDI_CHECK_ARG(!(a->C_lo && !a->C_hi));
    if (a->K % 8 != 0 || a->N % 4 != 0) return DIDI_ERR_UNSUPPORTED;
    DIDI_CHECK_ARG(a->lda % 8 == 0 && a->ldw % 8 == 0 && a->ldc % 4 == 0 && a->lda >= a->K && a->ldw >= a->K && a->ldc >= a->N);
    DIDI_CHECK_ARG(al16(a->A_hi) && al16(a->A_lo) && al16(a->W_hi) && al16(a->W_lo) && al16(a->bias) && al16(a->R_f32));
    DIDI_CHECK_ARG((reinterpret_cast<uintptr_t>(a->R_hi) & 7) == 0 && (reinterpret_cast<uintptr_t>(a->R_lo) & 7) == 0);
    DIDI_CHECK_ARG(al16(a->C_f32) && (reinterpret_cast<uintptr_t>(a->C_hi) & 7) == 0 && (reinterpret_cast<uintptr_t>(a->C_lo) & 7) == 0);
    if (a->R_f32 || a->R_hi) DIDI_CHECK_ARG(a->ldr % 4 == 0 && a->ldr >= a->N);
    DIDI_CHECK_ARG(a->act >= DIDI_ACT_NONE && a->act <= DIDI_ACT_TANH);
    if (a->M == 0) return DIDI_OK;
    GemmParams p{};
    p.M = a->M; p.N = a->N; p.K = a->K;
    int rc = fill_passes(p, a->precision, a->A_hi, a->A_lo, a->W_hi, a->W_lo);
    if (rc != DIDI_OK) return rc;
    p.lda = a->lda; p.ldw = a->ldw;
    p.bias = a->bias; p.act = a->act;
    p.R_f32 = a->R_f32; p.R_hi = a->R_hi; p.R_lo = a->R_lo; p.ldr = a->ldr;
    p.C_f32 = a->C_f32; p.C_hi = a->C_hi; p.C_lo = a->C_lo; p.ldc = a->ldc;
    p.tiles_m = cdiv(a->M, BM); p.tiles_n = cdiv(a->N, BN);
    const int grid = p.tiles_m * p.tiles_n;
    gemm_kernel<<<grid, 256, 4 * TILE_BYTES, as_stream(stream)>>>(p);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}

// ------------------------------------------------------------------ rounding: classifier GEMM + argmax
// Each workgroup owns one 128-row tile and a chunk of vocabulary tiles; it keeps a per-lane running
// (max, index) and never writes logits.  Cross-lane / cross-wave reduction once per workgroup.
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__global__ void __launch_bounds__(256, 2) gemm_argmax_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int mt = blockIdx.x % p.tiles_m, chunk = blockIdx.x / p.tiles_m;
    const int m0 = mt * BM;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lg = lane >> 4;
    float best[4];
    int bidx[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) { best[mi] = -INFINITY; bidx[mi] = 0x7fffffff; }
    const int t_begin = chunk * p.tiles_per_chunk;
    const int t_end = min(t_begin + p.tiles_per_chunk, p.tiles_n);
    for (int tn = t_begin; tn < t_end; ++tn) {
        const int n0 = tn * BN;
        f32x4 acc[4][4];
        gemm_tile(p, smem, m0, n0, acc);
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
    }
    // lanes lr, lr+16, lr+32, lr+48 hold the same rows
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float ov = __shfl_xor(best[mi], o, 64);
            const int oi = __shfl_xor(bidx[mi], o, 64);
            if (better(ov, oi, best[mi], bidx[mi])) { best[mi] = ov; bidx[mi] = oi; }
        }
    }
    __syncthreads();  // all waves are done with the staging buffers
    float* rv = reinterpret_cast<float*>(smem);       // [128]
    int* ri = reinterpret_cast<int*>(smem + 512);      // [128]
    if (wn == 1 && lg == 0) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) { rv[wm * 64 + mi * 16 + lr] = best[mi]; ri[wm * 64 + mi * 16 + lr] = bidx[mi]; }
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
            }
        }
    }
}

__global__ void __launch_bounds__(256) argmax_reduce_kernel(const float2* partial, int M, int nchunks, int V, int64_t* ids64, int32_t* ids32) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int c = 0; c < nchunks; ++c) {
        const float2 q = partial[(size_t)m * nchunks + c];
        const int i = __float_as_int(q.y);
        if (better(q.x, i, bv, bi)) { bv = q.x; bi = i; }
    }
    if (bi < 0 || bi >= V) bi = 0;
    if (ids64) ids64[m] = bi;
    if (ids32) ids32[m] = bi;
}

extern "C" size_t didi_round_argmax_bytes(int M, int V) {
    if (M <= 0 || V <= 0) return 0;
    return (size_t)M * (size_t)cdiv(V, BN) * sizeof(float2);
}

extern "C" int didi_round_argmax(const uint16_t* X_hi, const uint16_t* X_lo, int ldx, const uint16_t* W_hi,
                                 const uint16_t* W_lo, int ldw, int M, int V, int K, int precision, void* partial,
                                 int64_t* ids64, int32_t* ids32, void* stream) {
    DIDI_CHECK_ARG(X_hi && W_hi && partial && (ids64 || ids32) && M >= 0 && V > 0 && K > 0);
    if (K % 8 != 0) return DIDI_ERR_UNSUPPORTED;
    DIDI_CHECK_ARG(ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K);
    DIDI_CHECK_ARG(al16(X_hi) && al16(X_lo) && al16(W_hi) && al16(W_lo) && (reinterpret_cast<uintptr_t>(partial) & 7) == 0);
    if (M == 0) return DIDI_OK;
    GemmParams p{};
    p.M = M; p.N = V; p.K = K;
    int rc = fill_passes(p, precision, X_hi, X_lo, W_hi, W_lo);
    if (rc != DIDI_OK) return rc;
    p.lda = ldx; p.ldw = ldw;
    p.tiles_m = cdiv(M, BM); p.tiles_n = cdiv(V, BN);
    int nchunks = cdiv(1024, p.tiles_m);  // aim at >= 4 workgroups per CU
    if (nchunks > p.tiles_n) nchunks = p.tiles_n;
    if (nchunks < 1) nchunks = 1;
    p.tiles_per_chunk = cdiv(p.tiles_n, nchunks);
    p.nchunks = cdiv(p.tiles_n, p.tiles_per_chunk);
    p.partial = reinterpret_cast<float2*>(partial);
    hipStream_t st = as_stream(stream);
    gemm_argmax_kernel<<<p.tiles_m * p.nchunks, 256, 4 * TILE_BYTES, st>>>(p);
    DIDI_LAUNCH_CHECK();
    argmax_reduce_kernel<<<cdiv(M, 256), 256, 0, st>>>(p.partial, M, p.nchunks, V, ids64, ids32);
    DIDI_LAUNCH_CHECK();
    return DIDI_OK;
}
