// Stand-alone test bed: PERSISTENT form of the staggered 256x192 loop (variant 8 of gemm_lab1.hip).
// One workgroup per CU walks its XCD's tiles; the next tile's first two K-tiles are issued before the epilogue, all per-lane
// DMA state is tile-invariant (buffer descriptors + scalar tile bases), so nothing per-lane is recomputed per tile.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gemm_lab1.hip -o tools/gemm_lab1 && tools/gemm_lab1 [iters]
//
// K-tile 64 (128-byte LDS rows), two 56-KiB stages, 8 waves (4 along M x 2 along N, wave tile 64 x 96).
// Every variant accumulates in the same order, so all must agree with variant 0 bit for bit.
//   0  library loop: one-ahead LDS-DMA, one barrier per K-tile, all waves in phase
//   5  two wave groups staggered by half a K-tile; group X issues the whole stage's DMA in its MEM slot
//   8  staggered, three barriers per K-tile: a stage buffer is released as soon as group Y has read it
//      (slot B1), so DMA(t+2) starts 1.5 K-tiles before its first use; group Y issues the activation
//      pieces in slot B2 (it is idle there), group X the weight pieces after its reads in slot A
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short u16;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

struct P {
    int M, N, K, tiles_n;
    const u16 *A, *W;
    float* C;
    unsigned long long* stamps;
};

__device__ __forceinline__ int swz128(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ void glds16(const u16* src, char* dst) {
    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)dst, 16, 0, 0);
}
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = id & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

constexpr int BMM = 256, BNN = 192, BKK = 64;
constexpr int A_BYTES = BMM * 128, NA = 32, NW = 24, STAGE = (NA + NW) * 1024;
constexpr int LDS_TOTAL = 2 * STAGE;

#define READ_FRAGS(ST)                                                                          \
    do {                                                                                        \
        const char *As_ = (ST), *Ws_ = (ST) + A_BYTES;                                          \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                      \
            _Pragma("unroll") for (int i = 0; i < 6; ++i)                                       \
                wf[ks][i] = *reinterpret_cast<const bf16x8*>(Ws_ + swz128(wn * 96 + i * 16 + lr, ks * 4 + lg)); \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                       \
                af[ks][i] = *reinterpret_cast<const bf16x8*>(As_ + swz128(wm * 64 + i * 16 + lr, ks * 4 + lg)); \
        }                                                                                       \
    } while (0)

// MFMAs of k-step KS, weight fragments NI0 .. NI1-1
#define MFMA_RANGE(KS, NI0, NI1)                                                                           \
    _Pragma("unroll") for (int ni = (NI0); ni < (NI1); ++ni)                                               \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                   \
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[KS][ni], af[KS][mi], acc[ni][mi], 0, 0, 0);


typedef __attribute__((address_space(3))) void lds_v;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t r, int voff, int soff, char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_v*)dst, 16, voff, soff, 0, 0);
}

// PERSIST = 0: one tile per workgroup (grid = tiles), 1: grid = 256, each workgroup walks tiles
template <int PERSIST>
__global__ void __launch_bounds__(512, 2) lab2_kernel(P p, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lg = lane >> 4;
    const int grp = wave >> 2, w4 = wave & 3;
    // tile-invariant per-lane DMA state: ONE byte offset (piece j of a wave lies 32 rows below piece j-1, and the swizzled
    // 16-byte slot depends on (lane / 8) & 7 only); the piece, the K-tile and the tile go into the scalar offset
    const int q0 = grp ? w4 : NA + w4;
    const int row0 = w4 * 8 + lane / 8;
    const int voff0 = (row0 * p.K + ((lane % 8) ^ (row0 & 7)) * 8) * 2;
    const int jstride = 32 * p.K * 2;  // bytes between a wave's consecutive pieces in the source matrix
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((size_t)p.M * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)((size_t)p.N * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc = grp ? rA : rW;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)((size_t)p.M * p.N * 4), 0x00020000);
    const int cvoff = (lr * p.N + lg * 4) * 4;
    const int nt = p.K / BKK;
    const int np = grp ? 8 : 6;
#define ISSUE2(SB, T, BUF)                                                                                   \
    do {                                                                                                     \
        char* st_ = smem + (BUF) * STAGE;                                                                    \
        if (grp) { _Pragma("unroll") for (int j = 0; j < 8; ++j) blds16(rsrc, voff0, (SB) + (T) * 128 + j * jstride, st_ + (q0 + 4 * j) * 1024); } \
        else { _Pragma("unroll") for (int j = 0; j < 6; ++j) blds16(rsrc, voff0, (SB) + (T) * 128 + j * jstride, st_ + (q0 + 4 * j) * 1024); }     \
    } while (0)

    int first, last, step;
    if (PERSIST) {
        const int x = blockIdx.x & 7, s = blockIdx.x >> 3, q = ntiles >> 3;  // ntiles % 8 == 0 in this lab
        first = x * q + s; last = (x + 1) * q; step = 32;
    } else {
        first = xcd_remap(blockIdx.x, gridDim.x); last = first + 1; step = 1;
    }
    int tile = first;
    int sb = 0;
    if (tile < last) {
        const int m0 = (tile / p.tiles_n) * BMM, n0 = (tile % p.tiles_n) * BNN;
        sb = (grp ? m0 : n0) * p.K * 2;
        ISSUE2(sb, 0, 0);
        if (nt > 1) ISSUE2(sb, 1, 1);
    }
    (void)np;
    for (; tile < last; tile += step) {
        const int m0 = (tile / p.tiles_n) * BMM, n0 = (tile % p.tiles_n) * BNN;
        // the two wave groups run SEPARATE copies of the tile body (loop, next-tile issue, epilogue): accumulators and fragments
        // then live and die inside one branch and nothing is merged at the tile loop's back-edge
#define ZERO_ACC()                                                                  \
        f32x4 acc[6][4];                                                            \
        _Pragma("unroll") for (int i = 0; i < 6; ++i)                               \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; \
        bf16x8 wf[2][6], af[2][4];                                                  \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) { wf[0][i] = wf[1][i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { af[0][i] = af[1][i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
#define NEXT_ISSUE()                                                                \
        if (tile + step < last) {                                                   \
            const int t2 = tile + step;                                             \
            sb = (grp ? (t2 / p.tiles_n) * BMM : (t2 % p.tiles_n) * BNN) * p.K * 2; \
            ISSUE2(sb, 0, 0);                                                       \
            if (nt > 1) ISSUE2(sb, 1, 1);                                           \
        }
#ifdef LAB_TINY_EPI  /* the rounding kernel's situation: almost nothing leaves the workgroup */
#define EPILOGUE()                                                                  \
        {                                                                           \
            f32x4 t_ = acc[0][0];                                                   \
            _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                        \
                _Pragma("unroll") for (int ni = 0; ni < 6; ++ni)                    \
                    if (mi + ni) t_ = t_ + acc[ni][mi];                             \
            if (lg == 0 && wn == 0) p.C[(size_t)(m0 + wm * 64 + lr) * p.tiles_n + n0 / BNN] = t_[0] + t_[1] + t_[2] + t_[3]; \
        }
#else
#define EPILOGUE()                                                                  \
        {                                                                           \
            const int sbase = ((m0 + wm * 64) * p.N + n0 + wn * 96) * 4;            \
            _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                      \
                const int so = sbase + mi * 16 * p.N * 4;                           \
                _Pragma("unroll") for (int ni = 0; ni < 6; ++ni)                    \
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[ni][mi]), rC, cvoff + ni * 64, so, 0); \
            }                                                                       \
        }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (grp == 0) {
            ZERO_ACC()
            for (int i = 0; i < nt; ++i) {
                const char* st = smem + (i & 1) * STAGE;
                READ_FRAGS(st);
                if (i >= 1 && i + 1 < nt) ISSUE2(sb, i + 1, (i + 1) & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();  // end A_i
                __builtin_amdgcn_s_setprio(1);
                MFMA_RANGE(0, 0, 4)
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();  // end B1_i
                MFMA_RANGE(0, 4, 6)
                MFMA_RANGE(1, 0, 6)
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();  // end B2_i
            }
            NEXT_ISSUE()
            EPILOGUE()
        } else {
            ZERO_ACC()
            for (int i = 0; i < nt; ++i) {
                if (i >= 1) {
                    __builtin_amdgcn_s_setprio(1);
                    MFMA_RANGE(0, 0, 6)
                    MFMA_RANGE(1, 0, 6)
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_barrier();  // end A_i
                const char* st = smem + (i & 1) * STAGE;
                READ_FRAGS(st);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();  // end B1_i
                if (i + 2 < nt) {
                    ISSUE2(sb, i + 2, i & 1);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();  // end B2_i
            }
            NEXT_ISSUE()
            MFMA_RANGE(0, 0, 6)
            MFMA_RANGE(1, 0, 6)
            EPILOGUE()
        }
        __builtin_amdgcn_sched_barrier(0);  // the next tile's accumulators must not be set up before these stores have issued
    }
}

static u16 f2bf(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (u16)(u >> 16);
}

static void fill(std::vector<u16>& hi, size_t n, uint64_t seed, float scale) {
    hi.resize(n);
    uint64_t s = seed;
    for (size_t i = 0; i < n; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        hi[i] = f2bf(scale * ((float)((s >> 40) & 0xffffff) / 8388608.0f - 1.0f));
    }
}

template <int PERSIST>
static float run(const P& p, int iters, hipStream_t st) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab2_kernel<PERSIST>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
    const int ntiles = (p.M + BMM - 1) / BMM * p.tiles_n;
    const int grid = PERSIST ? 256 : ntiles;
    for (int i = 0; i < 3; ++i) lab2_kernel<PERSIST><<<grid, 512, LDS_TOTAL, st>>>(p, ntiles);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) lab2_kernel<PERSIST><<<grid, 512, LDS_TOTAL, st>>>(p, ntiles);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const int M = 16384;
#ifdef LAB_TINY_EPI
    const int shapes[4][2] = {{8064, 768}, {49920, 768}, {3072, 768}, {768, 768}};
#else
    const int shapes[4][2] = {{3072, 768}, {2304, 768}, {768, 768}, {768, 3072}};
#endif
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int s = 0; s < 4; ++s) {
        const int N = shapes[s][0], K = shapes[s][1];
        std::vector<u16> a, w;
        fill(a, (size_t)M * K, 1, 1.0f);
        fill(w, (size_t)N * K, 2, 0.05f);
        u16 *da, *dw;
        float *c0, *c1;
        CK(hipMalloc(&da, a.size() * 2)); CK(hipMalloc(&dw, w.size() * 2));
        CK(hipMalloc(&c0, (size_t)M * N * 4)); CK(hipMalloc(&c1, (size_t)M * N * 4));
        CK(hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice));
        const double fl = 2.0 * M * N * K;
        P p0{M, N, K, N / BNN, da, dw, c0, nullptr};
        const float t0 = run<0>(p0, iters, st);
        printf("N=%4d K=%4d  one tile per workgroup %7.1f us  %6.1f TF\n", N, K, t0, fl / t0 / 1e6);
        P p1{M, N, K, N / BNN, da, dw, c1, nullptr};
        CK(hipMemset(c1, 0, (size_t)M * N * 4));
        const float t1 = run<1>(p1, iters, st);
#ifdef LAB_TINY_EPI
        std::vector<float> r0((size_t)M * (N / BNN)), r1((size_t)M * (N / BNN));
#else
        std::vector<float> r0((size_t)M * N), r1((size_t)M * N);
#endif
        CK(hipMemcpy(r0.data(), c0, r0.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(r1.data(), c1, r1.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < r0.size(); ++i) bad += memcmp(&r0[i], &r1[i], 4) != 0;
        printf("N=%4d K=%4d  persistent (256 workgroups) %7.1f us  %6.1f TF  mismatches %zu\n", N, K, t1, fl / t1 / 1e6, bad);
        const float t0b = run<0>(p0, iters, st), t1b = run<1>(p1, iters, st), t0c = run<0>(p0, iters, st), t1c = run<1>(p1, iters, st);
        printf("            again: one tile %7.1f  persistent %7.1f  one tile %7.1f  persistent %7.1f\n", t0b, t1b, t0c, t1c);
        fflush(stdout);
        CK(hipFree(da)); CK(hipFree(dw)); CK(hipFree(c0)); CK(hipFree(c1));
    }
    return 0;
}
