// Stand-alone test bed for main-loop structures of the ONE-product (bf16 / fp16) 256x192 GEMM tile.
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

template <int V>
__global__ void __launch_bounds__(512, 2) lab_kernel(P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / p.tiles_n) * BMM, n0 = (tile % p.tiles_n) * BNN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lg = lane >> 4;
    const int grp = wave >> 2, w4 = wave & 3;  // waves w and w+4 share a SIMD

    const unsigned long long t_start = (V == 9) ? __builtin_amdgcn_s_memtime() : 0;
    unsigned long long t_loop0 = 0, t_loop1 = 0;
    f32x4 acc[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // DMA piece q (1 KiB at stage offset q*1024) = 8 rows x 128 B; lane -> row q*8 + lane/8, slot lane%8
    // V0: wave issues q = wave + 8j (j < 7).  V5: group X issues q = w4 + 4j (j < 14).
    // V8: group Y issues the A pieces q = w4 + 4j (j < 8), group X the W pieces q = 32 + w4 + 4j (j < 6).
    constexpr int NP = V == 0 ? 7 : (V == 5 ? 14 : 8);
    constexpr int SPL = V == 10 ? 6 : (V == 11 ? 2 : (V == 12 ? 5 : 4));  // weight fragments of k-step 0 multiplied in slot B1
    const u16* src[NP];
    int qoff[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        int q;
        if (V == 0) q = wave + 8 * j;
        else if (V == 5) q = w4 + 4 * j;
        else q = grp ? w4 + 4 * j : NA + w4 + 4 * (j < 6 ? j : 5);
        const bool isA = q < NA;
        const int row = (isA ? q : q - NA) * 8 + lane / 8;
        const int slot = lane % 8;
        const int g = slot ^ (row & 7);
        src[j] = (isA ? p.A : p.W) + (size_t)min((isA ? m0 : n0) + row, (isA ? p.M : p.N) - 1) * p.K + g * 8;
        qoff[j] = q * 1024;
    }
    const int nt = p.K / BKK;
#define ISSUE_N(T, BUF, N)                                                                         \
    do {                                                                                           \
        char* st_ = smem + (BUF) * STAGE;                                                          \
        _Pragma("unroll") for (int j = 0; j < (N); ++j) glds16(src[j] + (T) * BKK, st_ + qoff[j]); \
    } while (0)

    bf16x8 wf[2][6], af[2][4];

    if constexpr (V == 0) {
        ISSUE_N(0, 0, 7);
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + 1 < nt) ISSUE_N(t + 1, buf ^ 1, 7);
            const char* st = smem + buf * STAGE;
            READ_FRAGS(st);
            MFMA_RANGE(0, 0, 6)
            MFMA_RANGE(1, 0, 6)
        }
    } else if constexpr (V == 5) {
        if (grp == 0) {
            ISSUE_N(0, 0, 14);
            if (nt > 1) ISSUE_N(1, 1, 14);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (grp == 0) {
            for (int t = 0; t < nt; ++t) {
                const char* st = smem + (t & 1) * STAGE;
                READ_FRAGS(st);
                if (t >= 1 && t + 1 < nt) ISSUE_N(t + 1, (t + 1) & 1, 14);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
                MFMA_RANGE(0, 0, 6)
                MFMA_RANGE(1, 0, 6)
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        } else {
            __builtin_amdgcn_s_barrier();
            for (int t = 0; t < nt; ++t) {
                const char* st = smem + (t & 1) * STAGE;
                READ_FRAGS(st);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
                MFMA_RANGE(0, 0, 6)
                MFMA_RANGE(1, 0, 6)
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < nt) __builtin_amdgcn_s_barrier();
            }
        }
    } else {
        // ---- V8.  Slots of iteration i (every wave executes the three closing barriers):
        //   A_i : X reads tile i, then issues W(i+1)          | Y multiplies tile i-1
        //   B1_i: X multiplies tile i (first 16 MFMAs)         | Y reads tile i          -> buffer i&1 free
        //   B2_i: X multiplies tile i (other 32), waits W(i+1) | Y issues A(i+2) into buffer i&1, waits A(i+1)
        unsigned long long sA = 0, sB = 0, sC = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        if (grp == 0) { ISSUE_N(0, 0, 6); if (nt > 1) ISSUE_N(1, 1, 6); }
        else { ISSUE_N(0, 0, 8); if (nt > 1) ISSUE_N(1, 1, 8); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (V == 9) t_loop0 = __builtin_amdgcn_s_memtime();
        if (grp == 0) {
            for (int i = 0; i < nt; ++i) {
                if (V == 9) c0 = __builtin_amdgcn_s_memtime();
                const char* st = smem + (i & 1) * STAGE;
                READ_FRAGS(st);
                if (i >= 1 && i + 1 < nt) ISSUE_N(i + 1, (i + 1) & 1, 6);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();  // end A_i
                if (V == 9) c1 = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_s_setprio(1);
                MFMA_RANGE(0, 0, SPL)
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();  // end B1_i
                if (V == 9) c2 = __builtin_amdgcn_s_memtime();
                MFMA_RANGE(0, SPL, 6)
                MFMA_RANGE(1, 0, 6)
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();  // end B2_i
                if (V == 9) { c3 = __builtin_amdgcn_s_memtime(); sA += c1 - c0; sB += c2 - c1; sC += c3 - c2; }
            }
        } else {
            for (int i = 0; i < nt; ++i) {
                if (V == 9) c0 = __builtin_amdgcn_s_memtime();
                if (i >= 1) {
                    __builtin_amdgcn_s_setprio(1);
                    MFMA_RANGE(0, 0, 6)
                    MFMA_RANGE(1, 0, 6)
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_barrier();  // end A_i
                if (V == 9) c1 = __builtin_amdgcn_s_memtime();
                const char* st = smem + (i & 1) * STAGE;
                READ_FRAGS(st);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();  // end B1_i
                if (V == 9) c2 = __builtin_amdgcn_s_memtime();
                if (i + 2 < nt) {
                    ISSUE_N(i + 2, i & 1, 8);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();  // end B2_i
                if (V == 9) { c3 = __builtin_amdgcn_s_memtime(); sA += c1 - c0; sB += c2 - c1; sC += c3 - c2; }
            }
            MFMA_RANGE(0, 0, 6)
            MFMA_RANGE(1, 0, 6)
        }
        if (V == 9) t_loop1 = __builtin_amdgcn_s_memtime();
        if (V == 9 && lane == 0 && w4 == 0) {
            unsigned long long* o = p.stamps + (size_t)blockIdx.x * 12 + grp * 6;
            o[0] = sA; o[1] = sB; o[2] = sC; o[3] = t_loop0 - t_start; o[4] = t_loop1 - t_loop0;
        }
    }

#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + lr;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < 6; ++ni) {
            const int n = n0 + wn * 96 + ni * 16 + lg * 4;
            if (n >= p.N) continue;
            *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.N + n) = acc[ni][mi];
        }
    }
    if (V == 9) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stores acknowledged
        if (lane == 0 && w4 == 0) p.stamps[(size_t)blockIdx.x * 12 + grp * 6 + 5] = __builtin_amdgcn_s_memtime() - t_loop1;
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

template <int V>
static float run(const P& p, int iters, hipStream_t st) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
    const int grid = (p.M + BMM - 1) / BMM * p.tiles_n;
    for (int i = 0; i < 3; ++i) lab_kernel<V><<<grid, 512, LDS_TOTAL, st>>>(p);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) lab_kernel<V><<<grid, 512, LDS_TOTAL, st>>>(p);
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
    const int shapes[4][2] = {{768, 3072}, {3072, 768}, {2304, 768}, {768, 768}};
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int s = 0; s < 4; ++s) {
        const int N = shapes[s][0], K = shapes[s][1];
        std::vector<u16> a, w;
        fill(a, (size_t)M * K, 1, 1.0f);
        fill(w, (size_t)N * K, 2, 0.05f);
        u16 *da, *dw;
        float *c0, *c1;
        unsigned long long* stamps;
        CK(hipMalloc(&da, a.size() * 2)); CK(hipMalloc(&dw, w.size() * 2));
        CK(hipMalloc(&c0, (size_t)M * N * 4)); CK(hipMalloc(&c1, (size_t)M * N * 4));
        const int grid = M / BMM * (N / BNN);
        CK(hipMalloc(&stamps, grid * 12 * 8));
        CK(hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice));
        P p{M, N, K, N / BNN, da, dw, c0, stamps};
        const double fl = 2.0 * M * N * K;
        std::vector<float> ref((size_t)M * N), out((size_t)M * N);
        const float t0 = run<0>(p, iters, st);
        CK(hipMemcpy(ref.data(), c0, ref.size() * 4, hipMemcpyDeviceToHost));
        printf("N=%4d K=%4d  v0 library loop        %7.1f us  %6.1f TF\n", N, K, t0, fl / t0 / 1e6);
        p.C = c1;
        auto check = [&](const char* name, float us) {
            CK(hipMemcpy(out.data(), c1, out.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < out.size(); ++i) bad += memcmp(&out[i], &ref[i], 4) != 0;
            printf("N=%4d K=%4d  %-22s %7.1f us  %6.1f TF  mismatches %zu\n", N, K, name, us, fl / us / 1e6, bad);
            CK(hipMemset(c1, 0, out.size() * 4));
        };
        check("v5 staggered, X issues", run<5>(p, iters, st));
        check("v8 3-slot staggered", run<8>(p, iters, st));
        check("v9 = v8 + stamps", run<9>(p, iters, st));
        check("v10 B1 = 24 MFMAs", run<10>(p, iters, st));
        check("v11 B1 = 8 MFMAs", run<11>(p, iters, st));
        check("v12 B1 = 20 MFMAs", run<12>(p, iters, st));
        check("v8 again", run<8>(p, iters, st));
        {
            std::vector<unsigned long long> h(grid * 12);
            CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
            double x[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < grid; ++i) for (int j = 0; j < 12; ++j) x[j] += (double)h[i * 12 + j];
            const double nt = K / BKK;
            printf("      v9 per K-tile: X  A %.0f  B1 %.0f  B2 %.0f | Y  A %.0f  B1 %.0f  B2 %.0f\n", x[0] / grid / nt, x[1] / grid / nt,
                   x[2] / grid / nt, x[6] / grid / nt, x[7] / grid / nt, x[8] / grid / nt);
            printf("      v9 per workgroup (ticks): X prologue %.0f loop %.0f epilogue %.0f | Y prologue %.0f loop %.0f epilogue %.0f\n",
                   x[3] / grid, x[4] / grid, x[5] / grid, x[9] / grid, x[10] / grid, x[11] / grid);
        }
        fflush(stdout);
        CK(hipFree(da)); CK(hipFree(dw)); CK(hipFree(c0)); CK(hipFree(c1)); CK(hipFree(stamps));
    }
    return 0;
}
