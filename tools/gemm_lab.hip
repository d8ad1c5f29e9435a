// Stand-alone test bed for main-loop structures of the 256x192 bf16x3 GEMM tile (gemm_big_kernel<3,4>).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gemm_lab.hip -o tools/gemm_lab && tools/gemm_lab [iters]
//
// Every variant computes C[M,N] = (A_hi+A_lo)[M,K] * (W_hi+W_lo)[N,K]^T with the three split products
// in the library kernel's accumulation order, so variants must agree with variant 0 bit for bit.
//   0  library loop: one-ahead LDS-DMA, one barrier per K-tile, all waves in phase
//   1  variant 0 with s_memtime stamps (where a K-tile's cycles go)
//   2  two wave groups staggered by half a K-tile: one group's fragment reads and DMA issue run
//      under the other group's MFMAs
//   3  variant 0 + L2 touch-ahead of the activation rows (4-byte LDS-DMA to a dummy area, vmcnt(1))
//   4  variant 2 + touch-ahead
//   5  staggered, and group X alone issues the whole stage's DMA inside its MEM slot (group Y's MFMAs hide it)
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
    const u16 *A_hi, *A_lo, *W_hi, *W_lo;
    float* C;
    unsigned long long* stamps;  // [grid][4] variant 1
    int ta;                      // touch-ahead distance in K-tiles
};

__device__ __forceinline__ int swz64(int row, int chunk) { return row * 64 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 4); }
__device__ __forceinline__ void glds16(const u16* src, char* dst) {
    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)dst, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const u16* src, char* dst) {
    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)dst, 4, 0, 0);
}
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = id & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

constexpr int BMM = 256, BNN = 192, BKK = 32;
constexpr int A_BYTES = BMM * 128, NA = A_BYTES / 1024, NW = 24, STAGE = A_BYTES + NW * 1024, PPW = (NA + NW) / 8;
constexpr int LO_A = A_BYTES / 2, LO_W = 12288, W_OFF = A_BYTES;
constexpr int LDS_TOTAL = 2 * STAGE + 8 * 256;

#define READ_FRAGS(ST)                                                                     \
    do {                                                                                   \
        const char *Ah = (ST), *Al = (ST) + LO_A, *Wh = (ST) + W_OFF, *Wl = (ST) + W_OFF + LO_W; \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                    \
            const int wo = swz64(wn * 96 + i * 16 + lr, lg);                               \
            whf[i] = *reinterpret_cast<const bf16x8*>(Wh + wo);                            \
            wlf[i] = *reinterpret_cast<const bf16x8*>(Wl + wo);                            \
        }                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                    \
            const int ao = swz64(wm * 64 + i * 16 + lr, lg);                               \
            ahf[i] = *reinterpret_cast<const bf16x8*>(Ah + ao);                            \
            alf[i] = *reinterpret_cast<const bf16x8*>(Al + ao);                            \
        }                                                                                  \
    } while (0)

#define MFMA_NI(ni)                                                                                        \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                                     \
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlf[ni], ahf[mi], acc[ni][mi], 0, 0, 0);     \
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whf[ni], alf[mi], acc[ni][mi], 0, 0, 0);     \
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whf[ni], ahf[mi], acc[ni][mi], 0, 0, 0);     \
    }

#define MFMA_ALL()                                        \
    do {                                                  \
        _Pragma("unroll") for (int ni = 0; ni < 6; ++ni) { MFMA_NI(ni) } \
    } while (0)

template <int V>
__global__ void __launch_bounds__(512, 2) lab_kernel(P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / p.tiles_n) * BMM, n0 = (tile % p.tiles_n) * BNN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // staggered variants: waves w and w+4 share a SIMD; group = wave >> 2, and the (wm, wn) map keeps
    // each group a full 2x2 half of the tile's waves
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lg = lane >> 4;
    constexpr bool TOUCH = V == 3 || V == 4;

    f32x4 acc[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int NP = (V == 5 || V == 6) ? 2 * PPW : PPW;  // pieces this wave issues per stage
    constexpr int QS = (V == 5 || V == 6) ? 4 : 8;          // piece stride between consecutive j
    const u16* src[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int q = ((V == 5 || V == 6) ? (wave & 3) : wave) + QS * j;
        const u16* base;
        int row, ld, lim, r0;
        if (q < NA / 2) { base = p.A_hi; row = q * 16; }
        else if (q < NA) { base = p.A_lo; row = (q - NA / 2) * 16; }
        else if (q < NA + 12) { base = p.W_hi; row = (q - NA) * 16; }
        else { base = p.W_lo; row = (q - NA - 12) * 16; }
        if (q < NA) { ld = p.K; lim = p.M - 1; r0 = m0; } else { ld = p.K; lim = p.N - 1; r0 = n0; }
        row += lane / 4;
        const int slot = lane % 4;
        const int g = slot ^ ((0 - (row >> 2)) & 3);
        src[j] = base + (size_t)min(r0 + row, lim) * ld + g * 8;
    }
    // touch-ahead: 512 (plane, row) lines per K-tile, one per thread
    const u16* tsrc = (wave < 4 ? p.A_hi : p.A_lo) + (size_t)min(m0 + (wave & 3) * 64 + lane, p.M - 1) * p.K;
    char* tdst = smem + 2 * STAGE + wave * 256;
    const int nt = p.K / BKK;
#define ISSUE(T, BUF)                                                                             \
    do {                                                                                          \
        char* st_ = smem + (BUF) * STAGE + ((V == 5 || V == 6) ? (wave & 3) : wave) * 1024;                   \
        _Pragma("unroll") for (int j = 0; j < NP; ++j) glds16(src[j] + (T) * BKK, st_ + j * QS * 1024); \
    } while (0)
#define TOUCH_AT(T)                                                     \
    do {                                                                \
        const int tt_ = min((T), nt - 1);                               \
        glds4(tsrc + tt_ * BKK, tdst);                                  \
    } while (0)

    bf16x8 whf[6], wlf[6], ahf[4], alf[4];

    if constexpr (V == 0 || V == 1 || V == 3) {
        unsigned long long s_wait = 0, s_bar = 0, s_issue = 0, s_comp = 0;
        ISSUE(0, 0);
        if constexpr (TOUCH) TOUCH_AT(p.ta);
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
            if constexpr (V == 1) c0 = __builtin_amdgcn_s_memtime();
            if constexpr (TOUCH) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (V == 1) c1 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            if constexpr (V == 1) c2 = __builtin_amdgcn_s_memtime();
            if (t + 1 < nt) ISSUE(t + 1, buf ^ 1);
            if constexpr (TOUCH) TOUCH_AT(t + 1 + p.ta);
            if constexpr (V == 1) { __builtin_amdgcn_sched_barrier(0); c3 = __builtin_amdgcn_s_memtime(); }
            const char* st = smem + buf * STAGE;
            READ_FRAGS(st);
            MFMA_ALL();
            if constexpr (V == 1) {
                __builtin_amdgcn_sched_barrier(0);
                c4 = __builtin_amdgcn_s_memtime();
                s_wait += c1 - c0; s_bar += c2 - c1; s_issue += c3 - c2; s_comp += c4 - c3;
            }
        }
        if constexpr (V == 1) {
            if (tid == 0) {
                p.stamps[blockIdx.x * 4 + 0] = s_wait; p.stamps[blockIdx.x * 4 + 1] = s_bar;
                p.stamps[blockIdx.x * 4 + 2] = s_issue; p.stamps[blockIdx.x * 4 + 3] = s_comp;
            }
        }
    } else if constexpr (V == 5) {
        if (wave < 4) {
            ISSUE(0, 0);
            if (nt > 1) ISSUE(1, 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wave < 4) {
            for (int t = 0; t < nt; ++t) {
                const char* st = smem + (t & 1) * STAGE;
                READ_FRAGS(st);
                if (t >= 1 && t + 1 < nt) ISSUE(t + 1, (t + 1) & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
                MFMA_ALL();
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
                MFMA_ALL();
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < nt) __builtin_amdgcn_s_barrier();
            }
        }
} else if constexpr (V == 6) {
        unsigned long long sA = 0, sB = 0, sC = 0, sD = 0, c0, c1, c2, c3, c4;
        if (wave < 4) {
            ISSUE(0, 0);
            if (nt > 1) ISSUE(1, 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wave < 4) {
            for (int t = 0; t < nt; ++t) {
                const char* st = smem + (t & 1) * STAGE;
                c0 = __builtin_amdgcn_s_memtime();
                READ_FRAGS(st);
                if (t >= 1 && t + 1 < nt) ISSUE(t + 1, (t + 1) & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                c1 = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_s_barrier();
                c2 = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_s_setprio(1);
                MFMA_ALL();
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                c3 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                c4 = __builtin_amdgcn_s_memtime();
                sA += c1 - c0; sB += c2 - c1; sC += c3 - c2; sD += c4 - c3;
            }
        } else {
            __builtin_amdgcn_s_barrier();
            for (int t = 0; t < nt; ++t) {
                const char* st = smem + (t & 1) * STAGE;
                c0 = __builtin_amdgcn_s_memtime();
                READ_FRAGS(st);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                c1 = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_s_barrier();
                c2 = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_s_setprio(1);
                MFMA_ALL();
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                c3 = __builtin_amdgcn_s_memtime();
                if (t + 1 < nt) __builtin_amdgcn_s_barrier();
                c4 = __builtin_amdgcn_s_memtime();
                sA += c1 - c0; sB += c2 - c1; sC += c3 - c2; sD += c4 - c3;
            }
        }
        if (lane == 0 && (wave == 0 || wave == 4)) {
            unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8 + (wave >> 2) * 4;
            o[0] = sA; o[1] = sB; o[2] = sC; o[3] = sD;
        }
    } else {
        // ---- staggered: slot boundaries are barriers every wave executes; group X = waves 0-3 runs
        // MEM(t) in slot 2t and MFMA(t) in slot 2t+1, group Y = waves 4-7 one slot later.
        // DMA(t+2) -> buffer t&1 is issued by everyone in slot 2t+2 (after Y's reads of tile t, slot 2t+1)
        // and waited for by everyone at the end of slot 2t+3; X first reads it in slot 2t+4.
        ISSUE(0, 0);
        if (nt > 1) ISSUE(1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wave < 4) {
            for (int t = 0; t < nt; ++t) {
                // slot 2t: MEM(t)
                if (t >= 1 && t + 1 < nt) ISSUE(t + 1, (t + 1) & 1);
                if constexpr (TOUCH) TOUCH_AT(t + 1 + p.ta);
                const char* st = smem + (t & 1) * STAGE;
                READ_FRAGS(st);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                // slot 2t+1: MFMA(t)
                __builtin_amdgcn_s_setprio(1);
                MFMA_ALL();
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (TOUCH) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        } else {
            __builtin_amdgcn_s_barrier();  // slot 0: idle
            for (int t = 0; t < nt; ++t) {
                // slot 2t+1: MEM(t)
                const char* st = smem + (t & 1) * STAGE;
                READ_FRAGS(st);
                if constexpr (TOUCH) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                // slot 2t+2: MFMA(t), DMA(t+2) issued between the MFMA groups
                if (t + 2 < nt) ISSUE(t + 2, t & 1);
                if constexpr (TOUCH) TOUCH_AT(t + 2 + p.ta);
                __builtin_amdgcn_s_setprio(1);
                MFMA_ALL();
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < nt) __builtin_amdgcn_s_barrier();
            }
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
}

static u16 f2bf(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (u16)(u >> 16);
}
static float bf2f(u16 b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static void fill(std::vector<u16>& hi, std::vector<u16>& lo, size_t n, uint64_t seed, float scale) {
    hi.resize(n); lo.resize(n);
    uint64_t s = seed;
    for (size_t i = 0; i < n; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const float x = scale * ((float)((s >> 40) & 0xffffff) / 8388608.0f - 1.0f);
        hi[i] = f2bf(x);
        lo[i] = f2bf(x - bf2f(hi[i]));
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
        std::vector<u16> ah, al, wh, wl;
        fill(ah, al, (size_t)M * K, 1, 1.0f);
        fill(wh, wl, (size_t)N * K, 2, 0.05f);
        u16 *dah, *dal, *dwh, *dwl;
        float *c0, *c1;
        unsigned long long* stamps;
        CK(hipMalloc(&dah, ah.size() * 2)); CK(hipMalloc(&dal, al.size() * 2));
        CK(hipMalloc(&dwh, wh.size() * 2)); CK(hipMalloc(&dwl, wl.size() * 2));
        CK(hipMalloc(&c0, (size_t)M * N * 4)); CK(hipMalloc(&c1, (size_t)M * N * 4));
        const int grid = M / BMM * (N / BNN);
        CK(hipMalloc(&stamps, grid * 8 * 8));
        CK(hipMemcpy(dah, ah.data(), ah.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dal, al.data(), al.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dwh, wh.data(), wh.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dwl, wl.data(), wl.size() * 2, hipMemcpyHostToDevice));
        P p{M, N, K, N / BNN, dah, dal, dwh, dwl, c0, stamps, 4};
        const double fl = 2.0 * M * N * K;
        std::vector<float> ref((size_t)M * N), out((size_t)M * N);
        const float t0 = run<0>(p, iters, st);
        CK(hipMemcpy(ref.data(), c0, ref.size() * 4, hipMemcpyDeviceToHost));
        printf("N=%4d K=%4d  v0 library loop        %7.1f us  alg %6.1f TF\n", N, K, t0, fl / t0 / 1e6);
        p.C = c1;
        auto check = [&](const char* name, float us) {
            CK(hipMemcpy(out.data(), c1, out.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < out.size(); ++i) bad += memcmp(&out[i], &ref[i], 4) != 0;
            printf("N=%4d K=%4d  %-22s %7.1f us  alg %6.1f TF  mismatches %zu\n", N, K, name, us, fl / us / 1e6, bad);
            CK(hipMemset(c1, 0, out.size() * 4));
        };
        const float t1 = run<1>(p, iters, st);
        check("v1 stamps", t1);
        {
            std::vector<unsigned long long> h(grid * 4);
            CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
            double a[4] = {0, 0, 0, 0};
            for (int i = 0; i < grid; ++i) for (int j = 0; j < 4; ++j) a[j] += (double)h[i * 4 + j];
            const double nt = K / BKK;
            printf("      stamps per K-tile (wave 0, s_memtime ticks): vmcnt wait %.0f  barrier %.0f  DMA issue %.0f  reads+MFMA %.0f\n",
                   a[0] / grid / nt, a[1] / grid / nt, a[2] / grid / nt, a[3] / grid / nt);
        }
        check("v2 staggered", run<2>(p, iters, st));
        check("v5 staggered, X issues", run<5>(p, iters, st));
        check("v6 = v5 + stamps", run<6>(p, iters, st));
        {
            std::vector<unsigned long long> h(grid * 8);
            CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
            double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < grid; ++i) for (int j = 0; j < 8; ++j) a[j] += (double)h[i * 8 + j];
            const double nt = K / BKK;
            printf("      v6 per K-tile: X mem %.0f bar %.0f mfma %.0f vmcnt+bar %.0f | Y mem %.0f bar %.0f mfma %.0f bar %.0f\n",
                   a[0] / grid / nt, a[1] / grid / nt, a[2] / grid / nt, a[3] / grid / nt, a[4] / grid / nt, a[5] / grid / nt, a[6] / grid / nt, a[7] / grid / nt);
        }
        for (int ta : {4}) {
            p.ta = ta;
            char nm[64];
            snprintf(nm, sizeof nm, "v3 touch-ahead %d", ta);
            check(nm, run<3>(p, iters, st));
            snprintf(nm, sizeof nm, "v4 staggered+touch %d", ta);
            check(nm, run<4>(p, iters, st));
        }
        fflush(stdout);
        CK(hipFree(dah)); CK(hipFree(dal)); CK(hipFree(dwh)); CK(hipFree(dwl)); CK(hipFree(c0)); CK(hipFree(c1)); CK(hipFree(stamps));
    }
    return 0;
}
