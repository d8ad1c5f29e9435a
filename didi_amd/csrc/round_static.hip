// Rounding without a caller workspace and without any allocation (no counterpart in the reference).
#include <mutex>

#include "common.hpp"
// didi_round_argmax_static: didi_round_argmax with the (max, index) partials of its two launches kept in a file-scope device
// array instead of a caller workspace, for callers that cannot allocate (code under HIP graph capture, serving loops and
// harnesses that forbid hipMalloc).  Every GPU the library runs on holds its own copy of the array, at its own address, until
// the process ends; in host code the array's NAME is only a stand-in (passing it to a kernel would fault), so the device
// address is looked up once per device with hipGetSymbolAddress and only that pointer is used afterwards.  The partials need
// no clearing: the GEMM launch writes every slot the reduction reads.  Calls that overlap on one GPU share the array and race.
#define DIDI_ROUND_STATIC_BYTES (16u << 20)
static __device__ __align__(256) unsigned char scratch[DIDI_ROUND_STATIC_BYTES];

static void* scratch_ptr_of_current_device() {
    constexpr int MAXDEV = 64;
    static void* ptrs[MAXDEV];
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!ptrs[dev]) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(scratch)) != hipSuccess) return nullptr;
        ptrs[dev] = q;
    }
    return ptrs[dev];
}

extern "C" size_t didi_round_argmax_static_max_bytes(void) { return DIDI_ROUND_STATIC_BYTES; }

extern "C" int didi_round_argmax_static(const uint16_t* X_hi, const uint16_t* X_lo, int ldx, const uint16_t* W_hi,
                                        const uint16_t* W_lo, int ldw, int M, int V, int K, int precision, int64_t* ids64,
                                        int32_t* ids32, void* stream) {
    DIDI_CHECK_ARG(ids64 || ids32);
    DIDI_CHECK_ARG(M >= 0 && V > 0);
    if (didi_round_argmax_bytes(M, V) > DIDI_ROUND_STATIC_BYTES) return DIDI_ERR_WORKSPACE;  // M * ceil(V / 128) * 8 bytes
    void* partial = scratch_ptr_of_current_device();
    if (!partial) return DIDI_ERR_LAUNCH;
    return didi_round_impl(X_hi, X_lo, ldx, W_hi, W_lo, ldw, M, V, K, precision, partial, ids64, ids32, nullptr, nullptr, stream);
}
