// Host-only stand-in for <hip/hip_runtime.h> used by ONE thing: the sanitizer build of the C-ABI's host code
// (tests/test_abi_and_host.py::test_host_code_under_address_and_ub_sanitizers).  "Device" memory is host memory from a
// registry that knows every block's size, so a copy / memset / kernel-argument range outside a live block is reported;
// kernels are no-ops (tests/host_san/gen_stubs.py).  Never part of the product build.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <stdio.h>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct hipStreamStub* hipStream_t;
typedef struct hipEventStub* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };

inline std::map<const char*, size_t>& hip_stub_blocks() { static std::map<const char*, size_t> m; return m; }
inline size_t& hip_stub_limit() { static size_t v = (size_t)6 << 30; return v; }     // largest single allocation served
// a device range must lie inside ONE live block (host pointers -- e.g. the D2H destination on the stack -- are not checked)
inline bool hip_stub_range_ok(const void* p, size_t n, bool must_be_device) {
    auto& m = hip_stub_blocks();
    auto it = m.upper_bound((const char*)p);
    if (it != m.begin()) {
        --it;
        if ((const char*)p >= it->first && (const char*)p < it->first + it->second) return (const char*)p + n <= it->first + it->second;
    }
    return !must_be_device;
}
inline hipError_t hipMalloc(void** p, size_t n) {
    if (n > hip_stub_limit()) { *p = nullptr; return hipErrorOutOfMemory; }
    *p = malloc(n ? n : 1);                 // NOT zeroed: reads of never-written "device" memory by host code show up under MSan-like checks
    if (!*p) return hipErrorOutOfMemory;
    memset(*p, 0x5a, n < 4096 ? n : 4096);
    hip_stub_blocks()[(const char*)*p] = n ? n : 1;
    return hipSuccess;
}
inline hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    auto it = hip_stub_blocks().find((const char*)p);
    if (it == hip_stub_blocks().end()) { fprintf(stderr, "hip stub: hipFree of an unknown pointer\n"); abort(); }
    hip_stub_blocks().erase(it);
    free(p);
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) {
    if ((k != hipMemcpyHostToDevice && !hip_stub_range_ok(s, n, true)) || (k != hipMemcpyDeviceToHost && !hip_stub_range_ok(d, n, true))) {
        fprintf(stderr, "hip stub: hipMemcpyAsync range outside its device block (%zu bytes)\n", n); abort();
    }
    memmove(d, s, n);
    return hipSuccess;
}
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k) { return hipMemcpyAsync(d, s, n, k, nullptr); }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
    if (h && (!hip_stub_range_ok(s, (h - 1) * sp + w, true) || !hip_stub_range_ok(d, (h - 1) * dp + w, true))) {
        fprintf(stderr, "hip stub: hipMemcpy2DAsync range outside its device block\n"); abort();
    }
    return hipSuccess;                        // (contents irrelevant: kernels are no-ops)
}
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
    if (!hip_stub_range_ok(d, n, true)) { fprintf(stderr, "hip stub: hipMemsetAsync range outside its device block (%zu bytes)\n", n); abort(); }
    memset(d, v, n < ((size_t)64 << 20) ? n : ((size_t)64 << 20));
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : e == hipErrorOutOfMemory ? "out of memory" : "invalid value"; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
