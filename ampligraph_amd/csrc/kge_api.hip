// libamdkge: error plumbing and the device/stream helpers of the C ABI (include/amdkge.h).
#include <stdio.h>
#include <string.h>

#include "kge_host.h"

namespace kge {

static thread_local char g_err[512] = "";

int set_error(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
    return code;
}

int set_error_hip(hipError_t e, const char* where) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? AMDKGE_ENOMEM : AMDKGE_EHIP;
}

int check_launch(const char* kernel_name) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error_hip(e, kernel_name);
    return AMDKGE_OK;
}

}  // namespace kge

using namespace kge;

extern "C" int amdkge_abi_version(void) { return AMDKGE_ABI_VERSION; }

extern "C" const char* amdkge_last_error(void) { return g_err; }

extern "C" int amdkge_device_count(int* count) {
    if (!count) return set_error(AMDKGE_EINVAL, "device_count: NULL output");
    const hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) { *count = 0; return set_error_hip(e, "hipGetDeviceCount"); }
    return AMDKGE_OK;
}

extern "C" int amdkge_set_device(int device) {
    const hipError_t e = hipSetDevice(device);
    return e == hipSuccess ? AMDKGE_OK : set_error_hip(e, "hipSetDevice");
}

extern "C" int amdkge_dev_alloc(void** d_ptr, uint64_t bytes) {
    if (!d_ptr) return set_error(AMDKGE_EINVAL, "dev_alloc: NULL output");
    *d_ptr = nullptr;
    if (bytes == 0) return AMDKGE_OK;
    const hipError_t e = hipMalloc(d_ptr, bytes);
    return e == hipSuccess ? AMDKGE_OK : set_error_hip(e, "hipMalloc");
}

extern "C" int amdkge_dev_free(void* d_ptr) {
    if (!d_ptr) return AMDKGE_OK;
    const hipError_t e = hipFree(d_ptr);
    return e == hipSuccess ? AMDKGE_OK : set_error_hip(e, "hipFree");
}

extern "C" int amdkge_h2d(void* d_dst, const void* src, uint64_t bytes, void* stream) {
    if (bytes == 0) return AMDKGE_OK;
    if (!d_dst || !src) return set_error(AMDKGE_EINVAL, "h2d: NULL pointer");
    const hipError_t e = hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    return e == hipSuccess ? AMDKGE_OK : set_error_hip(e, "hipMemcpyAsync(H2D)");
}

extern "C" int amdkge_d2h(void* dst, const void* d_src, uint64_t bytes, void* stream) {
    if (bytes == 0) return AMDKGE_OK;
    if (!dst || !d_src) return set_error(AMDKGE_EINVAL, "d2h: NULL pointer");
    hipError_t e = hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? AMDKGE_OK : set_error_hip(e, "hipMemcpyAsync(D2H)");
}

extern "C" int amdkge_dev_memset(void* d_ptr, int value, uint64_t bytes, void* stream) {
    if (bytes == 0) return AMDKGE_OK;
    if (!d_ptr) return set_error(AMDKGE_EINVAL, "memset: NULL pointer");
    const hipError_t e = hipMemsetAsync(d_ptr, value, bytes, (hipStream_t)stream);
    return e == hipSuccess ? AMDKGE_OK : set_error_hip(e, "hipMemsetAsync");
}

extern "C" int amdkge_stream_sync(void* stream) {
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? AMDKGE_OK : set_error_hip(e, "hipStreamSynchronize");
}

extern "C" int amdkge_internal_k(int scoring_type, int k) {
    if (scoring_type < AMDKGE_TRANSE || scoring_type > AMDKGE_ROTATE || k <= 0) return set_error(AMDKGE_EINVAL, "internal_k: bad arguments");
    return internal_k_of(scoring_type, k);
}
