// HIP runtime call checking (the reference's CHECK_CUDA, util.h:27-34).
#pragma once

#include <hip/hip_runtime.h>

#include <mutex>
#include <set>
#include <sstream>
#include <utility>

#include "common.hpp"

namespace cu2rec {

inline void hip_check(hipError_t code, const char *expr, const char *file, int line) {
    if (code == hipSuccess) return;
    std::ostringstream msg;
    msg << "HIP error: " << hipGetErrorString(code) << " in `" << expr << "` (" << file << ":" << line << ")";
    const bool no_dev = code == hipErrorNoDevice || code == hipErrorInvalidDevice || code == hipErrorInsufficientDriver;
    throw Error(no_dev ? CU2REC_ENODEVICE : CU2REC_EHIP, msg.str());
}

#define CU2REC_HIP(expr) ::cu2rec::hip_check((expr), #expr, __FILE__, __LINE__)

// The hot path has no CPU fallback: fail loudly when no GPU is usable.
inline void require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        fail(CU2REC_ENODEVICE, "cu2rec_amd: no HIP device available; the SGD/loss path is GPU-only (no CPU fallback)");
    }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device): set once for each pair, whichever device
// is current when a launch needs it (a process-wide "done" flag would leave the second device of a process at 64 KB).
inline void ensure_max_dynamic_lds(const void *kernel, int bytes = 160 * 1024) {
    static std::mutex mutex;
    static std::set<std::pair<int, const void *>> done;
    int dev = 0;
    CU2REC_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mutex);
    if (done.count({dev, kernel})) return;
    CU2REC_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({dev, kernel});
}

template <class T>
struct DeviceBuffer {  // RAII hipMalloc, the role of CudaDenseMatrix / CudaCSRMatrix members (matrix.cu:12-46)
    T *ptr = nullptr;
    size_t count = 0;
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t n) { allocate(n); }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer() { release(); }
    void allocate(size_t n) {
        release();
        count = n;
        if (n) CU2REC_HIP(hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T)));
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
    void upload(const T *host, size_t n) { CU2REC_HIP(hipMemcpy(ptr, host, n * sizeof(T), hipMemcpyHostToDevice)); }
    void download(T *host, size_t n) const { CU2REC_HIP(hipMemcpy(host, ptr, n * sizeof(T), hipMemcpyDeviceToHost)); }
    void zero() {
        if (count) CU2REC_HIP(hipMemset(ptr, 0, count * sizeof(T)));
    }
    void swap(DeviceBuffer &other) {
        std::swap(ptr, other.ptr);
        std::swap(count, other.count);
    }
};

}  // namespace cu2rec
