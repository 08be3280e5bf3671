// common.hpp — shared helpers of the native HIP sample drivers (samples/*.hip).
//
// The drivers replay the call sequences of the reference's sample programs against the engine's C ABI in plain HIP —
// no CUDA names anywhere — and, unlike the reference samples (which print timings only), check what they computed
// against fp64 host arithmetic.  Counterparts: cuTENSOR/utils.cuh:35-202 (error macro, timer, allocation helpers,
// random fill); data is U(0,1) from a FIXED seed instead of the samples' nondeterministic one (utils.cuh:76-89).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include <cutensor.h>

#define HIP_OK(x)                                                                                       \
    do {                                                                                                \
        const hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } \
    } while (0)
#define CT_OK(x)                                                                                        \
    do {                                                                                                \
        const cutensorStatus_t s_ = (x);                                                                \
        if (s_ != CUTENSOR_STATUS_SUCCESS) { std::printf("cuTENSOR error %s at %s:%d\n", cutensorGetErrorString(s_), __FILE__, __LINE__); std::exit(3); } \
    } while (0)

namespace sample {

template <typename T>
struct DeviceBuffer {           // cuda_alloc<T> of utils.cuh:50-59
    T* p = nullptr;
    size_t n = 0;
    explicit DeviceBuffer(size_t count) : n(count) { if (n) HIP_OK(hipMalloc((void**)&p, n * sizeof(T))); }
    ~DeviceBuffer() { if (p) (void)hipFree(p); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    void upload(const std::vector<T>& h) { HIP_OK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }
    std::vector<T> download() const { std::vector<T> h(n); if (n) HIP_OK(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost)); return h; }
};

struct GpuTimer {               // utils.cuh:163-202: an event pair on the execution stream
    hipEvent_t a, b;
    hipStream_t s;
    explicit GpuTimer(hipStream_t stream = nullptr) : s(stream) { HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b)); }
    ~GpuTimer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
    void start() { HIP_OK(hipEventRecord(a, s)); }
    double seconds() { HIP_OK(hipEventRecord(b, s)); HIP_OK(hipEventSynchronize(b)); float ms = 0; HIP_OK(hipEventElapsedTime(&ms, a, b)); return ms * 1e-3; }
};

inline std::vector<float> uniform(size_t n, uint32_t seed) {
    std::mt19937 gen(seed);
    std::uniform_real_distribution<float> d(0.f, 1.f);
    std::vector<float> v(n);
    for (auto& x : v) x = d(gen);
    return v;
}

inline int64_t product(const std::vector<int64_t>& e) { int64_t p = 1; for (int64_t x : e) p *= x; return p; }

// packed generalized column-major strides (first mode fastest; blocksparse.cu:80-81)
inline std::vector<int64_t> packed_strides(const std::vector<int64_t>& e) {
    std::vector<int64_t> s(e.size());
    int64_t run = 1;
    for (size_t i = 0; i < e.size(); ++i) { s[i] = run; run *= e[i]; }
    return s;
}

inline int arg_int(int argc, char** argv, const char* name, int def) {
    for (int i = 1; i + 1 < argc; ++i) if (std::string(argv[i]) == name) return std::atoi(argv[i + 1]);
    return def;
}
inline bool arg_flag(int argc, char** argv, const char* name) {
    for (int i = 1; i < argc; ++i) if (std::string(argv[i]) == name) return true;
    return false;
}

}  // namespace sample
