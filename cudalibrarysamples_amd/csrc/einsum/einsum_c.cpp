// einsum_c.cpp — C entry points over cutensor_amd::Einsum<> so that Python (ctypes) can drive the
// same C++ helper the native samples use.  The reference exposes its helper to Python through
// pybind11 (cuTENSOR/python/cutensor/torch/einsum.cc:75-137); this is the same seam without a
// framework dependency: plain pointers and sizes only.
#include <cstring>
#include <new>

#include "einsum.hpp"
#include "../host/api_guard.hpp"

namespace {
constexpr int kMaxModes = 64;   // torch/einsum.cc:84
struct Base {
    virtual ~Base() {}
    virtual bool init() const = 0;
    virtual std::vector<int64_t> shape() const = 0;
    virtual bool plan(cutensorHandle_t h, uint64_t limit) = 0;
    virtual bool replan(cutensorHandle_t h, uint64_t limit) = 0;
    virtual uint64_t required() const = 0;
    virtual bool exec(cutensorHandle_t h, const void* A, const void* B, void* C, void* w, hipStream_t s) = 0;
    virtual cutensorPlan_t raw() const = 0;
    virtual void conj(bool a, bool b) = 0;
};
template <typename T>
struct Impl : Base {
    cutensor_amd::Einsum<T, int64_t, kMaxModes> e;
    Impl(const char* eq, const std::vector<int64_t>& a, const std::vector<int64_t>& b) : e(eq, a, b) {}
    bool init() const override { return e.isInitialized(); }
    std::vector<int64_t> shape() const override { return e.getOutputShape(); }
    bool plan(cutensorHandle_t h, uint64_t limit) override { return e.plan(h, limit); }
    bool replan(cutensorHandle_t h, uint64_t limit) override { return e.replan(h, limit); }
    uint64_t required() const override { return e.requiredWorkspace(); }
    bool exec(cutensorHandle_t h, const void* A, const void* B, void* C, void* w, hipStream_t s) override {
        return e.execute(h, A, B, C, w, s);
    }
    cutensorPlan_t raw() const override { return e.rawPlan(); }
    void conj(bool a, bool b) override { e.setConjugate(a, b); }
};
}  // namespace

extern "C" {

void* ctamdEinsumCreate(const char* equation, const int64_t* shapeA, int nA, const int64_t* shapeB, int nB, int dtype) try {
    if (equation == nullptr || nA < 0 || nB < 0) return nullptr;
    std::vector<int64_t> a(shapeA, shapeA + nA), b(shapeB, shapeB + nB);
    switch (dtype) {
        case HIP_R_32F:  return static_cast<Base*>(new (std::nothrow) Impl<float>(equation, a, b));
        case HIP_R_64F:  return static_cast<Base*>(new (std::nothrow) Impl<double>(equation, a, b));
        case HIP_R_16F:  return static_cast<Base*>(new (std::nothrow) Impl<__half>(equation, a, b));
        case HIP_R_16BF: return static_cast<Base*>(new (std::nothrow) Impl<__hip_bfloat16>(equation, a, b));
        case HIP_C_32F:  return static_cast<Base*>(new (std::nothrow) Impl<std::complex<float>>(equation, a, b));
        case HIP_C_64F:  return static_cast<Base*>(new (std::nothrow) Impl<std::complex<double>>(equation, a, b));
        default: return nullptr;
    }
} CTAMD_API_CATCH_NULL
void ctamdEinsumDestroy(void* e) try { delete static_cast<Base*>(e); } CTAMD_API_CATCH_VOID
void ctamdEinsumSetConjugate(void* e, int conjA, int conjB) try { if (e) static_cast<Base*>(e)->conj(conjA != 0, conjB != 0); } CTAMD_API_CATCH_VOID
int ctamdEinsumIsInitialized(void* e) try { return e && static_cast<Base*>(e)->init() ? 1 : 0; } CTAMD_API_CATCH_INT
int ctamdEinsumOutputShape(void* e, int64_t* out, int cap) try {
    if (!e) return -1;
    const std::vector<int64_t> s = static_cast<Base*>(e)->shape();
    for (size_t i = 0; i < s.size() && (int)i < cap; ++i) out[i] = s[i];
    return (int)s.size();
} CTAMD_API_CATCH_INT
int ctamdEinsumPlan(void* e, cutensorHandle_t h, uint64_t limit, uint64_t* required) try {
    if (!e || !static_cast<Base*>(e)->plan(h, limit)) return 0;
    if (required) *required = static_cast<Base*>(e)->required();
    return 1;
} CTAMD_API_CATCH_INT
// einsum.cc:104-123: plan again under a smaller workspace limit (0 = CUTENSOR_WORKSPACE_MIN) after an allocation failure
int ctamdEinsumReplan(void* e, cutensorHandle_t h, uint64_t limit, uint64_t* required) try {
    if (!e || !static_cast<Base*>(e)->replan(h, limit)) return 0;
    if (required) *required = static_cast<Base*>(e)->required();
    return 1;
} CTAMD_API_CATCH_INT
int ctamdEinsumExecute(void* e, cutensorHandle_t h, const void* A, const void* B, void* C, void* work, hipStream_t stream) try {
    return (e && static_cast<Base*>(e)->exec(h, A, B, C, work, stream)) ? 1 : 0;
} CTAMD_API_CATCH_INT
cutensorPlan_t ctamdEinsumRawPlan(void* e) try { return e ? static_cast<Base*>(e)->raw() : nullptr; } CTAMD_API_CATCH_NULL

}  // extern "C"
