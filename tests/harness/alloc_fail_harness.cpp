// alloc_fail_harness.cpp — "no exception crosses the C ABI" (csrc/host/api_guard.hpp), checked with a throwing operator new.
// The replaced global operator new of this executable is the one libcutensor.so's std::vector / std::map / std::string use (the
// library does not bind the symbol locally).  For k = 0, 1, 2, ... the k-th allocation of the call sequence of contraction.cu:123-235
// (handle, descriptors, contraction, preference, workspace estimate, plan, plan attribute) throws std::bad_alloc; every entry point
// must answer with a status — CUTENSOR_STATUS_ALLOC_FAILED — instead of letting the exception unwind into this C-style caller
// (std::terminate: the process would abort).  The sweep ends with the first k at which the whole sequence runs without a failure.
// No GPU needed: planning works on a handle without a device.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <new>

#include <cutensor.h>
#include <cutensorMg.h>

static long g_countdown = -1;   // < 0: never fail; otherwise the allocation that finds 0 here throws
static long g_thrown = 0;

static void* guarded_alloc(std::size_t n) {
    if (g_countdown == 0) { g_countdown = -1; ++g_thrown; throw std::bad_alloc(); }
    if (g_countdown > 0) --g_countdown;
    void* p = std::malloc(n ? n : 1);
    if (p == nullptr) throw std::bad_alloc();
    return p;
}
void* operator new(std::size_t n) { return guarded_alloc(n); }
void* operator new[](std::size_t n) { return guarded_alloc(n); }
void* operator new(std::size_t n, const std::nothrow_t&) noexcept { try { return guarded_alloc(n); } catch (...) { return nullptr; } }
void* operator new[](std::size_t n, const std::nothrow_t&) noexcept { try { return guarded_alloc(n); } catch (...) { return nullptr; } }
void operator delete(void* p) noexcept { std::free(p); }
void operator delete[](void* p) noexcept { std::free(p); }
void operator delete(void* p, std::size_t) noexcept { std::free(p); }
void operator delete[](void* p, std::size_t) noexcept { std::free(p); }

struct Outcome { int failed = 0, allocFailed = 0, other = 0; };

static bool ok(cutensorStatus_t s, Outcome& o) {
    if (s == CUTENSOR_STATUS_SUCCESS) return true;
    ++o.failed;
    if (s == CUTENSOR_STATUS_ALLOC_FAILED) ++o.allocFailed; else ++o.other;
    return false;
}

// the call sequence of contraction.cu:123-235 on a small problem; returns true when every call succeeded
static bool sequence(Outcome& o) {
    cutensorHandle_t h = nullptr;
    cutensorTensorDescriptor_t dA = nullptr, dB = nullptr, dC = nullptr;
    cutensorOperationDescriptor_t op = nullptr;
    cutensorPlanPreference_t pref = nullptr;
    cutensorPlan_t plan = nullptr;
    const int64_t eA[4] = {48, 16, 16, 48}, eB[4] = {32, 16, 32, 16}, eC[4] = {48, 32, 48, 32};
    const int32_t mA[4] = {'m', 'h', 'k', 'n'}, mB[4] = {'u', 'k', 'v', 'h'}, mC[4] = {'m', 'u', 'n', 'v'};
    bool good = ok(cutensorCreate(&h), o);
    good = good && ok(cutensorCreateTensorDescriptor(h, &dA, 4, eA, nullptr, HIP_R_32F, 128), o);
    good = good && ok(cutensorCreateTensorDescriptor(h, &dB, 4, eB, nullptr, HIP_R_32F, 128), o);
    good = good && ok(cutensorCreateTensorDescriptor(h, &dC, 4, eC, nullptr, HIP_R_32F, 128), o);
    good = good && ok(cutensorCreateContraction(h, &op, dA, mA, CUTENSOR_OP_IDENTITY, dB, mB, CUTENSOR_OP_IDENTITY, dC, mC,
                                                CUTENSOR_OP_IDENTITY, dC, mC, CUTENSOR_COMPUTE_DESC_32F), o);
    good = good && ok(cutensorCreatePlanPreference(h, &pref, CUTENSOR_ALGO_DEFAULT, CUTENSOR_JIT_MODE_NONE), o);
    uint64_t est = 0, req = 0;
    good = good && ok(cutensorEstimateWorkspaceSize(h, op, pref, CUTENSOR_WORKSPACE_DEFAULT, &est), o);
    good = good && ok(cutensorCreatePlan(h, &plan, op, pref, est), o);
    good = good && ok(cutensorPlanGetAttribute(h, plan, CUTENSOR_PLAN_REQUIRED_WORKSPACE, &req, sizeof(req)), o);
    // the destroy calls tolerate whatever was not created
    cutensorDestroyPlan(plan);
    cutensorDestroyPlanPreference(pref);
    cutensorDestroyOperationDescriptor(op);
    cutensorDestroyTensorDescriptor(dA);
    cutensorDestroyTensorDescriptor(dB);
    cutensorDestroyTensorDescriptor(dC);
    cutensorDestroy(h);
    return good;
}

// the call sequence of contraction_multi_gpu.cu:151-250 for C[i,j] = A[i,k] B[k,j] on two device ids (plan only: nothing executes)
static bool sequence_mg(Outcome& o) {
    cutensorMgHandle_t h = nullptr;
    cutensorMgTensorDescriptor_t dA = nullptr, dB = nullptr, dC = nullptr;
    cutensorMgContractionDescriptor_t con = nullptr;
    cutensorMgContractionFind_t find = nullptr;
    cutensorMgContractionPlan_t plan = nullptr;
    const int32_t devs[2] = {0, 1};
    const int64_t ext[2] = {256, 256}, blkRow[2] = {128, 256}, blkCol[2] = {256, 128}, blkC[2] = {128, 128};
    const int32_t cntRow[2] = {2, 1}, cntCol[2] = {1, 2};
    const int32_t mA[2] = {'i', 'k'}, mB[2] = {'k', 'j'}, mC[2] = {'i', 'j'};
    bool good = ok(cutensorMgCreate(&h, 2, devs), o);
    good = good && ok(cutensorMgCreateTensorDescriptor(h, &dA, 2, ext, nullptr, blkRow, nullptr, cntRow, 2, devs, HIP_R_32F), o);
    good = good && ok(cutensorMgCreateTensorDescriptor(h, &dB, 2, ext, nullptr, blkCol, nullptr, cntCol, 2, devs, HIP_R_32F), o);
    good = good && ok(cutensorMgCreateTensorDescriptor(h, &dC, 2, ext, nullptr, blkC, nullptr, cntRow, 2, devs, HIP_R_32F), o);
    good = good && ok(cutensorMgCreateContractionDescriptor(h, &con, dA, mA, dB, mB, dC, mC, dC, mC, CUTENSOR_COMPUTE_32F), o);
    good = good && ok(cutensorMgCreateContractionFind(h, &find, CUTENSORMG_ALGO_DEFAULT), o);
    int64_t devWs[2] = {0, 0}, hostWs = 0;
    good = good && ok(cutensorMgContractionGetWorkspace(h, con, find, CUTENSOR_WORKSPACE_DEFAULT, devWs, &hostWs), o);
    good = good && ok(cutensorMgCreateContractionPlan(h, &plan, con, find, devWs, hostWs), o);
    cutensorMgDestroyContractionPlan(plan);
    cutensorMgDestroyContractionFind(find);
    cutensorMgDestroyContractionDescriptor(con);
    cutensorMgDestroyTensorDescriptor(dA);
    cutensorMgDestroyTensorDescriptor(dB);
    cutensorMgDestroyTensorDescriptor(dC);
    cutensorMgDestroy(h);
    return good;
}

static int sweep(const char* what, bool (*seq)(Outcome&)) {
    Outcome warm;
    if (!seq(warm)) { std::printf("%s: the sequence fails without any injected failure\n", what); return 2; }
    long injected = 0, answered = 0;
    for (long k = 0; k < 100000; ++k) {
        Outcome o;
        const long before = g_thrown;
        g_countdown = k;
        const bool good = seq(o);
        g_countdown = -1;
        if (g_thrown == before) {            // the sequence needs fewer than k allocations: the sweep is complete
            if (!good) { std::printf("%s: k = %ld: failure without an injected bad_alloc\n", what, k); return 3; }
            std::printf("%s: alloc guard ok: %ld injected failures, %ld answered with a status\n", what, injected, answered);
            return injected > 0 && answered > 0 ? 0 : 4;
        }
        ++injected;
        if (!good) {
            ++answered;
            if (o.other != 0) { std::printf("%s: k = %ld: a status other than ALLOC_FAILED\n", what, k); return 5; }
        }
        // (good && thrown: the failure hit a nothrow allocation or an optional path the library recovers from)
    }
    std::printf("%s: the sweep did not terminate\n", what);
    return 6;
}

int main(int argc, char** argv) {
    int rc = sweep("cutensor", sequence);
    if (rc == 0 && argc > 1 && argv[1][0] == 'm') rc = sweep("cutensorMg", sequence_mg);
    return rc;
}
