// einsum.hip — native HIP driver of the einsum helper: the demo calls of cuTENSOR/einsum.cu:441-455 (shapes {2,4,5},{4,8,7},
// equations :447-451) and the headline equation 'abcd,dcbe->ae' (BASELINE configs[1]) through cutensor_amd::Einsum<>
// (csrc/einsum/einsum.hpp, the counterpart of Einsum<> in einsum.cu:57-391), each checked against fp64 host loops.
#include <chrono>
#include <cmath>
#include <functional>

#include "common.hpp"
#include "../cudalibrarysamples_amd/csrc/einsum/einsum.hpp"

using sample::DeviceBuffer;

static bool run(cutensorHandle_t handle, const std::string& eq, const std::vector<int64_t>& sa, const std::vector<int64_t>& sb,
                const std::function<double(const std::vector<float>&, const std::vector<float>&, const std::vector<int64_t>&)>& ref_at,
                int reps) {
    cutensor_amd::Einsum<float, int64_t, 40> e(eq, sa, sb);
    if (!e.isInitialized()) { std::printf("%-16s not supported\n", eq.c_str()); return false; }
    const std::vector<int64_t> so = e.getOutputShape();
    std::vector<float> A = sample::uniform(sample::product(sa), 11), B = sample::uniform(sb.empty() ? 1 : sample::product(sb), 12);
    DeviceBuffer<float> dA(A.size()), dB(B.size()), dC(std::max<int64_t>(sample::product(so), 1));
    DeviceBuffer<char> work(e.getWorksize());
    dA.upload(A); dB.upload(B);
    hipStream_t stream = nullptr;
    if (!e.execute(handle, dA.p, sb.empty() ? nullptr : dB.p, dC.p, work.p, stream)) { std::printf("%-16s execute failed\n", eq.c_str()); return false; }
    HIP_OK(hipDeviceSynchronize());
    double best = 1e100;
    for (int i = 0; i < reps; ++i) {
        sample::GpuTimer t(stream);
        t.start();
        e.execute(handle, dA.p, sb.empty() ? nullptr : dB.p, dC.p, work.p, stream);
        best = std::min(best, t.seconds());
    }
    const std::vector<float> C = dC.download();
    // row-major output: check up to 512 sampled positions
    std::mt19937 gen(3);
    const int64_t total = std::max<int64_t>(sample::product(so), 1);
    double worst = 0.0;
    for (int s = 0; s < 512; ++s) {
        int64_t lin = (int64_t)(gen() % (uint64_t)total), rest = lin;
        std::vector<int64_t> idx(so.size());
        for (int i = (int)so.size() - 1; i >= 0; --i) { idx[i] = rest % so[i]; rest /= so[i]; }
        const double ref = ref_at(A, B, idx);
        worst = std::max(worst, std::fabs((double)C[lin] - ref) / std::max(std::fabs(ref), 1e-30));
    }
    std::string shape;
    for (int64_t x : so) shape += std::to_string(x) + " ";
    std::printf("%-16s -> [ %s] %.3f ms  check: max rel err %.3e -> %s\n", eq.c_str(), shape.c_str(), best * 1e3, worst, worst < 1e-4 ? "ok" : "FAILED");
    return worst < 1e-4;
}

// --flow: Einsum::execute exactly as cuTENSOR/einsum.cu:264-339 runs it — three tensor descriptors, plan preference, contraction
// descriptor and plan created, used for ONE cutensorContract and destroyed inside every call, plan cache at 1024 entries
// (:443-445), 1-GiB workspace constant (:380), no synchronisation between calls — next to the plan-once loop on the same
// buffers.  Prints one JSON line (bench.py reads it): per-call wall time of both loops and what cutensorCreatePlan costs on a
// cache hit and on a miss (cache mode NONE).
static int flow(int calls, unsigned cacheLines) {
    cutensorHandle_t handle;
    CT_OK(cutensorCreate(&handle));
    CT_OK(cutensorHandleResizePlanCache(handle, cacheLines));
    const int32_t mA[] = {'d', 'c', 'b', 'a'}, mB[] = {'e', 'b', 'c', 'd'}, mC[] = {'e', 'a'};   // reversed modes (einsum.cu:186-196)
    const int64_t eA[] = {64, 64, 64, 96}, eB[] = {96, 64, 64, 64}, eC[] = {96, 96};
    const uint64_t kWorksize = 1024ull * 1024ull * 8ull * 128ull;
    std::vector<float> A = sample::uniform(96 * 64 * 64 * 64, 11), B = sample::uniform(64 * 64 * 64 * 96, 12);
    DeviceBuffer<float> dA(A.size()), dB(B.size()), dC(96 * 96), dC2(96 * 96);
    DeviceBuffer<char> work(kWorksize);
    dA.upload(A); dB.upload(B);
    hipStream_t stream = nullptr;
    const float alpha = 1.f, beta = 0.f;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double planUs = 0.0;
    auto one_call = [&](float* out, cutensorCacheMode_t mode, bool timePlan) {
        cutensorTensorDescriptor_t descA, descB, descC;
        CT_OK(cutensorCreateTensorDescriptor(handle, &descA, 4, eA, nullptr, CUTENSOR_R_32F, 128));
        CT_OK(cutensorCreateTensorDescriptor(handle, &descC, 2, eC, nullptr, CUTENSOR_R_32F, 128));
        cutensorPlanPreference_t pref;
        CT_OK(cutensorCreatePlanPreference(handle, &pref, CUTENSOR_ALGO_DEFAULT, CUTENSOR_JIT_MODE_NONE));
        if (mode != CUTENSOR_CACHE_MODE_PEDANTIC)
            CT_OK(cutensorPlanPreferenceSetAttribute(handle, pref, CUTENSOR_PLAN_PREFERENCE_CACHE_MODE, &mode, sizeof(mode)));
        CT_OK(cutensorCreateTensorDescriptor(handle, &descB, 4, eB, nullptr, CUTENSOR_R_32F, 128));
        cutensorOperationDescriptor_t desc;
        CT_OK(cutensorCreateContraction(handle, &desc, descA, mA, CUTENSOR_OP_IDENTITY, descB, mB, CUTENSOR_OP_IDENTITY, descC, mC,
                                        CUTENSOR_OP_IDENTITY, descC, mC, CUTENSOR_COMPUTE_DESC_32F));
        cutensorPlan_t plan;
        const double t0 = timePlan ? now() : 0.0;
        CT_OK(cutensorCreatePlan(handle, &plan, desc, pref, kWorksize));
        if (timePlan) planUs += now() - t0;
        CT_OK(cutensorContract(handle, plan, &alpha, dA.p, dB.p, &beta, out, out, work.p, kWorksize, stream));
        CT_OK(cutensorDestroyPlan(plan));
        CT_OK(cutensorDestroyOperationDescriptor(desc));
        CT_OK(cutensorDestroyTensorDescriptor(descB));
        CT_OK(cutensorDestroyPlanPreference(pref));
        CT_OK(cutensorDestroyTensorDescriptor(descC));
        CT_OK(cutensorDestroyTensorDescriptor(descA));
    };
    // plan-once loop (python/einsum.h:277-442 split)
    cutensor_amd::Einsum<float, int64_t, 40> e("abcd,dcbe->ae", {96, 64, 64, 64}, {64, 64, 64, 96});
    if (!e.isInitialized() || !e.plan(handle, kWorksize)) { std::printf("plan-once setup failed\n"); return 1; }
    auto timed = [&](const std::function<void()>& body, int n) {   // wall clock around n asynchronous calls + one sync
        HIP_OK(hipDeviceSynchronize());
        const double t0 = now();
        for (int i = 0; i < n; ++i) body();
        const double issued = now();
        HIP_OK(hipDeviceSynchronize());
        const double t1 = now();
        return std::pair<double, double>((t1 - t0) / n, (issued - t0) / n);
    };
    for (int i = 0; i < 50; ++i) { one_call(dC.p, CUTENSOR_CACHE_MODE_PEDANTIC, false); e.execute(handle, dA.p, dB.p, dC2.p, work.p, stream); }
    double bestFlow = 1e100, bestFlowHost = 0, bestOnce = 1e100, bestOnceHost = 0;
    for (int rep = 0; rep < 5; ++rep) {
        auto f = timed([&] { one_call(dC.p, CUTENSOR_CACHE_MODE_PEDANTIC, false); }, calls);
        if (f.first < bestFlow) { bestFlow = f.first; bestFlowHost = f.second; }
        auto o = timed([&] { e.execute(handle, dA.p, dB.p, dC2.p, work.p, stream); }, calls);
        if (o.first < bestOnce) { bestOnce = o.first; bestOnceHost = o.second; }
    }
    // plan-creation cost alone: hit (memo) vs miss (cache bypassed), host clock around cutensorCreatePlan
    HIP_OK(hipDeviceSynchronize());
    planUs = 0.0;
    for (int i = 0; i < calls; ++i) one_call(dC.p, CUTENSOR_CACHE_MODE_PEDANTIC, true);
    const double hitUs = planUs / calls;
    HIP_OK(hipDeviceSynchronize());
    planUs = 0.0;
    for (int i = 0; i < calls; ++i) one_call(dC.p, CUTENSOR_CACHE_MODE_NONE, true);
    const double missUs = planUs / calls;
    HIP_OK(hipDeviceSynchronize());
    const std::vector<float> c1 = dC.download(), c2 = dC2.download();
    double worst = 0.0;
    for (size_t i = 0; i < c1.size(); ++i) worst = std::max(worst, std::fabs((double)c1[i] - (double)c2[i]) / std::max(std::fabs((double)c2[i]), 1e-30));
    const double flop = 2.0 * 96 * 96 * 64.0 * 64 * 64;
    std::printf("{\"what\": \"einsum.cu flow (descriptors + plan + contract + destroy per call, einsum.cu:264-339) vs plan once\", \"calls\": %d, "
                "\"plan_cache_entries\": %u, \"flow_us_per_call\": %.3f, \"flow_host_issue_us_per_call\": %.3f, \"plan_once_us_per_call\": %.3f, "
                "\"plan_once_host_issue_us_per_call\": %.3f, \"flow_gflops\": %.1f, \"plan_once_gflops\": %.1f, \"flow_over_plan_once\": %.4f, "
                "\"plan_create_us_hit\": %.3f, \"plan_create_us_miss\": %.3f, \"max_rel_diff_flow_vs_plan_once\": %.3e}\n",
                calls, cacheLines, bestFlow, bestFlowHost, bestOnce, bestOnceHost, flop / bestFlow * 1e-3, flop / bestOnce * 1e-3, bestFlow / bestOnce,
                hitUs, missUs, worst);
    CT_OK(cutensorDestroy(handle));
    return worst < 1e-6 ? 0 : 1;
}

int main(int argc, char** argv) {
    if (sample::arg_flag(argc, argv, "--flow"))
        return flow(sample::arg_int(argc, argv, "--calls", 2000), (unsigned)sample::arg_int(argc, argv, "--cache", 1024));
    cutensorHandle_t handle;
    CT_OK(cutensorCreate(&handle));
    CT_OK(cutensorHandleResizePlanCache(handle, 64));       // einsum.cu:445
    const std::vector<int64_t> sa{2, 4, 5}, sb{4, 8, 7};     // A[i,j,n] / A[n,i,j], B[j,m,k]
    auto A3 = [&](const std::vector<float>& A, int64_t x, int64_t y, int64_t z) { return (double)A[(x * 4 + y) * 5 + z]; };
    auto B3 = [&](const std::vector<float>& B, int64_t x, int64_t y, int64_t z) { return (double)B[(x * 8 + y) * 7 + z]; };
    bool ok = true;
    // 'ijn,jmk->inkm'
    ok &= run(handle, "ijn,jmk->inkm", sa, sb, [&](const std::vector<float>& A, const std::vector<float>& B, const std::vector<int64_t>& o) {
        double s = 0; for (int64_t j = 0; j < 4; ++j) s += A3(A, o[0], j, o[1]) * B3(B, j, o[3], o[2]); return s; }, 3);
    // implicit output: sorted modes that occur once -> 'ikmn'
    ok &= run(handle, "ijn,jmk", sa, sb, [&](const std::vector<float>& A, const std::vector<float>& B, const std::vector<int64_t>& o) {
        double s = 0; for (int64_t j = 0; j < 4; ++j) s += A3(A, o[0], j, o[3]) * B3(B, j, o[2], o[1]); return s; }, 3);
    // unary: implicit 'nij' -> 'ijn' (a permutation through cutensorReduce), explicit permutation, reduction over n
    ok &= run(handle, "nij", sa, {}, [&](const std::vector<float>& A, const std::vector<float>&, const std::vector<int64_t>& o) { return A3(A, o[2], o[0], o[1]); }, 3);
    ok &= run(handle, "nij->ijn", sa, {}, [&](const std::vector<float>& A, const std::vector<float>&, const std::vector<int64_t>& o) { return A3(A, o[2], o[0], o[1]); }, 3);
    ok &= run(handle, "nij->ji", sa, {}, [&](const std::vector<float>& A, const std::vector<float>&, const std::vector<int64_t>& o) {
        double s = 0; for (int64_t n = 0; n < 2; ++n) s += A3(A, n, o[1], o[0]); return s; }, 3);
    // unsupported inputs stay uninitialised (einsum.cu:76-79, :118-127)
    if (cutensor_amd::Einsum<float, int64_t, 40>("ab...,bc->ac", {2, 3, 4}, {3, 4}).isInitialized()) { std::printf("'...' must not be supported\n"); ok = false; }
    // headline: 'abcd,dcbe->ae', a = e = 96, b = c = d = 64
    const std::vector<int64_t> ha{96, 64, 64, 64}, hb{64, 64, 64, 96};
    ok &= run(handle, "abcd,dcbe->ae", ha, hb, [&](const std::vector<float>& A, const std::vector<float>& B, const std::vector<int64_t>& o) {
        double s = 0;
        for (int64_t b = 0; b < 64; ++b) for (int64_t c = 0; c < 64; ++c) for (int64_t d = 0; d < 64; ++d)
            s += (double)A[((o[0] * 64 + b) * 64 + c) * 64 + d] * (double)B[((d * 64 + c) * 64 + b) * 96 + o[1]];
        return s; }, 20);
    CT_OK(cutensorDestroy(handle));
    std::printf("einsum: %s\n", ok ? "all checks ok" : "FAILED");
    return ok ? 0 : 1;
}
