// Host-side cost of the einsum.cu flow (cuTENSOR/einsum.cu:264-329): three tensor descriptors + contraction descriptor +
// plan preference + plan, created and destroyed per call, for 'abcd,dcbe->ae' (96/64/64/64/96 fp32).  No GPU needed:
// nothing is launched.  Usage: plan_cost [cacheEntries] [iterations]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cutensor.h>

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const unsigned cache = argc > 1 ? (unsigned)std::atoi(argv[1]) : 1024;
    const int iters = argc > 2 ? std::atoi(argv[2]) : 20000;
    cutensorHandle_t h;
    if (cutensorCreate(&h) != CUTENSOR_STATUS_SUCCESS) return 1;
    cutensorHandleResizePlanCache(h, cache);
    const int32_t mA[] = {'d', 'c', 'b', 'a'}, mB[] = {'e', 'b', 'c', 'd'}, mC[] = {'e', 'a'};
    const int64_t eA[] = {64, 64, 64, 96}, eB[] = {96, 64, 64, 64}, eC[] = {96, 96};
    std::vector<double> t(iters), tplan(iters);
    for (int it = 0; it < iters; ++it) {
        const double t0 = now_us();
        cutensorTensorDescriptor_t dA, dB, dC;
        cutensorCreateTensorDescriptor(h, &dA, 4, eA, nullptr, CUTENSOR_R_32F, 128);
        cutensorCreateTensorDescriptor(h, &dC, 2, eC, nullptr, CUTENSOR_R_32F, 128);
        cutensorPlanPreference_t pref;
        cutensorCreatePlanPreference(h, &pref, CUTENSOR_ALGO_DEFAULT, CUTENSOR_JIT_MODE_NONE);
        cutensorCreateTensorDescriptor(h, &dB, 4, eB, nullptr, CUTENSOR_R_32F, 128);
        cutensorOperationDescriptor_t op;
        cutensorCreateContraction(h, &op, dA, mA, CUTENSOR_OP_IDENTITY, dB, mB, CUTENSOR_OP_IDENTITY, dC, mC, CUTENSOR_OP_IDENTITY, dC, mC,
                                  CUTENSOR_COMPUTE_DESC_32F);
        const double t1 = now_us();
        cutensorPlan_t plan;
        if (cutensorCreatePlan(h, &plan, op, pref, 1ull << 30) != CUTENSOR_STATUS_SUCCESS) return 2;
        const double t2 = now_us();
        cutensorDestroyPlan(plan);
        cutensorDestroyOperationDescriptor(op);
        cutensorDestroyTensorDescriptor(dB);
        cutensorDestroyPlanPreference(pref);
        cutensorDestroyTensorDescriptor(dC);
        cutensorDestroyTensorDescriptor(dA);
        t[it] = now_us() - t0;
        tplan[it] = t2 - t1;
    }
    std::sort(t.begin() + 1, t.end());
    std::sort(tplan.begin() + 1, tplan.end());
    std::printf("{\"plan_cache\": %u, \"iters\": %d, \"first_call_us\": %.2f, \"flow_median_us\": %.2f, \"flow_p10_us\": %.2f, \"create_plan_median_us\": %.2f}\n",
                cache, iters, t[0], t[1 + (iters - 1) / 2], t[1 + (iters - 1) / 10], tplan[1 + (iters - 1) / 2]);
    cutensorDestroy(h);
    return 0;
}
