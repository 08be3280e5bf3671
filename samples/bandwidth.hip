// bandwidth.hip — native HIP driver of the two bandwidth-bound samples:
//   reduction   C_{m,v} = alpha * sum_{h,k} A_{m,h,k,v} + beta * C_{m,v}     cuTENSOR/reduction.cu:49, extents :56-59 (196, 256, 64, 64),
//               alpha 1.1, beta 0 (:45-46), OP_ADD (:134), workspace estimate -> plan -> cutensorReduce (:150-222), GB/s (:229-231)
//   permutation C_{c,w,h,n} = alpha * A_{w,h,c,n}                              cuTENSOR/elementwise_permute.cu:51, extents :66-69 (32, 128, 128, 128),
//               alpha 1.0 (:44), cutensorCreatePermutation -> plan with limit 0 -> cutensorPermute (:142-200), GB/s = 2 |C| (:208)
// each checked against host arithmetic (sums in fp64; the permutation must be bit-exact).
#include <cmath>

#include "common.hpp"

int main() {
    using namespace sample;
    cutensorHandle_t handle;
    CT_OK(cutensorCreate(&handle));
    bool ok = true;
    {   // ---- reduction -------------------------------------------------------------------------------------------
        const std::vector<int32_t> modeA{'m', 'h', 'k', 'v'}, modeC{'m', 'v'};
        const std::vector<int64_t> eA{196, 256, 64, 64}, eC{196, 64};
        std::vector<float> A = uniform(product(eA), 21), C = uniform(product(eC), 22);
        DeviceBuffer<float> dA(A.size()), dC(C.size());
        dA.upload(A); dC.upload(C);
        cutensorTensorDescriptor_t descA, descC;
        CT_OK(cutensorCreateTensorDescriptor(handle, &descA, 4, eA.data(), nullptr, CUTENSOR_R_32F, 256));
        CT_OK(cutensorCreateTensorDescriptor(handle, &descC, 2, eC.data(), nullptr, CUTENSOR_R_32F, 256));
        cutensorOperationDescriptor_t desc;
        CT_OK(cutensorCreateReduction(handle, &desc, descA, modeA.data(), CUTENSOR_OP_IDENTITY, descC, modeC.data(), CUTENSOR_OP_IDENTITY,
                                      descC, modeC.data(), CUTENSOR_OP_ADD, CUTENSOR_COMPUTE_DESC_32F));
        cutensorPlanPreference_t pref;
        CT_OK(cutensorCreatePlanPreference(handle, &pref, CUTENSOR_ALGO_DEFAULT, CUTENSOR_JIT_MODE_NONE));
        uint64_t estimate = 0, required = 0;
        CT_OK(cutensorEstimateWorkspaceSize(handle, desc, pref, CUTENSOR_WORKSPACE_DEFAULT, &estimate));
        cutensorPlan_t plan;
        CT_OK(cutensorCreatePlan(handle, &plan, desc, pref, estimate));
        CT_OK(cutensorPlanGetAttribute(handle, plan, CUTENSOR_PLAN_REQUIRED_WORKSPACE, &required, sizeof(required)));
        DeviceBuffer<char> work(required);
        const float alpha = 1.1f, beta = 0.f;
        double best = 1e100;
        for (int i = 0; i < 3; ++i) {
            GpuTimer t;
            t.start();
            CT_OK(cutensorReduce(handle, plan, &alpha, dA.p, &beta, dC.p, dC.p, work.p, required, nullptr));
            best = std::min(best, t.seconds());
        }
        const std::vector<float> D = dC.download();
        double worst = 0.0;
        for (int64_t v = 0; v < 64; v += 7)
            for (int64_t m = 0; m < 196; m += 5) {
                double s = 0.0;
                for (int64_t k = 0; k < 64; ++k)
                    for (int64_t h = 0; h < 256; ++h) s += (double)A[m + 196 * (h + 256 * (k + 64 * v))];
                worst = std::max(worst, std::fabs((double)D[m + 196 * v] - 1.1 * s) / (1.1 * s));
            }
        std::printf("reduction: %.2f GB/s (%.3f ms)  check: max rel err %.3e -> %s\n", 4.0 * (A.size() + C.size()) / best / 1e9, best * 1e3, worst,
                    worst < 1e-5 ? "ok" : "FAILED");
        ok &= worst < 1e-5;
        CT_OK(cutensorDestroyPlan(plan)); CT_OK(cutensorDestroyPlanPreference(pref)); CT_OK(cutensorDestroyOperationDescriptor(desc));
        CT_OK(cutensorDestroyTensorDescriptor(descA)); CT_OK(cutensorDestroyTensorDescriptor(descC));
    }
    {   // ---- permutation -----------------------------------------------------------------------------------------
        const std::vector<int32_t> modeC{'c', 'w', 'h', 'n'}, modeA{'w', 'h', 'c', 'n'};
        const int64_t W = 32, H = 128, Cc = 128, N = 128;
        const std::vector<int64_t> eA{W, H, Cc, N}, eC{Cc, W, H, N};
        std::vector<float> A = uniform(product(eA), 23);
        DeviceBuffer<float> dA(A.size()), dC(A.size());
        dA.upload(A);
        cutensorTensorDescriptor_t descA, descC;
        CT_OK(cutensorCreateTensorDescriptor(handle, &descA, 4, eA.data(), nullptr, CUTENSOR_R_32F, 128));
        CT_OK(cutensorCreateTensorDescriptor(handle, &descC, 4, eC.data(), nullptr, CUTENSOR_R_32F, 128));
        cutensorOperationDescriptor_t desc;
        CT_OK(cutensorCreatePermutation(handle, &desc, descA, modeA.data(), CUTENSOR_OP_IDENTITY, descC, modeC.data(), CUTENSOR_COMPUTE_DESC_32F));
        cutensorPlanPreference_t pref;
        CT_OK(cutensorCreatePlanPreference(handle, &pref, CUTENSOR_ALGO_DEFAULT, CUTENSOR_JIT_MODE_NONE));
        cutensorPlan_t plan;
        CT_OK(cutensorCreatePlan(handle, &plan, desc, pref, 0));
        const float alpha = 1.0f;
        double best = 1e100;
        for (int i = 0; i < 3; ++i) {
            GpuTimer t;
            t.start();
            CT_OK(cutensorPermute(handle, plan, &alpha, dA.p, dC.p, nullptr));
            best = std::min(best, t.seconds());
        }
        const std::vector<float> D = dC.download();
        int64_t bad = 0;
        for (int64_t n = 0; n < N; ++n)
            for (int64_t h = 0; h < H; ++h)
                for (int64_t w = 0; w < W; ++w)
                    for (int64_t c = 0; c < Cc; ++c)
                        bad += D[c + Cc * (w + W * (h + H * n))] != A[w + W * (h + H * (c + Cc * n))];
        std::printf("permutation: %.2f GB/s (%.3f ms)  check: %lld mismatches of %zu -> %s\n", 2.0 * 4.0 * A.size() / best / 1e9, best * 1e3,
                    (long long)bad, A.size(), bad == 0 ? "ok" : "FAILED");
        ok &= bad == 0;
        CT_OK(cutensorDestroyPlan(plan)); CT_OK(cutensorDestroyPlanPreference(pref)); CT_OK(cutensorDestroyOperationDescriptor(desc));
        CT_OK(cutensorDestroyTensorDescriptor(descA)); CT_OK(cutensorDestroyTensorDescriptor(descC));
    }
    CT_OK(cutensorDestroy(handle));
    return ok ? 0 : 1;
}
