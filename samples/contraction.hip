// contraction.hip — native HIP driver of the canonical contraction (the call sequence of cuTENSOR/contraction.cu:122-270):
//   C_{m,u,n,v} = alpha * A_{m,h,k,n} B_{u,k,v,h} + beta * C_{m,u,n,v}     (:43-59; alpha 1.1, beta 0: :184-185)
// descriptors (:131-137) -> contraction (:162-168) -> scalar type query (:176-180) -> plan preference (:194-198) ->
// workspace estimate (:207-211) -> plan (:218-222) -> required workspace (:231-239) -> cutensorContract, min of 3 (:252-270),
// GFLOP/s and GB/s by the sample's formulas (:61, :274-277).  Extras the reference sample does not have: --check N compares N
// sampled outputs with fp64 host dot products; --shrink runs the 12/12/12/8/8/8 extents and checks every element.
#include <cmath>
#include <unordered_map>

#include "common.hpp"

int main(int argc, char** argv) {
    using namespace sample;
    const bool shrink = arg_flag(argc, argv, "--shrink");
    const int nCheck = arg_int(argc, argv, "--check", 4096);
    std::vector<int32_t> modeC{'m', 'u', 'n', 'v'}, modeA{'m', 'h', 'k', 'n'}, modeB{'u', 'k', 'v', 'h'};
    std::unordered_map<int32_t, int64_t> extent;
    extent['m'] = shrink ? 12 : 96; extent['n'] = shrink ? 12 : 96; extent['u'] = shrink ? 12 : 96;
    extent['v'] = shrink ? 8 : 64;  extent['h'] = shrink ? 8 : 64;  extent['k'] = shrink ? 8 : 64;
    auto ext_of = [&](const std::vector<int32_t>& m) { std::vector<int64_t> e; for (int32_t x : m) e.push_back(extent[x]); return e; };
    const std::vector<int64_t> eA = ext_of(modeA), eB = ext_of(modeB), eC = ext_of(modeC);
    double gflops = 2.0;
    for (auto& kv : extent) gflops *= (double)kv.second;
    gflops /= 1e9;

    std::vector<float> A = uniform(product(eA), 1234), B = uniform(product(eB), 1235), C = uniform(product(eC), 1236);
    DeviceBuffer<float> dA(A.size()), dB(B.size()), dC(C.size());
    dA.upload(A); dB.upload(B); dC.upload(C);

    cutensorHandle_t handle;
    CT_OK(cutensorCreate(&handle));
    const uint32_t kAlignment = 128;
    cutensorTensorDescriptor_t descA, descB, descC;
    CT_OK(cutensorCreateTensorDescriptor(handle, &descA, (uint32_t)eA.size(), eA.data(), nullptr, CUTENSOR_R_32F, kAlignment));
    CT_OK(cutensorCreateTensorDescriptor(handle, &descB, (uint32_t)eB.size(), eB.data(), nullptr, CUTENSOR_R_32F, kAlignment));
    CT_OK(cutensorCreateTensorDescriptor(handle, &descC, (uint32_t)eC.size(), eC.data(), nullptr, CUTENSOR_R_32F, kAlignment));
    cutensorOperationDescriptor_t desc;
    CT_OK(cutensorCreateContraction(handle, &desc, descA, modeA.data(), CUTENSOR_OP_IDENTITY, descB, modeB.data(), CUTENSOR_OP_IDENTITY,
                                    descC, modeC.data(), CUTENSOR_OP_IDENTITY, descC, modeC.data(), CUTENSOR_COMPUTE_DESC_32F));
    cutensorDataType_t scalarType;
    CT_OK(cutensorOperationDescriptorGetAttribute(handle, desc, CUTENSOR_OPERATION_DESCRIPTOR_SCALAR_TYPE, &scalarType, sizeof(scalarType)));
    if (scalarType != CUTENSOR_R_32F) { std::printf("unexpected scalar type\n"); return 1; }
    const float alpha = 1.1f, beta = 0.f;
    cutensorPlanPreference_t pref;
    CT_OK(cutensorCreatePlanPreference(handle, &pref, CUTENSOR_ALGO_DEFAULT, CUTENSOR_JIT_MODE_NONE));
    uint64_t estimate = 0;
    CT_OK(cutensorEstimateWorkspaceSize(handle, desc, pref, CUTENSOR_WORKSPACE_DEFAULT, &estimate));
    cutensorPlan_t plan;
    CT_OK(cutensorCreatePlan(handle, &plan, desc, pref, estimate));
    uint64_t required = 0;
    CT_OK(cutensorPlanGetAttribute(handle, plan, CUTENSOR_PLAN_REQUIRED_WORKSPACE, &required, sizeof(required)));
    if (required > estimate) { std::printf("required workspace exceeds the estimate\n"); return 1; }
    DeviceBuffer<char> work(required);

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    double best = 1e100;
    for (int i = 0; i < 3; ++i) {
        dC.upload(C);
        HIP_OK(hipDeviceSynchronize());
        GpuTimer t(stream);
        t.start();
        CT_OK(cutensorContract(handle, plan, &alpha, dA.p, dB.p, &beta, dC.p, dC.p, work.p, required, stream));
        best = std::min(best, t.seconds());
    }
    const double bytes = 4.0 * (A.size() + B.size() + C.size());
    std::printf("contraction: %.2f GFLOPs/s %.2f GB/s (%.3f ms)\n", gflops / best, bytes / best / 1e9, best * 1e3);

    // ---- value check: D[m,u,n,v] = 1.1 * sum_{h,k} A[m,h,k,n] B[u,k,v,h] -------------------------------------------
    const std::vector<float> D = dC.download();
    const std::vector<int64_t> sA = packed_strides(eA), sB = packed_strides(eB), sC = packed_strides(eC);
    std::mt19937 gen(7);
    const int64_t total = product(eC);
    const int64_t samples = shrink ? total : std::min<int64_t>(nCheck, total);
    double worst = 0.0;
    for (int64_t sIdx = 0; sIdx < samples; ++sIdx) {
        const int64_t lin = shrink ? sIdx : (int64_t)(gen() % (uint64_t)total);
        int64_t rest = lin, m, u, n, v;
        m = rest % extent['m']; rest /= extent['m'];
        u = rest % extent['u']; rest /= extent['u'];
        n = rest % extent['n']; rest /= extent['n'];
        v = rest;
        double acc = 0.0;
        for (int64_t h = 0; h < extent['h']; ++h)
            for (int64_t k = 0; k < extent['k']; ++k)
                acc += (double)A[m * sA[0] + h * sA[1] + k * sA[2] + n * sA[3]] * (double)B[u * sB[0] + k * sB[1] + v * sB[2] + h * sB[3]];
        const double ref = 1.1 * acc;
        worst = std::max(worst, std::fabs((double)D[m * sC[0] + u * sC[1] + n * sC[2] + v * sC[3]] - ref) / std::fabs(ref));
    }
    std::printf("check: %lld outputs, max rel err %.3e -> %s\n", (long long)samples, worst, worst < 1e-4 ? "ok" : "FAILED");

    CT_OK(cutensorDestroyPlan(plan));
    CT_OK(cutensorDestroyPlanPreference(pref));
    CT_OK(cutensorDestroyOperationDescriptor(desc));
    CT_OK(cutensorDestroyTensorDescriptor(descA));
    CT_OK(cutensorDestroyTensorDescriptor(descB));
    CT_OK(cutensorDestroyTensorDescriptor(descC));
    CT_OK(cutensorDestroy(handle));
    HIP_OK(hipStreamDestroy(stream));
    return worst < 1e-4 ? 0 : 1;
}
