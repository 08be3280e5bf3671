// einsum.hpp — host-side einsum front end above the cuTENSOR C ABI.
//
// Mirrors the interface of the reference's helper `Einsum<ComputeType, IntType, kMaxNumModes_>`
// (cuTENSOR/einsum.cu:57-391; library-grade variant cuTENSOR/python/einsum.h:73-470) so that code
// written against it ports by changing one include: same constructor arguments, isInitialized(),
// getOutputShape(), getWorksize(), execute().  Behaviour that callers rely on:
//   * "..." is not supported, size/rank mismatches and > kMaxNumModes_ modes leave the object
//     uninitialised (einsum.cu:71-79, :118-127) and execute() returns false;
//   * without "->" the output is the sorted list of modes that occur in exactly one operand (:164-179);
//   * framework tensors are row-major, cuTENSOR's are column-major: every mode list and extent
//     list is reversed before descriptors are built (:186-196);
//   * binary equations go to cutensorContract, unary ones to cutensorReduce with OP_ADD — which also
//     serves pure permutations (:340-373, demo lines :449-450);
//   * alpha = 1, beta = 0 (:331-332).
// Unlike einsum.cu (which rebuilds descriptors and plan in every execute(), :264-329) the plan is
// built once, on first use, and reused — the split cuTENSOR/python/einsum.h makes (plan() :277-399,
// execute() :411-442).
#pragma once
#include <algorithm>
#include <array>
#include <complex>
#include <cstdint>
#include <string>
#include <vector>

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <cutensor.h>

namespace cutensor_amd {

template <typename T> struct EinsumTypeTraits;
template <> struct EinsumTypeTraits<double> {
    static cutensorDataType_t dataType() { return CUTENSOR_R_64F; }
    static cutensorComputeDescriptor_t computeDesc() { return CUTENSOR_COMPUTE_DESC_64F; }
    typedef double ScalarType;
};
template <> struct EinsumTypeTraits<float> {
    static cutensorDataType_t dataType() { return CUTENSOR_R_32F; }
    static cutensorComputeDescriptor_t computeDesc() { return CUTENSOR_COMPUTE_DESC_32F; }
    typedef float ScalarType;
};
template <> struct EinsumTypeTraits<__half> {
    static cutensorDataType_t dataType() { return CUTENSOR_R_16F; }
    static cutensorComputeDescriptor_t computeDesc() { return CUTENSOR_COMPUTE_DESC_16F; }
    typedef float ScalarType;
};
template <> struct EinsumTypeTraits<__hip_bfloat16> {
    static cutensorDataType_t dataType() { return CUTENSOR_R_16BF; }
    static cutensorComputeDescriptor_t computeDesc() { return CUTENSOR_COMPUTE_DESC_16BF; }
    typedef float ScalarType;
};

// complex data (python/cutensor/torch/einsum_test.py:56-68: complex64 / complex128 cases): contractions only
template <> struct EinsumTypeTraits<std::complex<float>> {
    static cutensorDataType_t dataType() { return CUTENSOR_C_32F; }
    static cutensorComputeDescriptor_t computeDesc() { return CUTENSOR_COMPUTE_DESC_32F; }
    typedef std::complex<float> ScalarType;
};
template <> struct EinsumTypeTraits<std::complex<double>> {
    static cutensorDataType_t dataType() { return CUTENSOR_C_64F; }
    static cutensorComputeDescriptor_t computeDesc() { return CUTENSOR_COMPUTE_DESC_64F; }
    typedef std::complex<double> ScalarType;
};

template <typename ComputeType, typename IntType, int kMaxNumModes_>
class Einsum {
public:
    // conjugate an operand inside the contraction (python/einsum.h: the conjA / conjB arguments the PyTorch binding
    // uses for complex gradients, torch/einsum.py:50-61); call before plan() / execute()
    void setConjugate(bool conjA, bool conjB) { conjA_ = conjA; conjB_ = conjB; }

    Einsum(const std::string& equation, const std::vector<IntType>& A_shape,
           const std::vector<IntType>& B_shape = std::vector<IntType>()) {
        if (equation.find("...") != std::string::npos) return;   // broadcasting: not supported
        // strip blanks, then cut at ',' and "->"
        std::string eq;
        for (char c : equation)
            if (c != ' ') eq.push_back(c);
        const size_t arrow = eq.find("->");
        const size_t comma = eq.find(',');
        const bool implicit = (arrow == std::string::npos);
        const size_t lhsEnd = implicit ? eq.size() : arrow;
        std::string sA, sB, sC;
        if (comma != std::string::npos && comma < lhsEnd) {
            hasB_ = true;   // two operands even when the second is a scalar ("i,->i": zero modes, e.g. a gradient of "i,i->")
            sA = eq.substr(0, comma);
            sB = eq.substr(comma + 1, lhsEnd - comma - 1);
        } else {
            sA = eq.substr(0, lhsEnd);
        }
        if (!implicit) sC = eq.substr(arrow + 2);
        if (sA.size() != A_shape.size() || sB.size() != B_shape.size()) return;
        if (sA.size() > (size_t)kMaxNumModes_ || sB.size() > (size_t)kMaxNumModes_) return;
        if (implicit) {
            for (char m : sA)
                if (sB.find(m) == std::string::npos) sC.push_back(m);
            for (char m : sB)
                if (sA.find(m) == std::string::npos) sC.push_back(m);
            std::sort(sC.begin(), sC.end());
        }
        if (sC.size() > (size_t)kMaxNumModes_) return;
        // row-major (framework) -> column-major (cuTENSOR): reverse everything
        for (size_t i = 0; i < sA.size(); ++i) {
            modesA_.push_back(sA[sA.size() - 1 - i]);
            extentA_.push_back((int64_t)A_shape[sA.size() - 1 - i]);
        }
        for (size_t i = 0; i < sB.size(); ++i) {
            modesB_.push_back(sB[sB.size() - 1 - i]);
            extentB_.push_back((int64_t)B_shape[sB.size() - 1 - i]);
        }
        for (size_t i = 0; i < sC.size(); ++i) {
            const int32_t m = sC[sC.size() - 1 - i];
            modesC_.push_back(m);
            int64_t e = 0;
            auto ia = std::find(modesA_.begin(), modesA_.end(), m);
            if (ia != modesA_.end()) {
                e = extentA_[ia - modesA_.begin()];
            } else {
                auto ib = std::find(modesB_.begin(), modesB_.end(), m);
                if (ib != modesB_.end()) e = extentB_[ib - modesB_.begin()];
            }
            extentC_.push_back(e);
        }
        isInitialized_ = true;
    }

    ~Einsum() {
        if (plan_) cutensorDestroyPlan(plan_);
    }
    Einsum(const Einsum&) = delete;
    Einsum& operator=(const Einsum&) = delete;

    bool isInitialized() const { return isInitialized_; }
    size_t getWorksize() const { return kWorksize_; }

    std::vector<IntType> getOutputShape() const {
        if (!isInitialized_) return {};
        std::vector<IntType> shape(extentC_.size());
        for (size_t i = 0; i < extentC_.size(); ++i) shape[i] = (IntType)extentC_[extentC_.size() - 1 - i];
        return shape;
    }

    // Builds descriptors and the plan (once).  Returns false on any non-success status.
    bool plan(const cutensorHandle_t handle, uint64_t workspaceLimit) {
        if (!isInitialized_) return false;
        if (plan_) return true;
        const cutensorDataType_t type = EinsumTypeTraits<ComputeType>::dataType();
        const cutensorComputeDescriptor_t compute = EinsumTypeTraits<ComputeType>::computeDesc();
        const uint32_t kAlignment = 128;
        cutensorTensorDescriptor_t dA = nullptr, dB = nullptr, dC = nullptr;
        cutensorOperationDescriptor_t op = nullptr;
        cutensorPlanPreference_t pref = nullptr;
        bool ok = cutensorCreateTensorDescriptor(handle, &dA, (uint32_t)modesA_.size(), extentA_.data(), nullptr, type, kAlignment) == CUTENSOR_STATUS_SUCCESS &&
                  cutensorCreateTensorDescriptor(handle, &dC, (uint32_t)modesC_.size(), extentC_.data(), nullptr, type, kAlignment) == CUTENSOR_STATUS_SUCCESS &&
                  cutensorCreatePlanPreference(handle, &pref, CUTENSOR_ALGO_DEFAULT, CUTENSOR_JIT_MODE_NONE) == CUTENSOR_STATUS_SUCCESS;
        if (ok && hasB_) {
            ok = cutensorCreateTensorDescriptor(handle, &dB, (uint32_t)modesB_.size(), extentB_.data(), nullptr, type, kAlignment) == CUTENSOR_STATUS_SUCCESS &&
                 cutensorCreateContraction(handle, &op, dA, modesA_.data(), conjA_ ? CUTENSOR_OP_CONJ : CUTENSOR_OP_IDENTITY, dB, modesB_.data(),
                                           conjB_ ? CUTENSOR_OP_CONJ : CUTENSOR_OP_IDENTITY,
                                           dC, modesC_.data(), CUTENSOR_OP_IDENTITY, dC, modesC_.data(), compute) == CUTENSOR_STATUS_SUCCESS;
        } else if (ok) {
            ok = cutensorCreateReduction(handle, &op, dA, modesA_.data(), CUTENSOR_OP_IDENTITY, dC, modesC_.data(), CUTENSOR_OP_IDENTITY,
                                         dC, modesC_.data(), CUTENSOR_OP_ADD, compute) == CUTENSOR_STATUS_SUCCESS;
        }
        if (ok) ok = cutensorCreatePlan(handle, &plan_, op, pref, workspaceLimit) == CUTENSOR_STATUS_SUCCESS;
        if (ok) ok = cutensorPlanGetAttribute(handle, plan_, CUTENSOR_PLAN_REQUIRED_WORKSPACE, &requiredWorkspace_, sizeof(requiredWorkspace_)) == CUTENSOR_STATUS_SUCCESS;
        cutensorDestroyOperationDescriptor(op);
        cutensorDestroyPlanPreference(pref);
        cutensorDestroyTensorDescriptor(dA);
        cutensorDestroyTensorDescriptor(dB);
        cutensorDestroyTensorDescriptor(dC);
        if (!ok && plan_) { cutensorDestroyPlan(plan_); plan_ = nullptr; }
        return ok;
    }

    uint64_t requiredWorkspace() const { return requiredWorkspace_; }

    // Drops the plan and builds it again under a different workspace limit — the CUTENSOR_WORKSPACE_MIN retry of
    // python/cutensor/torch/einsum.cc:104-123 when the workspace of the first plan cannot be allocated.
    bool replan(const cutensorHandle_t handle, uint64_t workspaceLimit) {
        if (plan_) { cutensorDestroyPlan(plan_); plan_ = nullptr; }
        requiredWorkspace_ = 0;
        return plan(handle, workspaceLimit);
    }

    // C = einsum(A, B) on `stream`; work_raw must hold getWorksize() bytes (or requiredWorkspace()
    // after an explicit plan()).
    bool execute(const cutensorHandle_t handle, const void* A_raw, const void* B_raw, void* C_raw,
                 void* work_raw, cudaStream_t stream) {
        const bool plannedByCaller = plan_ != nullptr;    // plan()/replan() called explicitly: the buffer holds requiredWorkspace()
        if (!plan(handle, kWorksize_)) return false;
        const uint64_t workSize = plannedByCaller ? requiredWorkspace_ : (uint64_t)kWorksize_;
        typename EinsumTypeTraits<ComputeType>::ScalarType alpha = 1, beta = 0;
        cutensorStatus_t st;
        if (hasB_)
            st = cutensorContract(handle, plan_, &alpha, A_raw, B_raw, &beta, C_raw, C_raw, work_raw, workSize, stream);
        else
            st = cutensorReduce(handle, plan_, &alpha, A_raw, &beta, C_raw, C_raw, work_raw, workSize, stream);
        return st == CUTENSOR_STATUS_SUCCESS;
    }

    // introspection used by the tests (cuTENSOR-order, i.e. reversed, lists)
    const std::vector<int32_t>& modesA() const { return modesA_; }
    const std::vector<int32_t>& modesB() const { return modesB_; }
    const std::vector<int32_t>& modesC() const { return modesC_; }
    const std::vector<int64_t>& extentC() const { return extentC_; }
    cutensorPlan_t rawPlan() const { return plan_; }

private:
    static const size_t kWorksize_ = 1024ULL * 1024ULL * 1024ULL;   // 1 GiB, as einsum.cu:380
    bool isInitialized_ = false;
    bool hasB_ = false;
    bool conjA_ = false, conjB_ = false;
    std::vector<int32_t> modesA_, modesB_, modesC_;
    std::vector<int64_t> extentA_, extentB_, extentC_;
    cutensorPlan_t plan_ = nullptr;
    uint64_t requiredWorkspace_ = 0;
};

}  // namespace cutensor_amd
