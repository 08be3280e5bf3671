// einsum.hip — native HIP driver of the einsum helper: the demo calls of cuTENSOR/einsum.cu:441-455 (shapes {2,4,5},{4,8,7},
// equations :447-451) and the headline equation 'abcd,dcbe->ae' (BASELINE configs[1]) through cutensor_amd::Einsum<>
// (csrc/einsum/einsum.hpp, the counterpart of Einsum<> in einsum.cu:57-391), each checked against fp64 host loops.
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

int main() {
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
