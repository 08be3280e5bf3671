// multi_gpu.hip — native HIP driver of the two cuTENSORMg programs, with the value check the reference samples lack:
//   default        cuTENSORMg/contraction_multi_gpu.cu:126-383: C_{i,j} = A_{i,k} B_{k,j}, extents 4096, block 2048, 2 x 2
//                  block-cyclic grids per tensor, cells owned by the listed devices cyclically (fillUp(), :178-187);
//                  usage: multi_gpu [extent [block]] [--devices d0,d1,...]
//   --blog n s     cuTENSORMg/blog_post.cu:131-302: <numDevices> <scaling>, the multi-mode contraction
//                  C_{M0,N0,M1,N1,M2,N2} = A_{K0,M0,M1,K1,M2,K2} B_{K0,N0,K1,N1,K2,N2} (:177-179) with the sample's
//                  block sizes (:168-175) and its "as evenly as possible" device grids (:78-101)
// One process, one host thread, one stream per device; wall clock + per-device sync, min of 3 (:323-345 / :288-302).
#include <chrono>
#include <cmath>
#include <map>
#include <numeric>
#include <sstream>

#include <cutensorMg.h>

#include "common.hpp"

namespace {

struct Dist {                       // a block-cyclic tensor: host mirror + device cells
    std::vector<int32_t> modes;
    std::vector<int64_t> extent, block;
    std::vector<int32_t> dcount, devices;
    std::vector<int64_t> nlb, elemStride, blockStride, cellStride;
    int64_t cellElems = 1, cells = 1;
    std::vector<std::vector<float>> host;
    std::vector<void*> dev;
    cutensorMgTensorDescriptor_t desc = nullptr;

    void layout() {
        const size_t n = modes.size();
        nlb.resize(n); elemStride.resize(n); blockStride.resize(n); cellStride.resize(n);
        int64_t run = 1;
        cells = 1;
        for (size_t i = 0; i < n; ++i) {
            nlb[i] = ((extent[i] + block[i] - 1) / block[i] + dcount[i] - 1) / dcount[i];   // whole blocks per cell (blog_post.cu:107-113)
            cellStride[i] = cells; cells *= dcount[i];
            elemStride[i] = run; run *= block[i];
        }
        int64_t brun = run;
        for (size_t i = 0; i < n; ++i) { blockStride[i] = brun; brun *= nlb[i]; }
        cellElems = brun;
    }
    void locate(const std::vector<int64_t>& idx, int64_t& cell, int64_t& off) const {
        cell = 0; off = 0;
        for (size_t i = 0; i < modes.size(); ++i) {
            const int64_t b = idx[i] / block[i], w = idx[i] % block[i];
            cell += (b % dcount[i]) * cellStride[i];
            off += w * elemStride[i] + (b / dcount[i]) * blockStride[i];
        }
    }
    float at(const std::vector<int64_t>& idx) const { int64_t c, o; locate(idx, c, o); return host[(size_t)c][(size_t)o]; }
    void create(cutensorMgHandle_t h, uint32_t seed, bool fill) {
        layout();
        CT_OK(cutensorMgCreateTensorDescriptor(h, &desc, (uint32_t)modes.size(), extent.data(), nullptr, block.data(), nullptr, dcount.data(),
                                               (uint32_t)devices.size(), devices.data(), HIP_R_32F));
        for (int64_t c = 0; c < cells; ++c) {
            host.push_back(fill ? sample::uniform((size_t)cellElems, seed + (uint32_t)c) : std::vector<float>((size_t)cellElems, 0.f));
            void* p = nullptr;
            HIP_OK(hipSetDevice(devices[(size_t)c]));
            HIP_OK(hipMalloc(&p, (size_t)cellElems * 4));
            HIP_OK(hipMemcpy(p, host.back().data(), (size_t)cellElems * 4, hipMemcpyHostToDevice));
            dev.push_back(p);
        }
    }
    void download() {
        for (int64_t c = 0; c < cells; ++c) {
            HIP_OK(hipSetDevice(devices[(size_t)c]));
            HIP_OK(hipMemcpy(host[(size_t)c].data(), dev[(size_t)c], (size_t)cellElems * 4, hipMemcpyDeviceToHost));
        }
    }
    void destroy() {
        for (size_t c = 0; c < dev.size(); ++c) { (void)hipSetDevice(devices[c]); (void)hipFree(dev[c]); }
        if (desc) CT_OK(cutensorMgDestroyTensorDescriptor(desc));
    }
};

}  // namespace

int main(int argc, char** argv) {
    int visible = 0;
    HIP_OK(hipGetDeviceCount(&visible));
    const bool blog = sample::arg_flag(argc, argv, "--blog");
    std::vector<int32_t> devices;
    std::map<int32_t, int64_t> extent, block;
    std::map<int32_t, int32_t> dc;
    std::vector<int32_t> modesA, modesB, modesC;
    Dist A, B, C;
    if (!blog) {
        int64_t E = 4096, BS = 2048;
        std::vector<int64_t> pos;
        for (int i = 1; i < argc; ++i) {
            const std::string a = argv[i];
            if (a == "--devices" && i + 1 < argc) {
                std::stringstream ss(argv[++i]);
                std::string tok;
                while (std::getline(ss, tok, ',')) devices.push_back(std::atoi(tok.c_str()));
            } else pos.push_back(std::atoll(a.c_str()));
        }
        if (pos.size() > 0) E = pos[0];
        if (pos.size() > 1) BS = pos[1];
        if (devices.empty()) for (int i = 0; i < visible; ++i) devices.push_back(i);        // :129-139
        for (int32_t m : {'i', 'j', 'k'}) { extent[m] = E; block[m] = BS; dc[m] = 2; }          // :154-167
        modesA = {'i', 'k'}; modesB = {'k', 'j'}; modesC = {'i', 'j'};
    } else {
        int nDev = 1, scaling = 2;
        for (int i = 1; i < argc; ++i)
            if (std::string(argv[i]) == "--blog" && i + 2 < argc) { nDev = std::atoi(argv[i + 1]); scaling = std::atoi(argv[i + 2]); }
        // --virtual: the n devices are n logical devices on GPU 0 (the layouts, plans and kernels of an n-GPU run, value-checked
        // on a one-GPU box; the sample itself falls back to i % numDevices the same way, contraction_multi_gpu.cu:159-167)
        const bool virt = sample::arg_flag(argc, argv, "--virtual");
        if (!virt) nDev = std::min(nDev, visible);
        devices.resize((size_t)nDev);
        if (virt) std::fill(devices.begin(), devices.end(), 0); else std::iota(devices.begin(), devices.end(), 0);
        const int32_t M0 = 0, M1 = 1, M2 = 2, N0 = 3, N1 = 4, N2 = 5, K0 = 6, K1 = 7, K2 = 8;
        extent[M0] = 16; extent[M1] = 8 * scaling; extent[M2] = 8; extent[N0] = 16; extent[N1] = 8 * scaling; extent[N2] = 8;
        extent[K0] = 16; extent[K1] = 32; extent[K2] = 8;                                          // blog_post.cu:155-164
        const int nM = nDev >= 4 ? nDev / 2 : nDev, nN = nDev / nM;
        const double M = (double)(extent[M0] * extent[M1] * extent[M2]), N = (double)(extent[N0] * extent[N1] * extent[N2]);
        block[M0] = 16; block[M2] = 8; block[N0] = 16; block[N2] = 8; block[K0] = 16; block[K1] = 16; block[K2] = 8;   // :168-175
        block[M1] = (int64_t)std::ceil(std::ceil(M / std::ceil(M / 4096.0 / nM)) / nM / (double)extent[M0] / (double)extent[M2]);
        block[N1] = (int64_t)std::ceil(std::ceil(N / std::ceil(N / 4096.0 / nN)) / nN / (double)extent[N0] / (double)extent[N1]);   // sic (:173)
        modesA = {K0, M0, M1, K1, M2, K2}; modesB = {K0, N0, K1, N1, K2, N2}; modesC = {M0, N0, M1, N1, M2, N2};
    }
    cutensorMgHandle_t handle;
    CT_OK(cutensorMgCreate(&handle, (uint32_t)devices.size(), devices.data()));

    auto setup = [&](Dist& T, const std::vector<int32_t>& modes, uint32_t seed, bool fill) {
        T.modes = modes;
        for (int32_t m : modes) { T.extent.push_back(extent[m]); T.block.push_back(block[m]); }
        if (!blog) {
            for (int32_t m : modes) T.dcount.push_back(dc[m]);
            const int cells = std::accumulate(T.dcount.begin(), T.dcount.end(), 1, std::multiplies<int>());
            for (int c = 0; c < cells; ++c) T.devices.push_back(devices[(size_t)c % devices.size()]);
        } else {    // blog_post.cu:78-101: from the last mode down, double a mode's device count while blocks and devices remain
            T.dcount.assign(modes.size(), 1);
            int remaining = (int)devices.size();
            bool changed = true;
            while (changed) {
                changed = false;
                for (int i = (int)modes.size() - 1; i >= 0 && remaining > 1; --i) {
                    const int32_t maxCount = (int32_t)(extent[modes[(size_t)i]] / block[modes[(size_t)i]]);
                    if (T.dcount[(size_t)i] < maxCount) { T.dcount[(size_t)i] *= 2; remaining /= 2; changed = true; }
                }
            }
            const int cells = (int)devices.size() / remaining;
            for (int c = 0; c < cells; ++c) T.devices.push_back(devices[(size_t)c]);
        }
        T.create(handle, seed, fill);
    };
    setup(A, modesA, 100, true);
    setup(B, modesB, 200, true);
    setup(C, modesC, 300, false);

    cutensorMgContractionDescriptor_t cdesc;
    CT_OK(cutensorMgCreateContractionDescriptor(handle, &cdesc, A.desc, modesA.data(), B.desc, modesB.data(), C.desc, modesC.data(), C.desc,
                                                modesC.data(), CUTENSOR_COMPUTE_32F));
    cutensorMgContractionFind_t find;
    CT_OK(cutensorMgCreateContractionFind(handle, &find, CUTENSORMG_ALGO_DEFAULT));
    std::vector<int64_t> wsSize(devices.size());
    int64_t wsHost = 0;
    CT_OK(cutensorMgContractionGetWorkspace(handle, cdesc, find, CUTENSOR_WORKSPACE_DEFAULT, wsSize.data(), &wsHost));
    cutensorMgContractionPlan_t plan;
    CT_OK(cutensorMgCreateContractionPlan(handle, &plan, cdesc, find, wsSize.data(), wsHost));
    std::vector<void*> ws(devices.size());
    std::vector<hipStream_t> streams(devices.size());
    for (size_t g = 0; g < devices.size(); ++g) {
        HIP_OK(hipSetDevice(devices[g]));
        HIP_OK(hipMalloc(&ws[g], (size_t)wsSize[g]));
        HIP_OK(hipStreamCreate(&streams[g]));
    }
    const float alpha = 1.f, beta = 0.f;
    int current = 0;
    HIP_OK(hipGetDevice(&current));
    double best = 0;
    for (int rep = 0; rep < 3; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        CT_OK(cutensorMgContraction(handle, plan, &alpha, const_cast<const void**>(A.dev.data()), const_cast<const void**>(B.dev.data()), &beta,
                                    const_cast<const void**>(C.dev.data()), C.dev.data(), ws.data(), nullptr, streams.data()));
        for (int32_t d : devices) { HIP_OK(hipSetDevice(d)); HIP_OK(hipDeviceSynchronize()); }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rep == 0 || ms < best) best = ms;
    }
    HIP_OK(hipSetDevice(current));
    double flops = 2.0;
    for (auto& kv : extent) flops *= (double)kv.second;
    std::printf("execution took: %.3e ms, %.1f GFLOPs/s on %zu device(s)\n", best, flops / (best * 1e-3) / 1e9, devices.size());

    // ---- value check: sampled outputs against fp64 sums over the contracted modes --------------------------------------
    C.download();
    std::vector<int32_t> kModes;
    for (int32_t m : modesA) if (std::find(modesC.begin(), modesC.end(), m) == modesC.end()) kModes.push_back(m);
    std::mt19937 gen(9);
    double worst = 0.0;
    const int samples = 256;
    for (int s = 0; s < samples; ++s) {
        std::map<int32_t, int64_t> at;
        for (int32_t m : modesC) at[m] = (int64_t)(gen() % (uint64_t)extent[m]);
        std::vector<int64_t> kIdx(kModes.size(), 0);
        double acc = 0.0;
        while (true) {
            for (size_t i = 0; i < kModes.size(); ++i) at[kModes[i]] = kIdx[i];
            std::vector<int64_t> ia, ib;
            for (int32_t m : modesA) ia.push_back(at[m]);
            for (int32_t m : modesB) ib.push_back(at[m]);
            acc += (double)A.at(ia) * (double)B.at(ib);
            size_t d = 0;
            while (d < kModes.size() && ++kIdx[d] == extent[kModes[d]]) kIdx[d++] = 0;
            if (d == kModes.size()) break;
        }
        std::vector<int64_t> ic;
        for (int32_t m : modesC) ic.push_back(at[m]);
        worst = std::max(worst, std::fabs((double)C.at(ic) - acc) / std::fabs(acc));
    }
    std::printf("check: %d outputs, max rel err %.3e -> %s\n", samples, worst, worst < 1e-4 ? "ok" : "FAILED");

    for (size_t g = 0; g < devices.size(); ++g) { HIP_OK(hipSetDevice(devices[g])); (void)hipFree(ws[g]); (void)hipStreamDestroy(streams[g]); }
    CT_OK(cutensorMgDestroyContractionPlan(plan));
    CT_OK(cutensorMgDestroyContractionFind(find));
    CT_OK(cutensorMgDestroyContractionDescriptor(cdesc));
    A.destroy(); B.destroy(); C.destroy();
    CT_OK(cutensorMgDestroy(handle));
    return worst < 1e-4 ? 0 : 1;
}
