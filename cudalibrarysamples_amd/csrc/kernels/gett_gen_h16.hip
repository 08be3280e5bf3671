// gett_gen_h16.hip — bf16 / fp16 instantiations of the general MFMA GETT kernel (gett_gen.inc) and the merged table.
//   V = 8: 16-byte lanes on both operands, any K extent (ragged last K-tile), any M / N — 128 x 128 x 64 and 64 x 64 x 64 tiles
//   V = 2: 4-byte pairs (the reference's own test list: extents of 50) — 64 x 64 x 32
//   V = 1: 2-byte gathers, any strides at all — 64 x 64 x 32
#include "gett_gen.inc"

#include <vector>

namespace ctamd {

#define CTAMD_GEN_H16(GE)                      \
    CTAMD_GEN_ORIENTS(GE, 128, 128, 64, 8)     \
    CTAMD_GEN_ORIENTS(GE, 64, 64, 64, 8)       \
    CTAMD_GEN_ORIENTS(GE, 64, 64, 32, 2)       \
    CTAMD_GEN_ORIENTS(GE, 64, 64, 32, 1)

static const GettKernelInfo g_gen_h16_table[] = {CTAMD_GEN_H16(GEN_BF16) CTAMD_GEN_H16(GEN_F16)};

const GettKernelInfo* gett_gen_h16_kernels(int* count) {
    *count = (int)(sizeof(g_gen_h16_table) / sizeof(g_gen_h16_table[0]));
    return g_gen_h16_table;
}

const GettKernelInfo* gett_gen_kernels(int* count) {
    static const std::vector<GettKernelInfo> merged = [] {
        std::vector<GettKernelInfo> v;
        for (auto fn : {&gett_gen_h16_kernels, &gett_gen_f64_kernels, &gett_gen_cplx_kernels}) {
            int n = 0;
            const GettKernelInfo* t = fn(&n);
            v.insert(v.end(), t, t + n);
        }
        return v;
    }();
    *count = (int)merged.size();
    return merged.data();
}

}  // namespace ctamd
