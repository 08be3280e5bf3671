// gett_gen_cplx.hip — complex64 / complex128 instantiations of the general MFMA GETT kernel (gett_gen.inc): four real MFMAs per
// complex product into three accumulators (Re·Re, Im·Im, Re·Im + Im·Re); interleaved (re, im) elements stay interleaved in LDS.
//   complex64:  V = 2 (16-byte lanes) 128 x 64 x 16 and 64 x 64 x 16;  V = 1 (8-byte gathers) 64 x 64 x 16
//   complex128: V = 1 (one element = 16 bytes) 64 x 64 x 8
#include "gett_gen.inc"

namespace ctamd {

static const GettKernelInfo g_gen_cplx_table[] = {
    CTAMD_GEN_ORIENTS(GEN_C32, 128, 64, 16, 2)
    CTAMD_GEN_ORIENTS(GEN_C32, 64, 64, 16, 2)
    CTAMD_GEN_ORIENTS(GEN_C32, 64, 64, 16, 1)
    CTAMD_GEN_ORIENTS(GEN_C64, 64, 64, 8, 1)};

const GettKernelInfo* gett_gen_cplx_kernels(int* count) {
    *count = (int)(sizeof(g_gen_cplx_table) / sizeof(g_gen_cplx_table[0]));
    return g_gen_cplx_table;
}

}  // namespace ctamd
