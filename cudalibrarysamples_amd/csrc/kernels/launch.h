// launch.h — host-visible launch interface of the gfx950 kernels (implemented in the .hip files).
#pragma once
#include <hip/hip_runtime_api.h>

#include "params.h"

namespace ctamd {

enum OperandLayout : int {
    LAY_F = 0,  // fastest free mode has stride 1: 16-byte lanes along rows
    LAY_K = 1,  // fastest contracted mode has stride 1: 16-byte lanes along k
    LAY_S = 2   // arbitrary strides: 4-byte gathers
};

struct GettKernelInfo {
    int bm, bn, bk;      // workgroup tile
    int wm, wn, wk;      // wave grid inside the workgroup
    int layA, layB;      // OperandLayout of kernel-A / kernel-B
    int threads;
    int pf;              // K-tiles in flight in registers
    int kfast;           // 1: requires extent(fastest K mode) % bk == 0
    int ablation;        // != 0: measurement-only variant (wrong results), never ranked by default
    hipError_t (*launch)(const GettParams&, hipStream_t);
    int fragPartials;    // 1: split-K partials are written in accumulator order (padded tiles), folded by
                         //    launch_splitk_reduce_frag; 0: row-major [M][N], launch_splitk_reduce
    int nt;              // 1: operands are streamed with the nontemporal policy (no Infinity-Cache allocation): ranked only for
                         //    problems that read every operand byte once and whose operands exceed the Infinity Cache anyway
    int elem;            // general family (gett_gen_kernels): GenElem of the instantiation
    int vec;             // general family: elements per staged unit = per global load (both operands)
};

// element types of the general MFMA family (gett_gen.inc)
enum GenElem : int { GEN_BF16 = 0, GEN_F16 = 1, GEN_F64 = 2, GEN_C32 = 3, GEN_C64 = 4 };

// fp32 data, fp32 MFMA (v_mfma_f32_16x16x4_f32)
const GettKernelInfo* gett_f32_kernels(int* count);
hipError_t launch_splitk_reduce(const SplitKReduceParams& p, hipStream_t stream);
hipError_t launch_splitk_reduce_frag(const SplitKReduceParams& p, hipStream_t stream);
// streaming (LDS-DMA ring) kernels, gett_f32_stream.hip; gett_f32_kernels() returns the merged table
const GettKernelInfo* gett_f32_stream_kernels(int* count);

// bf16 / fp16 data, fp32 accumulation (v_mfma_f32_32x32x16_{bf16,f16}), gett_h16.hip
const GettKernelInfo* gett_h16_kernels(int* count);
const GettKernelInfo* gett_h16v_kernels(int* count);   // gett_h16v.hip: appended to the table above as entries 40..87
const GettKernelInfo* gett_h16p_kernels(int* count);   // gett_h16p.hip (persistent 256 x 256 kernel): entries 88..95

// general MFMA family: bf16 / fp16 shapes the aligned kernels above refuse (no 16-byte lanes, K not in whole 64-deep tiles), fp64,
// complex64 / complex128 — register-staged, any strides and extents (gett_gen.inc; the table is the concatenation of the three
// translation units gett_gen_h16.hip / gett_gen_f64.hip / gett_gen_cplx.hip)
const GettKernelInfo* gett_gen_kernels(int* count);
const GettKernelInfo* gett_gen_h16_kernels(int* count);
const GettKernelInfo* gett_gen_f64_kernels(int* count);
const GettKernelInfo* gett_gen_cplx_kernels(int* count);
// split-K fold of the general family's fp64 / complex kernels: D = alpha * sum_s partial[s] + beta * op(C); partials
// [slice][L][M][N] in the accumulator type of `elem` (GEN_F64: double, GEN_C32: float2, GEN_C64: double2)
hipError_t launch_gen_splitk_reduce(const SplitKReduceParams& p, int elem, hipStream_t stream);

// simple one-thread-per-output contraction for every other dtype (and > kMaxGroupModes problems)
hipError_t launch_gett_simple(const GettParams& p, int dtype /*hipDataType*/, bool accumulate64,
                              hipStream_t stream);

// any number of modes (mode table in device memory), one output element per lane
hipError_t launch_gett_wide(const WideParams& p, int dtype /*hipDataType*/, bool accumulate64, hipStream_t stream);

// element-wise family (elementwise.hip)
enum EwVariant : int {
    EW_TRANSPOSE = 0,  // sD0 == 1 and sA1 == 1: 64x64 LDS tile, 16-byte lanes on both sides
    EW_ROWCOPY   = 1,  // sD0 == 1 and sA0 == 1: 16-byte lanes along dim0, no LDS
    EW_GENERIC   = 2,  // any strides, any dtype: one element per lane
    EW_TRANSPOSE_ANY = 4,  // pure permutation of 2- / 4-byte elements, D contiguous along dim0 and A along dim1, any extents / alignment: 64 x 64 LDS tile, element-wise
    EW_BLOCK     = 3   // pure permutation of 2- / 4-byte elements whose leading modes are the same packed set in A and D: contiguous blocks
                       // through LDS, permuted inside (Ew2DParams::blk*); falls back to EW_GENERIC when a C / E / X operand is attached
};
hipError_t launch_elementwise(const Ew2DParams& p, int variant, int dtype, hipStream_t stream);
// D[0 .. n) = value (contiguous; the padded-permutation border fill)
hipError_t launch_fill(void* D, uint64_t n, int dtype, double value, hipStream_t stream);

// reduction family (reduce.hip)
enum ReduceVariant : int {
    RED_COL     = 0,   // A's stride-1 mode is kept: lanes along it (float4), loop over reduced modes
    RED_ROW     = 1,   // A's stride-1 mode is reduced: one workgroup per kept element
    RED_GENERIC = 2    // any strides, any dtype
};
hipError_t launch_reduce(const ReduceParams& p, int variant, int dtype, bool acc64, hipStream_t stream);
hipError_t launch_reduce_finalize(const ReduceParams& p, int dtype, bool acc64, hipStream_t stream);

}  // namespace ctamd
