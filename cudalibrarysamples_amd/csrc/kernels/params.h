// params.h — plain-old-data kernel arguments shared by the host planner and the gfx950 kernels.
//
// A tensor contraction is executed as a GETT: the modes of A, B and C are classified into four
// groups — L (batch: in A, B and C), M (free: A and C), N (free: B and C), K (contracted: A and B) —
// and each group is linearised as a mixed-radix number, fastest mode first.  A group index is turned
// into an element offset inside each tensor by peeling its digits with a multiply-high division and
// multiplying by the per-tensor stride.  No tensor is ever physically transposed or copied.
#pragma once
#include <stdint.h>

namespace ctamd {

constexpr int kMaxGroupModes = 4;   // modes per group after fusion (more => NOT_SUPPORTED)

// Exact unsigned division n / d for n < 2^31, d in [2, 2^31): q = mulhi(n, magic) >> shift.
struct FastDiv {
    uint32_t d;
    uint32_t magic;
    uint32_t shift;
};

// One mode group.  stride[t][i]: element stride of mode i in tensor slot t.
//   M group: slot 0 = A, slot 1 = C/D          N group: slot 0 = B, slot 1 = C/D
//   K group: slot 0 = A, slot 1 = B            L group: slot 0 = A, slot 1 = B, slot 2 = C/D
struct ModeGroup {
    int32_t  n;                          // number of real modes; entries n.. are padding {d=1, magic=0, stride=0}
    uint32_t total;                      // product of extents (< 2^31)
    FastDiv  div[kMaxGroupModes];
    int64_t  stride[3][kMaxGroupModes];
};

struct GettParams {
    const void* A;
    const void* B;
    const void* C;        // source for beta (may alias D)
    void*       D;
    float*      partial;  // split-K workspace: [slice][L][M][N] fp32, or nullptr
    unsigned long long* timing;   // diagnostics: 16 x uint64 per workgroup (nullptr = off)
    uint32_t*   sync;     // {arrivals, departures} of an in-launch split-K fold (streaming kernels), or nullptr
    unsigned long long xcdTiles;  // balanced split-K (streaming kernels): byte x = K-tiles per slice on XCD x; 0 = uniform
    ModeGroup   gM, gN, gK, gL;
    float       alpha, beta;
    double      alpha64, beta64;  // same scalars at full width (fp64 data)
    uint32_t    tilesM, tilesN;   // number of output tiles
    uint32_t    splitK;           // number of K slices (>= 1)
    uint32_t    kPerSlice;        // K elements per slice, multiple of the kernel's BK
    uint32_t    nBlocks;          // total workgroups = tilesM*tilesN*splitK*L
    // C and D may have different strides only through separate descriptors; the engine requires
    // identical mode order/extents (as the ABI does) and carries D strides in slot 1 / 2 above and
    // C strides here.
    int64_t     cStrideM[kMaxGroupModes];
    int64_t     cStrideN[kMaxGroupModes];
    int64_t     cStrideL[kMaxGroupModes];
    // complex data (general MFMA family, gett_gen.inc): imaginary parts of the scalars (real parts in alpha64 / beta64),
    // conjugation of kernel-A / kernel-B / C
    double      alphaIm, betaIm;
    int32_t     conjA, conjB, conjC;
    // measurement switch (CUTENSOR_AMD_PARTIAL_STORE): cache policy of the streaming fp32 kernels' split-K partial stores —
    // 0 = write-through (sc1, the default), 1 = plain (write-back: the lines stay dirty in the XCD's L2 until the kernel ends),
    // 2 = nontemporal
    int32_t     partialPolicy;
    // Operands without 16-byte lanes on the LDS-DMA kernels (round 6).  endA / endB: byte address one past the last element of
    // kernel-A / kernel-B (set by cutensorContract from the plan's element spans): a 16-byte unit that would read past it is staged
    // masked and patched element by element (gett_h16x_common.h, x_rag_fix).  ragged: bit 0 — the launch needs the kernels' RAG
    // instantiation although K holds whole K-tiles (a free-contiguous operand whose row units can straddle the end of the tensor).
    // bit 1 — sweep-ragged K (several contracted modes, the fastest one without whole K-tiles: the last K-tile of every sweep of that mode
    // is staged masked, gett_h16x_common.h x_rag_toggle); bits 2..31 then hold the K-tile count of the padded index space.
    unsigned long long endA, endB;
    uint32_t    ragged;
    // Origin of this launch's tile grid inside the M x N index space (elements): a plan may cover the output with an interior launch
    // of large tiles and edge strips of small ones (plan_contraction.cpp, strip plans); tile (mt, nt) starts at (mOrg + mt BM, nOrg + nt BN).
    uint32_t    mOrg, nOrg;
    // A second tile rectangle in the same launch (gett_h16w4q_kernel only — the strip kernel): tile ids tilesM * tilesN .. address
    // tilesM2 x tilesN2 tiles from (mOrg2, nOrg2); tilesM2 * tilesN2 == 0: none.  The two edge strips of an output (rows past the
    // interior, columns past the interior) are ONE launch that way.
    uint32_t    tilesM2, tilesN2, mOrg2, nOrg2;
};

// ---------------------------------------------------------------------------------------------
// Contractions with more unfusable modes per group than kMaxGroupModes (e.g. the 25-mode extent-2 tensors of
// cuTENSOR/contraction_jit.cu:50-56): the mode list lives in device memory instead of the argument block.
// Entries [0, nOut) are the output modes in any order, entries [nOut, nOut + nK) the contracted modes.
// ---------------------------------------------------------------------------------------------
struct WideMode {
    FastDiv div;                 // extent as divisor (contracted modes are decoded with it; d = extent)
    int64_t sA, sB, sC, sD;      // element strides (0 where the tensor does not carry the mode)
};
struct WideParams {
    const void* A;
    const void* B;
    const void* C;
    void*       D;
    const WideMode* modes;       // device memory, owned by the plan
    uint32_t    nOut, nK;
    uint64_t    outTotal;        // product of the output extents
    uint32_t    kTotal;          // product of the contracted extents (< 2^31)
    float       alpha, beta;
    double      alpha64, beta64;
    // complex data (HIP_C_32F / HIP_C_64F): imaginary parts of the scalars, conjugation of each input
    double      alphaIm, betaIm;
    int32_t     conjA, conjB, conjC;
};

// Second stage of split-K: D = alpha * sum_s partial[s] + beta * C
struct SplitKReduceParams {
    const float* partial;
    const void*  C;
    void*        D;
    ModeGroup    gM, gN, gL;     // slot 1 (M,N) / slot 2 (L) hold D strides
    int64_t      cStrideM[kMaxGroupModes];
    int64_t      cStrideN[kMaxGroupModes];
    int64_t      cStrideL[kMaxGroupModes];
    float        alpha, beta;
    uint32_t     splitK;
    // accumulator-order partials (streaming kernels): output tile grid and 16x16 fragments per wave
    uint32_t     tilesM, tilesN;
    uint32_t     fragTM, fragTN;
    int32_t      outType;        // C / D element type of the row-major fold: 0 = fp32, 1 = bf16, 2 = fp16
    // general MFMA family, fp64 / complex data (launch_gen_splitk_reduce): partials in the accumulator type (double, float2,
    // double2), scalars at full width, conjugation of C
    double       alpha64, beta64, alphaIm, betaIm;
    int32_t      conjC;
};

// ---------------------------------------------------------------------------------------------
// Element-wise family (cutensorPermute / cutensorElementwiseBinaryExecute and the permutation-only
// form of cutensorReduce):  D = alpha * perm(A) [+ gamma * perm(C)].
// The planner picks two "tile" modes — dim0 = D's fastest mode, dim1 = A's fastest mode (or D's
// second mode when A and D share the fastest one) — and linearises all remaining modes into
// `rest`.  A workgroup owns one T0 x T1 tile of one rest index.
// ---------------------------------------------------------------------------------------------
struct Ew2DParams {
    const void* A;
    const void* C;                 // may be nullptr (gamma == 0)
    void*       D;
    uint32_t    E0, E1;            // extents of the two tile modes (E1 = 1 when there is none)
    int64_t     sA0, sA1, sD0, sD1, sC0, sC1;
    uint32_t    tiles0, tiles1;
    uint32_t    tile0;             // EW_TRANSPOSE fp32: tile extent along dim0 (64 / 128 / 256), chosen by the planner
    uint32_t    tile1;             // EW_TRANSPOSE 16-bit wide kernel: tile extent along dim1 (64 / 128)
    uint32_t    order;             // EW_TRANSPOSE: 0 = tile ids walk dim0, dim1, rest; 1 = rest, dim1, dim0 with each XCD
                                   // (workgroup id % 8) owning one contiguous eighth of that sequence (elementwise.hip)
    uint32_t    idsPerXcd;         // order 1: ceil(nBlocks / 8)
    FastDiv     divRest;           // order 1: division by rest.total
    FastDiv     divTiles0, divTiles1;
    ModeGroup   rest;              // slot 0 = A, slot 1 = D, slot 2 = C strides
    uint32_t    nBlocks;
    float       alpha, gamma;
    double      alpha64, gamma64;
    // Trinary form (cutensorElementwiseTrinaryExecute): D = opAC(opAB(delta * E, alpha * perm(A)), gamma * perm(C))
    // where E has D's strides (E = D for the in-place second pass, or the operand that already has D's
    // layout); E == nullptr: D = opAC(alpha * perm(A), gamma * perm(C)).  Operators: cutensorOperator_t
    // ADD / MUL / MAX / MIN (0 is read as ADD).
    const void* E;
    float       delta;
    double      delta64;
    int32_t     opAB, opAC;
    // Second permuted operand (element-wise trinary whose A and B are both permuted but share the tile modes):
    // D = opAC(opAB(xi * perm(X), alpha * perm(A)), gamma * perm(C)); X == nullptr: unused.  X is walked with its own
    // strides through the same tile decomposition as A (a second LDS tile in the transposing variant).
    const void* X;
    int64_t     sX0, sX1;
    int64_t     restX[kMaxGroupModes];
    float       xi;
    double      xi64;
    // complex data (HIP_C_32F / HIP_C_64F, ew_generic_cplx_kernel): imaginary parts of alpha / gamma (alpha64 / gamma64 hold the
    // real parts) and conjugation of the permuted operand A / of C
    double      alphaIm, gammaIm;
    int32_t     conjA, conjC;
    // EW_BLOCK (round 6, ew_block_kernel): the first blkN modes of D (packed: D-relative offset f = their mixed-radix index) are the SAME
    // set of modes that is packed at the front of A — a block of blkTotal elements that is contiguous on both sides and only permuted
    // inside ([d, c, b | a] -> [b, c, d | a]: the copy in front of a contraction, api.cpp plan_repack).  blkDiv: extents in D's order;
    // blkSrc: each mode's stride in A = its position stride inside the block as it is loaded; blkRest: every other mode (slot 0 = A,
    // slot 1 = D); a workgroup moves blkGroup consecutive rest indices; blkBlocks workgroups.  blkN == 0: no such form.
    uint32_t    blkN, blkTotal, blkGroup, blkBlocks;
    uint32_t    blkVec;            // 16-byte lanes on both sides: block size and every rest stride a multiple of the lane's elements
    FastDiv     blkDiv[4];
    uint32_t    blkSrc[4];
    ModeGroup   blkRest;
};

// ---------------------------------------------------------------------------------------------
// Reduction (cutensorReduce):  D[kept] = alpha * reduce_{red} A[kept, red] + beta * C[kept].
// ---------------------------------------------------------------------------------------------
struct ReduceParams {
    const void* A;
    const void* C;
    void*       D;
    void*       partial;           // [splitR][keptTotal] accumulators (float or double), or nullptr
    ModeGroup   kept;              // slot 0 = A, slot 1 = D, slot 2 = C strides
    ModeGroup   red;               // slot 0 = A strides
    uint32_t    splitR;            // reduction range split over this many workgroups
    uint32_t    redPerSplit;       // reduced elements per split (multiple of 4)
    int32_t     op;                // cutensorOperator_t: ADD / MUL / MIN / MAX
    float       alpha, beta;
    double      alpha64, beta64;
    // complex data (reduce_generic_cplx_kernel): imaginary parts of the scalars, conjugation of A / C; partials are (re, im) pairs
    double      alphaIm, betaIm;
    int32_t     conjA, conjC;
    // RED_GENERIC on real data with A's stride-1 mode among the REDUCED ones (round 6): one wave per kept element, lanes along that mode
    // (reduce_row_any_kernel) instead of one lane per kept element
    uint32_t    rowAny;
};

}  // namespace ctamd
