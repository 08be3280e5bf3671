// gett_gen_layout.h — index arithmetic of the general MFMA GETT kernels (gett_gen.inc): the LDS image of an operand tile, which
// thread stages which piece of it, and where a lane finds its MFMA fragment.  Plain C++ (no HIP), so that the same functions are
// compiled into the kernels AND into a host harness that replays a tile on the CPU (tests/test_gen_layout_cpu.py): every
// staged element must be the one the fragment read expects, for every element type / tile / orientation / vector width the
// kernel table instantiates.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CTAMD_HD __host__ __device__ __forceinline__
#else
#define CTAMD_HD inline
#endif

namespace ctamd {

// LDS image of one operand tile: [row][BK elements of ES bytes], rows of 128 or 64 bytes, the 16-byte units of a row
// XOR-swizzled with the row index (period 16 rows) so that the 16-row fragment reads (ds_read_b128: 16 lanes = 16 rows x one unit per cycle)
// and the staging writes spread over all banks.  The swizzle has a period of 16 rows: a fragment's per-lane byte offset does
// not depend on which 16-row block of the tile it reads.
template <int ES_, int BK_>
struct GenImage {
    static constexpr int ES  = ES_;           // element bytes: 2 (bf16 / fp16), 8 (fp64, complex64), 16 (complex128)
    static constexpr int BK  = BK_;
    static constexpr int RB  = BK * ES;       // bytes per row
    static constexpr int UR  = RB / 16;       // 16-byte units per row
    static constexpr int RPB = 256 / RB;      // rows per 256 bytes (one pass over all 64 banks)
    static_assert(RB == 128 || RB == 64, "rows of 64 or 128 bytes");
    // Conflict-free for the fragment reads below on gfx950's ds_read_b128 lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...;
    // found by exhaustive search over the GF(2)-linear maps of the row's low four bits, tools/lds_swizzle_search.py):
    //   128-byte rows, one unit per lane (16-bit):      unit ^ ((row >> 1) & 7)
    //   128-byte rows, two units per lane (8 / 16-byte elements): unit ^ (((row >> 1) & 1) | (row & 4))
    //   64-byte rows (16-bit, BK = 32):                 unit ^ (((row >> 2) & 1) << 1)
    static CTAMD_HD int sw(int row) {
        if (RB == 64) return ((row >> 2) & 1) << 1;
        return ES == 2 ? ((row >> 1) & 7) : (((row >> 1) & 1) | (row & 4));
    }
    // byte address of element (row, k) of the tile
    static CTAMD_HD int addr(int row, int k) {
        const int kb = k * ES;
        return row * RB + ((((kb >> 4) ^ sw(row)) & (UR - 1)) << 4) + (kb & 15);
    }
    // byte address of the 16-byte unit `unit` (logical index inside the row) of row `r` of a 16-row block starting at row rb
    static CTAMD_HD int unit_addr(int rb, int r, int unit) { return (rb + r) * RB + (((unit ^ sw(r)) & (UR - 1)) << 4); }
};

// Which pieces of a ROWS x BK tile thread `tid` of THREADS stages.  A piece ("unit") is V consecutive elements along the
// operand's contiguous direction, fetched by ONE global load of V * ES bytes:
//   ORIENT = 1 (K-contiguous operand): V elements along k.   unit i of a thread: row tid / KV + i * RSTEP, k (tid % KV) * V
//   ORIENT = 0 (free-contiguous):      V elements along rows. unit i: rows ((tid % TPK) + i * TPK) * V + [0, V), k tid / TPK
// Either way ALL units of a thread share one k (one contracted-index decode per thread per K-tile) and neighbouring lanes read
// neighbouring addresses.
template <int ORIENT, int ROWS, int BK, int V, int THREADS>
struct GenUnitMap {
    static constexpr int KV    = BK / V;
    static constexpr int RV    = ROWS / V;
    static constexpr int NU    = ROWS * BK / (V * THREADS);
    static constexpr int TPK   = THREADS / BK;            // ORIENT 0: threads per k-row
    static constexpr int RSTEP = THREADS / KV;            // ORIENT 1: rows between two units of a thread
    static_assert(BK % V == 0 && ROWS % V == 0, "whole units");
    static_assert(NU >= 1 && NU * V * THREADS == ROWS * BK, "the tile divides evenly over the workgroup");
    static_assert(ORIENT ? (THREADS % KV == 0) : (THREADS % BK == 0 && RV % (THREADS / BK) == 0), "every unit of a thread has the same k");
    static CTAMD_HD int unit_row(int tid, int i) { return ORIENT ? tid / KV + i * RSTEP : ((tid % TPK) + i * TPK) * V; }
    static CTAMD_HD int unit_k(int tid)          { return ORIENT ? (tid % KV) * V : tid / TPK; }
};

// MFMA fragment of a 16-row block for k-block s of the tile: lane (r = lane & 15, q = lane >> 4) reads UPL consecutive 16-byte
// units, logical unit (4 s + q) * UPL + h, h < UPL, of row r.  The elements it gets are k = ((4 s + q) * UPL + h) * (16 / ES) + e.
//   16-bit: UPL = 1 — 8 consecutive k for one v_mfma_f32_16x16x32_{bf16,f16} (A[i = r][k = 8 q + e], B likewise)
//   fp64 / complex64: UPL = 2 — 4 elements k = 4 q + j; step j of the k-block (v_mfma_*_16x16x4: A[i = r][k = q]) takes element j
//   complex128: UPL = 2 — 2 elements k = 2 q + j
// A and B use the same (q, j) -> k assignment, which is all the products need.
template <int ES>
struct GenFrag {
    static constexpr int UPL = (ES == 2) ? 1 : 2;
    static constexpr int EPU = 16 / ES;                 // elements per 16-byte unit
    static constexpr int KPB = 4 * UPL * EPU;           // k per k-block: 32 (16-bit), 16 (8-byte elements), 8 (complex128)
    static CTAMD_HD int unit(int s, int q, int h) { return (4 * s + q) * UPL + h; }
    static CTAMD_HD int k_of(int s, int q, int h, int e) { return unit(s, q, h) * EPU + e; }
};

// Row of accumulator register t of lane-group q inside a 16 x 16 fragment (column = lane & 15):
// fp32 accumulators (v_mfma_f32_16x16x*): 4 q + t;  fp64 (v_mfma_f64_16x16x4_f64): q + 4 t.
template <bool ACC64>
CTAMD_HD int gen_acc_row(int q, int t) { return ACC64 ? q + 4 * t : 4 * q + t; }

}  // namespace ctamd
