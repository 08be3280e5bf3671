// gett_h16p_layout.h — index arithmetic of gett_h16w4p_kernel's transposed epilogue image (gett_h16p.hip), plain C++ so that the
// kernel and tests/harness/h16p_layout_harness.cpp compile the SAME functions (the harness replays a pass on the CPU: every element
// of the 16 x 128 pass reaches the lane and register that stores it, and no LDS access of the pass has a bank conflict).
//
// A pass = rows 16 i + [0, 16) of a wave's 128 x 128 quadrant = the eight accumulator fragments acc[i][0..7]; fragment j holds, in
// lane (g = lane >> 4, cl = lane & 15), rows 4 g + [0, 4) of column 16 j + cl.  Image: [128 columns][16 rows] of 16-bit values,
// 32 bytes per column: column c lives at image row R(c) = c ^ 4 ((c >> 3) & 1), and inside it rows 4 s + [0, 4) (8 bytes) at slot
// s ^ ((R(c) >> 2) & 3).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CTAMD_HD __host__ __device__ inline
#else
#define CTAMD_HD inline
#endif

namespace ctamd {

constexpr int kPImgBytes = 4096;          // one pass of one wave

// byte offset of the 8 bytes lane writes for fragment j (ds_write_b64; fragments j and j + 1 are 512 bytes apart)
CTAMD_HD uint32_t p_img_write_off(int lane, int j) {
    const int g = lane >> 4, cl = lane & 15;
    const int rw = cl ^ (((cl >> 3) & 1) << 2);                       // R(16 j + cl) = 16 j + rw
    return (uint32_t)(512 * j + rw * 32 + ((g ^ ((rw >> 2) & 3)) << 3));
}
// byte offset of the 8 bytes lane supplies to the transposing read (ds_read_b64_tr_b16) of chunk q = 4 it + (lane >> 4) — columns
// 8 q + [0, 8) of the pass's 16 rows — half h (columns 8 q + 4 h + [0, 4)): the lane addresses column 8 q + 4 h + (cl >> 2), rows
// 4 (cl & 3) + [0, 4); after the read it holds columns 8 q + 4 h + [0, 4) of row cl
CTAMD_HD uint32_t p_img_read_off(int lane, int it, int h) {
    const int g = lane >> 4, cl = lane & 15, ge = g & 1;
    const int row = (4 * (h ^ ge) + (cl >> 2));                      // R(8 q + 4 h + (cl >> 2)) - 8 q: bit 3 of the column is q & 1 = g & 1
    const int slot = (cl & 3) ^ ((2 * ge + (h ^ ge)) & 3);           // ((R >> 2) & 3) = (2 q + (h ^ ge)) & 3, 2 q = 2 g (mod 4) = 2 ge (mod 4)
    return (uint32_t)(1024 * it + 256 * g + row * 32 + (slot << 3));
}

// ---- second image (EP = 2): the pass again as [16 rows][256 bytes], so that ONE store instruction writes 4 rows x 256 contiguous bytes
// (whole cache lines) instead of the 16 rows x 64 bytes the transposing reads deliver.  16-byte unit u of row r lives at unit
// u ^ (r & 7) of the row: the ds_write_b128 of 8 consecutive lanes (8 rows, one chunk) and the ds_read_b128 of a 16-lane group both
// spread over all banks.
// where lane (g = lane >> 4, ii = lane & 15) parks the chunk q = 4 it + g it received from the transposing reads
CTAMD_HD uint32_t p_row_write_off(int lane, int it) {
    const int g = lane >> 4, ii = lane & 15, q = 4 * it + g;
    return (uint32_t)(ii * 256 + ((q ^ (ii & 7)) << 4));
}
// where lane (r4 = lane >> 4, ch = lane & 15) fetches 8 columns (8 ch + [0, 8)) of row 4 it2 + r4 for its store
CTAMD_HD uint32_t p_row_read_off(int lane, int it2) {
    const int row = 4 * it2 + (lane >> 4), ch = lane & 15;
    return (uint32_t)(row * 256 + ((ch ^ (row & 7)) << 4));
}

// ---- C image (beta != 0, round 6): the pass's 16 rows x 128 columns of C, row-major in the row image's 4 KiB while that image is idle
// (between the fetch of pass I - 1 and the park of pass I).  C arrives the way D leaves — lane (r4 = lane >> 4, ch = lane & 15) holds the
// 16-byte chunk ch of row 4 it + r4 — and has to reach the accumulator layout: fragment j, lane (g, cl) = rows 4 g + [0, 4) of column
// 16 j + cl.  That is the transposing read again, the other way round: four ROWS are the lines, a fragment's 16 columns the elements.
// 32-byte pair P (columns 16 P + [0, 16) = fragment P) of row r lives at pair P ^ (r & 7): the 8 rows a 32-lane half of the read touches
// (4 g + (cl >> 2), g = 0, 1 or 2, 3) sit in 8 different pairs = all 64 banks once; 8 consecutive lanes of the ds_write_b128 (one row,
// chunks 0-7 or 8-15) cover four whole pairs that differ mod 128 bytes.
// where lane (r4, ch) writes the chunk it loaded from C for row 4 it + r4
CTAMD_HD uint32_t p_c_write_off(int lane, int it) {
    const int row = 4 * it + (lane >> 4), ch = lane & 15;
    return (uint32_t)(row * 256 + (((ch >> 1) ^ (row & 7)) << 5) + ((ch & 1) << 4));
}
// the 8 bytes lane (g, cl) supplies to the transposing read of fragment j: row 4 g + (cl >> 2), columns 16 j + 4 (cl & 3) + [0, 4); after
// the read it holds column 16 j + cl of rows 4 g + [0, 4)
CTAMD_HD uint32_t p_c_read_off(int lane, int j) {
    const int g = lane >> 4, cl = lane & 15, row = 4 * g + (cl >> 2);
    return (uint32_t)(row * 256 + ((j ^ (row & 7)) << 5) + ((cl & 3) << 3));
}

}  // namespace ctamd
