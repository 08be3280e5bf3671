// h16p_layout_harness.cpp — CPU replay of one epilogue pass of gett_h16w4p_kernel (csrc/kernels/gett_h16p_layout.h, the functions the
// kernel itself compiles): 64 lanes write the eight fragments of a 16 x 128 pass into the transposed 4-KiB image, read it back with
// the transposing read's semantics and must end up with, per lane and chunk, eight consecutive columns of ONE row — what a 16-byte
// store to D needs.  Also: every byte of the image written exactly once, and no bank conflict in the ds_write_b64 of a 16-lane group
// (bank = (address / 4) mod 32) nor in the ds_read_b64_tr_b16 of a 32-lane half (bank = (address / 4) mod 64) — MI355X guide, LDS table.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <set>

#include "gett_h16p_layout.h"

using namespace ctamd;

int main() {
    uint16_t img[kPImgBytes / 2];
    int writes[kPImgBytes];
    std::memset(writes, 0, sizeof(writes));
    std::memset(img, 0xff, sizeof(img));
    auto code = [](int row, int col) { return (uint16_t)(row * 128 + col); };
    // ---- write side: fragment j, lane (g, cl): rows 4 g + r, column 16 j + cl
    for (int j = 0; j < 8; ++j) {
        for (int grp = 0; grp < 4; ++grp) {                  // ds_write_b64: lane groups of 16 contiguous lanes
            std::set<int> banks;
            for (int lane = 16 * grp; lane < 16 * grp + 16; ++lane) {
                const uint32_t off = p_img_write_off(lane, j);
                if (off % 8 != 0 || off + 8 > (uint32_t)kPImgBytes) { std::printf("write offset out of range / misaligned\n"); return 1; }
                for (int r = 0; r < 4; ++r) img[off / 2 + r] = code(4 * (lane >> 4) + r, 16 * j + (lane & 15));
                for (int b = 0; b < 8; ++b) ++writes[off + b];
                banks.insert((off / 4) % 32);
                banks.insert((off / 4 + 1) % 32);
            }
            if (banks.size() != 32) { std::printf("ds_write_b64 bank conflict: fragment %d lane group %d touches %zu banks\n", j, grp, banks.size()); return 2; }
        }
    }
    for (int b = 0; b < kPImgBytes; ++b)
        if (writes[b] != 1) { std::printf("image byte %d written %d times\n", b, writes[b]); return 3; }
    // ---- read side: iteration it, half h; within a 16-lane group the block M[kk][r] = what lane 4 kk + (r >> 2) loaded, element r & 3;
    //      output lane i = {M[0][i], M[1][i], M[2][i], M[3][i]}
    for (int it = 0; it < 4; ++it) {
        uint16_t out[64][8];
        for (int h = 0; h < 2; ++h) {
            for (int half = 0; half < 2; ++half) {           // ds_read_b64_tr_b16: 2 x 32 lanes
                std::set<int> banks;
                for (int lane = 32 * half; lane < 32 * half + 32; ++lane) {
                    const uint32_t off = p_img_read_off(lane, it, h);
                    if (off % 8 != 0 || off + 8 > (uint32_t)kPImgBytes) { std::printf("read offset out of range / misaligned\n"); return 4; }
                    banks.insert((off / 4) % 64);
                    banks.insert((off / 4 + 1) % 64);
                }
                if (banks.size() != 64) { std::printf("ds_read_b64_tr_b16 bank conflict: it %d half %d lanes %d.. touch %zu banks\n", it, h, 32 * half, banks.size()); return 5; }
            }
            for (int grp = 0; grp < 4; ++grp)
                for (int i = 0; i < 16; ++i)
                    for (int kk = 0; kk < 4; ++kk) {
                        const int src = 16 * grp + 4 * kk + (i >> 2);
                        out[16 * grp + i][4 * h + kk] = img[p_img_read_off(src, it, h) / 2 + (i & 3)];
                    }
        }
        for (int lane = 0; lane < 64; ++lane) {
            const int row = lane & 15, q = 4 * it + (lane >> 4);
            for (int e = 0; e < 8; ++e)
                if (out[lane][e] != code(row, 8 * q + e)) {
                    std::printf("lane %d it %d element %d: got (row %d, col %d), want (row %d, col %d)\n", lane, it, e, out[lane][e] / 128,
                                out[lane][e] % 128, row, 8 * q + e);
                    return 6;
                }
        }
    }
    // ---- second image (EP = 2): every lane parks its four chunks (rows = lane & 15) in the row image and fetches 4 rows x 16 chunks back
    {
        uint16_t row_img[16 * 128];
        std::memset(row_img, 0xff, sizeof(row_img));
        int w2[4096];
        std::memset(w2, 0, sizeof(w2));
        for (int it = 0; it < 4; ++it)
            for (int grp8 = 0; grp8 < 8; ++grp8) {          // ds_write_b128: 8 x 8 contiguous lanes, bank = (address / 4) mod 32
                std::set<int> banks;
                for (int lane = 8 * grp8; lane < 8 * grp8 + 8; ++lane) {
                    const uint32_t off = p_row_write_off(lane, it);
                    if (off % 16 != 0 || off + 16 > 4096u) { std::printf("row image write offset\n"); return 7; }
                    const int row = lane & 15, q = 4 * it + (lane >> 4);
                    for (int e = 0; e < 8; ++e) row_img[off / 2 + e] = code(row, 8 * q + e);
                    for (int b = 0; b < 16; ++b) ++w2[off + b];
                    for (int b = 0; b < 4; ++b) banks.insert((off / 4 + b) % 32);
                }
                if (banks.size() != 32) { std::printf("ds_write_b128 bank conflict: it %d lanes %d.. touch %zu banks\n", it, 8 * grp8, banks.size()); return 8; }
            }
        for (int b = 0; b < 4096; ++b)
            if (w2[b] != 1) { std::printf("row image byte %d written %d times\n", b, w2[b]); return 9; }
        static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                          {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
        for (int it2 = 0; it2 < 4; ++it2) {
            for (int gi = 0; gi < 4; ++gi) {                // ds_read_b128: the four 16-lane groups of the MI355X guide, bank = (address / 4) mod 64
                std::set<int> banks;
                for (int k = 0; k < 16; ++k) {
                    const uint32_t off = p_row_read_off(groups[gi][k], it2);
                    for (int b = 0; b < 4; ++b) banks.insert((off / 4 + b) % 64);
                }
                if (banks.size() != 64) { std::printf("ds_read_b128 bank conflict: it2 %d group %d touches %zu banks\n", it2, gi, banks.size()); return 10; }
            }
            for (int lane = 0; lane < 64; ++lane) {
                const uint32_t off = p_row_read_off(lane, it2);
                const int row = 4 * it2 + (lane >> 4), ch = lane & 15;
                for (int e = 0; e < 8; ++e)
                    if (row_img[off / 2 + e] != code(row, 8 * ch + e)) { std::printf("row image: lane %d it2 %d element %d wrong\n", lane, it2, e); return 11; }
            }
        }
    }
    // ---- C image (beta != 0): lanes write the chunks they loaded from C (row 4 it + (lane >> 4), chunk lane & 15) row-major, the transposing
    //      read of fragment j must hand lane (g, cl) rows 4 g + [0, 4) of column 16 j + cl — the accumulator fragment's layout
    {
        uint16_t c_img[16 * 128];
        std::memset(c_img, 0xff, sizeof(c_img));
        int w3[4096];
        std::memset(w3, 0, sizeof(w3));
        for (int it = 0; it < 4; ++it)
            for (int grp8 = 0; grp8 < 8; ++grp8) {          // ds_write_b128: 8 x 8 contiguous lanes, bank = (address / 4) mod 32
                std::set<int> banks;
                for (int lane = 8 * grp8; lane < 8 * grp8 + 8; ++lane) {
                    const uint32_t off = p_c_write_off(lane, it);
                    if (off % 16 != 0 || off + 16 > 4096u) { std::printf("C image write offset\n"); return 12; }
                    const int row = 4 * it + (lane >> 4), ch = lane & 15;
                    for (int e = 0; e < 8; ++e) c_img[off / 2 + e] = code(row, 8 * ch + e);
                    for (int b = 0; b < 16; ++b) ++w3[off + b];
                    for (int b = 0; b < 4; ++b) banks.insert((off / 4 + b) % 32);
                }
                if (banks.size() != 32) { std::printf("C image ds_write_b128 bank conflict: it %d lanes %d.. touch %zu banks\n", it, 8 * grp8, banks.size()); return 13; }
            }
        for (int b = 0; b < 4096; ++b)
            if (w3[b] != 1) { std::printf("C image byte %d written %d times\n", b, w3[b]); return 14; }
        for (int j = 0; j < 8; ++j) {
            for (int half = 0; half < 2; ++half) {           // ds_read_b64_tr_b16: 2 x 32 lanes, bank = (address / 4) mod 64
                std::set<int> banks;
                for (int lane = 32 * half; lane < 32 * half + 32; ++lane) {
                    const uint32_t off = p_c_read_off(lane, j);
                    if (off % 8 != 0 || off + 8 > 4096u) { std::printf("C image read offset\n"); return 15; }
                    banks.insert((off / 4) % 64);
                    banks.insert((off / 4 + 1) % 64);
                }
                if (banks.size() != 64) { std::printf("C image ds_read_b64_tr_b16 bank conflict: fragment %d half %d touches %zu banks\n", j, half, banks.size()); return 16; }
            }
            for (int grp = 0; grp < 4; ++grp)
                for (int i = 0; i < 16; ++i)
                    for (int kk = 0; kk < 4; ++kk) {
                        const int src = 16 * grp + 4 * kk + (i >> 2);
                        const uint16_t got = c_img[p_c_read_off(src, j) / 2 + (i & 3)];
                        if (got != code(4 * grp + kk, 16 * j + i)) {
                            std::printf("C image: fragment %d lane %d element %d: got (row %d, col %d), want (row %d, col %d)\n", j, 16 * grp + i, kk,
                                        got / 128, got % 128, 4 * grp + kk, 16 * j + i);
                            return 17;
                        }
                    }
        }
    }
    std::printf("h16p layout ok\n");
    return 0;
}
