// Host replay of the general MFMA GETT kernel's tile staging (cudalibrarysamples_amd/csrc/kernels/gett_gen_layout.h): for every
// (element size, tile rows, BK, orientation, vector width) the kernel table instantiates, all 256 threads stage their units into
// a byte image exactly as GenOperand::store does, and every lane of a wave then reads its MFMA fragments exactly as the kernel's
// compute step does.  Checked: every byte of the image is written exactly once; the element a lane reads for (k-block s, unit h,
// element e) of row rb + r is element (rb + r, GenFrag::k_of(s, q, h, e)) of the tile.  Test infrastructure (tests/test_gen_layout_cpu.py).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gett_gen_layout.h"

using namespace ctamd;

static int failures = 0;

template <int ES, int ORIENT, int ROWS, int BK, int V>
static void replay(const char* name) {
    using Map = GenUnitMap<ORIENT, ROWS, BK, V, 256>;
    using Img = GenImage<ES, BK>;
    using Frag = GenFrag<ES>;
    const int bytes = ROWS * Img::RB;
    std::vector<int> writes(bytes, 0);
    std::vector<unsigned char> image(bytes, 0);
    // element (row, k) byte b holds a hash of (row, k, b)
    auto val = [](int row, int k, int b) { return (unsigned char)((row * 131 + k * 17 + b * 7 + 3) & 0xff); };
    for (int tid = 0; tid < 256; ++tid) {
        const int kl = Map::unit_k(tid);
        for (int i = 0; i < Map::NU; ++i) {
            const int row = Map::unit_row(tid, i);
            for (int e = 0; e < V; ++e) {
                // the element this unit's e-th slot holds, and where GenOperand::store puts it
                const int er = ORIENT ? row : row + e, ek = ORIENT ? kl + e : kl;
                const int a = (ORIENT || V == 1) ? Img::addr(row, kl) + e * ES : Img::addr(row + e, kl);
                if (er >= ROWS || ek >= BK || a < 0 || a + ES > bytes) { std::printf("%s: unit out of the tile (tid %d unit %d)\n", name, tid, i); ++failures; return; }
                for (int b = 0; b < ES; ++b) { image[a + b] = val(er, ek, b); ++writes[a + b]; }
            }
        }
    }
    for (int a = 0; a < bytes; ++a)
        if (writes[a] != 1) { std::printf("%s: byte %d written %d times\n", name, a, writes[a]); ++failures; return; }
    const int KB = BK / Frag::KPB;
    if (KB < 1 || BK % Frag::KPB != 0) { std::printf("%s: BK is not whole k-blocks\n", name); ++failures; return; }
    std::vector<int> kSeen(BK, 0);
    for (int rb = 0; rb < ROWS; rb += 16)
        for (int lane = 0; lane < 64; ++lane) {
            const int r = lane & 15, q = lane >> 4;
            for (int s = 0; s < KB; ++s)
                for (int h = 0; h < Frag::UPL; ++h) {
                    // the kernel: fragOff[s][h] = unit_addr(0, r, unit(s, q, h)); address = tile + rb * RB + fragOff
                    const int a = rb * Img::RB + Img::unit_addr(0, r, Frag::unit(s, q, h));
                    if (a != Img::unit_addr(rb, r, Frag::unit(s, q, h))) { std::printf("%s: swizzle period broken at rb %d\n", name, rb); ++failures; return; }
                    for (int e = 0; e < Frag::EPU; ++e) {
                        const int k = Frag::k_of(s, q, h, e);
                        if (rb == 0 && r == 0) ++kSeen[k];
                        for (int b = 0; b < ES; ++b)
                            if (image[a + e * ES + b] != val(rb + r, k, b)) {
                                std::printf("%s: lane %d block %d unit %d elem %d of row %d is not (row, k = %d)\n", name, lane, s, h, e, rb + r, k);
                                ++failures;
                                return;
                            }
                    }
                }
        }
    for (int k = 0; k < BK; ++k)
        if (kSeen[k] != 1) { std::printf("%s: k = %d consumed %d times per row\n", name, k, kSeen[k]); ++failures; return; }
}

#define REPLAY(ES, ROWS, BK, V) replay<ES, 0, ROWS, BK, V>(#ES "B " #ROWS "x" #BK " V" #V " free-contiguous"); \
                                replay<ES, 1, ROWS, BK, V>(#ES "B " #ROWS "x" #BK " V" #V " K-contiguous");

int main() {
    // 16-bit (gett_gen_h16.hip)
    REPLAY(2, 128, 64, 8) REPLAY(2, 64, 64, 8) REPLAY(2, 64, 32, 2) REPLAY(2, 64, 32, 1)
    // fp64 / complex64 (8-byte elements)
    REPLAY(8, 128, 16, 2) REPLAY(8, 64, 16, 2) REPLAY(8, 64, 16, 1)
    // complex128
    REPLAY(16, 64, 8, 1)
    if (failures) { std::printf("%d layout failures\n", failures); return 1; }
    std::printf("gen layout ok\n");
    return 0;
}
