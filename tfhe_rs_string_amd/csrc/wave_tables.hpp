// wave_tables.hpp -- host-side construction of the lane-ordered twiddle tables used by the
// wave-per-ciphertext kernel (wave_pbs.hpp).  Pure table re-ordering of tw[] / W[]; no arithmetic.
#pragma once
#include <cstring>
#include <vector>

#include "pair_pbs.hpp"
#include "wave_pbs.hpp"

namespace tfhe_hip {
namespace wave {

// tw, w: FN interleaved (re, im) pairs (twiddle.cpp).  Returns the GTAB_BYTES image.
inline std::vector<char> build_tables(const double *tw, const double *w)
{
    std::vector<char> img(GTAB_BYTES, 0);
    auto put = [&](int off, const double *src, int idx) { std::memcpy(img.data() + off, src + 2 * idx, 16); };
    for (int r = 0; r < 32; r++)
        for (int l = 0; l < 32; l++) put((TAB_TW - TAB_TW) + (r * 32 + l) * 16, tw, l + 32 * r);
    for (int c = 0; c < 8; c++)
        for (int t = 0; t < 3; t++)
            for (int l = 0; l < 32; l++) put((TAB_T1 - TAB_TW) + ((c * 3 + t) * 32 + l) * 16, w, (l + 32 * c) * (t + 1));
    for (int par = 0; par < 2; par++)
        for (int j = 0; j < 32; j++) {
            const int jj = j & 15, q3 = par + 2 * (j >> 4);
            put((TAB_T3 - TAB_TW) + (par * 32 + j) * 16, w, 16 * jj * q3);
        }
    for (int c = 0; c < 2; c++)
        for (int t = 0; t < 3; t++)
            for (int l = 0; l < 32; l++) put(GTAB_T2 + ((c * 3 + t) * 32 + l) * 16, w, 4 * (l + 32 * c) * (t + 1));
    for (int m = 0; m < 16; m++) put(GTAB_WC + m * 16, w, 64 * m);
    return img;
}

} // namespace wave

namespace pair {

// image of the pair kernel's tables (GTAB_BYTES): twist [r 16][l 64], stage 1 [c 4][t 3][t1_slot(l) 64],
// stage-3 lane twiddles [slot 2][l 64]
inline std::vector<char> build_tables(const double *tw, const double *w)
{
    std::vector<char> img(GTAB_BYTES, 0);
    auto put = [&](int off, const double *src, int idx) { std::memcpy(img.data() + off, src + 2 * idx, 16); };
    for (int r = 0; r < 16; r++)
        for (int l = 0; l < 64; l++) put(TAB_TW + (r * 64 + l) * 16, tw, l + 64 * r);
    for (int c = 0; c < 4; c++)
        for (int t = 0; t < 3; t++)
            for (int l = 0; l < 64; l++) put(TAB_T1 + ((c * 3 + t) * 64 + t1_slot(l)) * 16, w, (l + 64 * c) * (t + 1));
    for (int sl = 0; sl < 2; sl++)
        for (int l = 0; l < 64; l++) {
            const int jj = l & 15, q3 = ((l >> 4) & 1) + 2 * sl;
            put(TAB_T3 + (sl * 64 + l) * 16, w, 16 * jj * q3);
        }
    return img;
}

} // namespace pair
} // namespace tfhe_hip
