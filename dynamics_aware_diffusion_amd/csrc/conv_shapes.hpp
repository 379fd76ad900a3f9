// conv_shapes.hpp — sizing formulas shared by the conv-GEMM kernels (device) and the launch
// planner (host).  No HIP dependency: the planner is also compiled host-only under
// -fsanitize=address,undefined (tests/sanitize/).
#pragma once
#include <stddef.h>

#if defined(__HIPCC__)
#define DAD_HD __host__ __device__
#else
#define DAD_HD
#endif

namespace dad {

constexpr int kXSwzPad = 64;    // floats: room for the per-sample slot shifts of the X stage
constexpr size_t kLdsBytes = 160 * 1024;   // LDS of one gfx950 CU

// Rows of the X stage: every sample of the tile with its zero halo.
DAD_HD inline int conv_xrows(int BN, int Lin, int Lout, int taps) {
    return (BN / Lout) * (Lin + 2 * (taps / 2));
}
// LDS floats of one block (the host sizes the dynamic allocation with the same formula).
// wtaps: weight rows staged per chunk, in taps (taps + 1 when a 1x1 residual conv rides along).
DAD_HD inline size_t conv_lds_floats(int BM, int BN, int KC, int taps, int Lin, int Lout, int SK,
                                     bool bdir = false, int wtaps = 0) {
    const size_t kp = KC + 4;
    const size_t wt = wtaps ? wtaps : taps;
    const size_t stage = (size_t)conv_xrows(BN, Lin, Lout, taps) * kp + kXSwzPad +
                         (bdir ? 0 : wt * BM * kp);
    const size_t epi = (size_t)SK * BN * (BM + 4) + 64;
    const size_t k = 2 * stage;
    return k > epi ? k : epi;
}

// conv_cc.hpp: rows of a block's X stage — whole samples with their zero halo, or (layers of more than `nr`
// positions: windowed tiles) `nr` rows of one sample plus the halo on both sides.
DAD_HD inline int cc_xrows(int taps, int Lin, int Lout, int nr) {
    return Lout > nr ? nr + 2 * (taps / 2) : (nr / Lout) * (Lin + 2 * (taps / 2));
}

// conv_ccw.hpp (wide small-batch convs).  K phase: X rows with halo [XROWS][slice + 4], the additive
// terms [rows][slice + 4], gamma / beta [2][slice], pair statistics; afterwards the exchange tile.
constexpr int kCcwMaxPairs = 64;         // (sample, group) pairs of one block's input slice
DAD_HD inline size_t ccw_lds_floats(int slice_ch, int taps, int Lin, int Lout, int nr) {
    const int spt = nr / Lout;
    const size_t xs = slice_ch + 4;
    const size_t k = (size_t)spt * (Lin + 2 * (taps / 2)) * xs + (size_t)spt * Lin * xs + 2 * (size_t)slice_ch +
                     2 * kCcwMaxPairs;
    const size_t e = (size_t)2 * 8 * nr * 36;
    return k > e ? k : e;
}

}  // namespace dad
