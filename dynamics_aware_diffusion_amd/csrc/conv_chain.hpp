// conv_chain.hpp — the level-0 encoder chain of a narrow TemporalUnet (dim <= 128) in ONE launch.
//
// At batch 256 the five level-0 launches of a PointMaze step (downs.0.0: conv 6->128 + 1x1 residual ride,
// conv 128->128; downs.0.1: two convs 128->128; the down-sampling conv) run at 29..48 % of the fp32 matrix
// peak: each is 5.2 MFLOP per CU behind ~5.4 us of fixed cost (kernel boundary, first operand round trip,
// epilogue) and a four-chunk K loop that never reaches steady state.  A 128-channel x 32-position tile IS
// one sample, so a block can carry its sample through the whole chain
//   (temporal_unet.py:106-122 twice, then :35-43)
// with every intermediate activation resident in LDS; only the weights stream — straight from L2 into the
// MFMA B operand (a 32-position tile is ONE wave tile wide, so a weight fragment is used by exactly one wave
// and staging it through LDS would only add 2.5x the LDS write traffic per MFMA of the batch kernels), two
// 16-channel chunks ahead in registers, the next conv's first chunks in flight under the current epilogue.
// The K loops have no barrier at all.
// The level-0 block outputs are never read again (the reference pushes that skip and never pops it,
// temporal_unet.py:221,230), so the chain's only global store is the down-sampled tensor.
//
// Same arithmetic contract as conv_gemm.hpp: exact fp32 products (v_mfma_f32_32x32x2_f32), two-pass
// GroupNorm statistics in a fixed order, Mish on the transcendental units; only the order of the fp32
// additions differs.  Inference only (one shared timestep); fp32 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_cc.hpp"

namespace dad {

constexpr int CH_L = 32;            // positions per sample (the chain exists for horizon 32)
constexpr int CH_PAD = 2;
constexpr int CH_ROWS = CH_L + 2 * CH_PAD;
constexpr int CH_KC = 16;           // channels per weight chunk = one packed granule
constexpr int CH_KP = CH_KC + 4;    // LDS row stride of the weight stage (odd number of 16-byte slots)
constexpr int CH_MAX = 6;

struct ChainConv {
    const float* w;                 // chain image [chunk][pair = tap * 2 + g][C / 32][64 lanes][4]: lane (m, h) of wave
                                    // tile tm finds W[tm*32 + m][chunk*16 + g*8 + 4h .. +3][tap] (chain_repack_kernel)
    const float* bias; const float* gamma; const float* beta; const float* rbias;
    int32_t temb_off;               // offset into the time-table row, or -1
    int32_t cin;                    // input channels (the trajectory's for the first conv)
    int32_t taps, stride, wtaps;    // 5 / 1 / 5 (6 with the riding 1x1 conv);  3 / 2 / 3
    int32_t src;                    // 0: TX (trajectory tile), 1: T0, 2: T1
    int32_t dst;                    // 1: T0, 2: T1, 3: the global output (bias only, L / stride rows)
    int32_t add_t0;                 // add T0 after Mish (the block's residual; dst is T0 then)
    int32_t ride;                   // tap slot `taps` of the image is the block's 1x1 residual conv: its output
                                    // (+ rbias) goes to T0
    int32_t pad_;
};
constexpr int CH_N = 5;             // conv0 (+ride), conv1, conv0', conv1', down-sampling conv
struct ChainParams {
    const float* x;                 // (B, 32, td)
    float* out;                     // [B * 16][C]
    const float* temb_row;          // time-table row of this step
    int32_t B, td;
    // by value: every field is a kernel argument (scalar loads from the kernarg segment, compile-time conv
    // index after unrolling) — a descriptor table in global memory was read with VECTOR loads whose
    // s_waitcnt vmcnt(0) drained the weight prefetch at every chunk (781 vs 746 us per step)
    ChainConv c[CH_N];
};

template <int C> struct ChainShape {
    static constexpr int TM = C / 32;                       // wave tiles along the channels
    static constexpr int SK = C == 128 ? 2 : 4;             // waves sharing a tile (split over the K units)
    static constexpr int NW = TM * SK, NT = 64 * NW;
    static constexpr int RS = C + 4;                        // activation row stride (odd number of 16-byte slots)
    static constexpr int T_FLOATS = CH_ROWS * RS;
    static constexpr int TX_FLOATS = CH_ROWS * CH_KP;
    static constexpr int E_FLOATS = SK * CH_L * RS;         // exchange tile [SK][32 rows][RS]
    static constexpr int PRM_FLOATS = (4 * (CH_N - 1) + 2) * C;   // bias, gamma, beta, time row per GroupNorm'd conv; ride bias; down bias
    static constexpr int LDS_FLOATS = 2 * T_FLOATS + TX_FLOATS + E_FLOATS + PRM_FLOATS;
    static constexpr int NP = (12 + SK - 1) / SK;           // (tap, 8-channel half) pairs of a chunk one wave owns: pu = ks + SK j
};

// chain image from the standard packed image [chunk][wtaps][C][16] (csrc/host_plan.hpp pack_op); one thread per float4
__global__ void chain_repack_kernel(float* dst, const float* src, int nch, int wtaps, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // float4 index of dst
    const int TM = C / 32;
    const long n = (long)nch * 2 * wtaps * TM * 64;
    if (i >= n) return;
    const int lane = (int)(i & 63);
    long r = i >> 6;
    const int tm = (int)(r % TM); r /= TM;
    const int pu = (int)(r % (2 * wtaps));
    const int ch = (int)(r / (2 * wtaps));
    const int tap = pu >> 1, g = pu & 1, m = tm * 32 + (lane & 31), h = lane >> 5;
    reinterpret_cast<float4*>(dst)[i] =
        *reinterpret_cast<const float4*>(src + (((long)ch * wtaps + tap) * C + m) * 16 + g * 8 + 4 * h);
}

#ifdef DAD_CHAIN_STAMPS
__device__ unsigned long long g_chain_stamps[32];
#define CHAIN_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_chain_stamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CHAIN_STAMP(k) do {} while (0)
#endif

template <int C>
__global__ __launch_bounds__(ChainShape<C>::NT) void chain_l0_kernel(const ChainParams p) {
    using S = ChainShape<C>;
    constexpr int TM = S::TM, SK = S::SK, NW = S::NW, NT = S::NT, RS = S::RS, NP = S::NP;
    constexpr int CPG = C / 8, CQ = CPG / 4;                // channels / float4 per GroupNorm group row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const T0 = smem;
    float* const T1 = T0 + S::T_FLOATS;
    float* const TX = T1 + S::T_FLOATS;
    float* const E = TX + S::TX_FLOATS;                     // [SK][32 rows][RS]
    float* const PRM = E + S::E_FLOATS;                     // [conv][bias | gamma | beta | time row][C], then ride bias
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tm = wave % TM, ks = wave / TM;
    const int l32 = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    CHAIN_STAMP(0);

    // ---- weight stream: items = (conv, 16-channel chunk) in order, TWO items ahead in registers ----------
    // Register set s & 1 holds item s: the NP (tap, half) pairs this wave owns, one float4 per lane each,
    // contiguous 1 KiB per wave in the chain image.  A slot is refilled with item s + 2 right after its MFMAs.
    // The first conv has one chunk (cin <= 16), every later one an even number (cin = C), so a conv starts on
    // parity 1 and ends on parity 0: the two-item body below is static, no register array is indexed at run time.
    float4 wA[NP], wB[NP];
    // pair j of chunk `ch` of conv `cv` (ch may run past its last chunk: then chunk ch - nch of the next conv `nx`)
    auto wload = [&](const ChainConv& cv, const ChainConv& nx, int ch, int j) -> float4 {
        const int nch = (cv.cin + CH_KC - 1) >> 4;
        const bool here = ch < nch;
        const float* w = here ? cv.w : nx.w;
        const int wt2 = 2 * (here ? cv.wtaps : nx.wtaps);
        const int k = here ? ch : ch - nch;
        const int pu = min(ks + SK * j, wt2 - 1);           // (pairs past the conv's taps re-read the last one: unconditional load)
        return ldg4(w + ((((long)k * wt2 + pu) * TM + tm) * 64 + lane) * 4);
    };
#pragma unroll
    for (int j = 0; j < NP; ++j) { wA[j] = wload(p.c[0], p.c[1], 0, j); wB[j] = wload(p.c[0], p.c[1], 1, j); }

    // ---- per-channel parameters of all five convs into LDS, once: an epilogue that fetched them from memory
    // had to wait for them with vmcnt(0), which also drains the two weight items in flight
    for (int e = tid; e < C / 4; e += NT) {
#pragma unroll
        for (int i = 0; i < CH_N; ++i) {
            const ChainConv& c = p.c[i];
            float* dst = PRM + i * 4 * C + 4 * e;
            *reinterpret_cast<float4*>(dst) = ldg4(c.bias + 4 * e);
            if (c.gamma != nullptr) {
                *reinterpret_cast<float4*>(dst + C) = ldg4(c.gamma + 4 * e);
                *reinterpret_cast<float4*>(dst + 2 * C) = ldg4(c.beta + 4 * e);
                *reinterpret_cast<float4*>(dst + 3 * C) = c.temb_off >= 0 ? ldg4(p.temb_row + c.temb_off + 4 * e) : zero4;
            }
        }
        *reinterpret_cast<float4*>(PRM + (4 * (CH_N - 1) + 1) * C + 4 * e) = ldg4(p.c[0].rbias + 4 * e);
    }

    // ---- the trajectory tile (channels padded to 16) and the zero halos --------------------------------
    for (int i = tid; i < CH_ROWS * CH_KP; i += NT) {
        const int row = i / CH_KP, cidx = i - row * CH_KP;
        const int l = row - CH_PAD;
        float v = 0.0f;
        if (l >= 0 && l < CH_L && cidx < p.td) v = p.x[((long)b * CH_L + l) * p.td + cidx];
        TX[i] = v;
    }
    for (int i = tid; i < 2 * CH_PAD * (RS / 4); i += NT) {
        const int hr = i / (RS / 4), q = i - hr * (RS / 4);
        const int row = hr < CH_PAD ? hr : CH_L + hr;
        *reinterpret_cast<float4*>(T0 + row * RS + 4 * q) = zero4;
        *reinterpret_cast<float4*>(T1 + row * RS + 4 * q) = zero4;
    }

    CHAIN_STAMP(1);
    __syncthreads();                                        // tiles, halos and parameters are in place
    CHAIN_STAMP(2);
#pragma unroll
    for (int i = 0; i < CH_N; ++i) {
        const ChainConv& c = p.c[i];
        const ChainConv& nx = p.c[i + 1 < CH_N ? i + 1 : i];     // (past the last conv: re-reads it, unused)
        const int nch = (c.cin + CH_KC - 1) >> 4;
        const int wtaps = c.wtaps, taps = c.taps;
        const float* const A = c.src == 0 ? TX : (c.src == 1 ? T0 : T1);
        const int rsA = c.src == 0 ? CH_KP : RS;
        // lane's GEMM row -> tile row of tap 0 (the strided conv has 16 output positions: rows beyond re-read row 15)
        const int nn = c.stride == 2 ? min(l32, CH_L / 2 - 1) : l32;
        const int arow0 = nn * c.stride + (CH_PAD - (taps >> 1));
        f32x16 acc, acc2, accr;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; accr[r] = 0.f; }
        // one item: the wave's pairs of chunk `ch` — A fragment from the resident tile, B fragment from the
        // register set, four MFMAs, and the slot takes the same pair of item s + 2.  No barrier.
        auto item = [&](float4 (&wr)[NP], int ch) {
            // all A fragments of the item first: their LDS latency overlaps the MFMAs of the earlier pairs
            float4 af[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int pu = min(ks + SK * j, 2 * wtaps - 1);
                const int tap = pu >> 1, g = pu & 1;
                const int tap_eff = tap == taps ? (taps >> 1) : tap;
                af[j] = *reinterpret_cast<const float4*>(A + (arow0 + tap_eff) * rsA + ch * CH_KC + g * 8 + 4 * h);
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int pu = ks + SK * j;
                if (pu < 2 * wtaps) {
                    const int tap = pu >> 1;
                    const bool is_ride = tap == taps;       // (only when c.ride: wtaps == taps + 1)
                    const float4 a = af[j];
                    const float4 w = wr[j];
                    if (is_ride) {
                        accr = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, accr, 0, 0, 0);
                        accr = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, accr, 0, 0, 0);
                        accr = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, accr, 0, 0, 0);
                        accr = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, accr, 0, 0, 0);
                    } else {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc2, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc2, 0, 0, 0);
                    }
                }
                wr[j] = wload(c, nx, ch + 2, j);            // lands under the next two chunks
            }
        };
        if (i == 0) {
            item(wA, 0);                                    // the first conv: one chunk, parity 0
        } else {
            for (int ch = 0; ch < nch; ch += 2) {           // parity 1, then parity 0
                item(wB, ch);
                item(wA, ch + 1);
            }
        }
        CHAIN_STAMP(3 + 2 * i);
        // ---- epilogue: the SK partial tiles meet in LDS
        if (c.ride) {                                       // the 1x1 residual conv: bias only, into T0
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                E[(ks * CH_L + row) * RS + tm * 32 + l32] = accr[r];
            }
            __syncthreads();
            for (int e = tid; e < CH_L * (C / 4); e += NT) {
                const int row = e / (C / 4), q = e - row * (C / 4);
                float4 v = *reinterpret_cast<const float4*>(E + row * RS + 4 * q);
#pragma unroll
                for (int k = 1; k < SK; ++k) {
                    const float4 u = *reinterpret_cast<const float4*>(E + (k * CH_L + row) * RS + 4 * q);
                    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                }
                const float4 rb = *reinterpret_cast<const float4*>(PRM + (4 * (CH_N - 1) + 1) * C + 4 * q);
                *reinterpret_cast<float4*>(T0 + (row + CH_PAD) * RS + 4 * q) = make_float4(v.x + rb.x, v.y + rb.y, v.z + rb.z, v.w + rb.w);
            }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            E[(ks * CH_L + row) * RS + tm * 32 + l32] = acc[r] + acc2[r];
        }
        __syncthreads();
        if (c.dst == 3) {                                   // down-sampling conv: bias only, 16 rows, to memory
            for (int e = tid; e < (CH_L / 2) * (C / 4); e += NT) {
                const int row = e / (C / 4), q = e - row * (C / 4);
                float4 v = *reinterpret_cast<const float4*>(E + row * RS + 4 * q);
#pragma unroll
                for (int k = 1; k < SK; ++k) {
                    const float4 u = *reinterpret_cast<const float4*>(E + (k * CH_L + row) * RS + 4 * q);
                    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                }
                const float4 bb = *reinterpret_cast<const float4*>(PRM + i * 4 * C + 4 * q);
                store_f4_sc1(p.out + ((long)b * (CH_L / 2) + row) * C + 4 * q, make_float4(v.x + bb.x, v.y + bb.y, v.z + bb.z, v.w + bb.w));
            }
        } else {
            // GroupNorm(8) + Mish (+ time embedding, + the block's residual): one wave per group at a time,
            // the pair's 32 x CPG values in registers (lane j: row j / CQ, channel quad j % CQ)
            float* const D = c.dst == 1 ? T0 : T1;
            constexpr int F4 = (8 * CPG + 63) / 64;         // float4 per lane (1 or 2)
            const float inv_cnt = 1.0f / (float)(CH_L * CPG);
            for (int g = wave; g < 8; g += NW) {
                float4 v[F4];
                int at[F4];
                float sum = 0.0f;
#pragma unroll
                for (int k = 0; k < F4; ++k) {
                    const int j = lane + 64 * k;
                    const bool on = j < 8 * CPG;
                    const int row = (on ? j : 0) / CQ, cq = (on ? j : 0) - row * CQ;
                    const int col = g * CPG + 4 * cq;
                    at[k] = on ? (row + CH_PAD) * RS + col : -1;
                    float4 t = *reinterpret_cast<const float4*>(E + row * RS + col);
#pragma unroll
                    for (int s = 1; s < SK; ++s) {
                        const float4 u = *reinterpret_cast<const float4*>(E + (s * CH_L + row) * RS + col);
                        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
                    }
                    const float4 bb = *reinterpret_cast<const float4*>(PRM + i * 4 * C + col);
                    t.x += bb.x; t.y += bb.y; t.z += bb.z; t.w += bb.w;
                    v[k] = on ? t : zero4;
                    sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
                }
                const float mean = wave_sum(sum) * inv_cnt;
                float sq = 0.0f;
#pragma unroll
                for (int k = 0; k < F4; ++k)
                    if (at[k] >= 0) {
                        const float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
                        sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
                    }
                const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_cnt + 1e-5f);
#pragma unroll
                for (int k = 0; k < F4; ++k) {
                    if (at[k] < 0) continue;
                    const int col = at[k] % RS;
                    const float4 gam = *reinterpret_cast<const float4*>(PRM + (i * 4 + 1) * C + col);
                    const float4 bet = *reinterpret_cast<const float4*>(PRM + (i * 4 + 2) * C + col);
                    float4 y;
                    y.x = mish_fast_f32((v[k].x - mean) * rstd * gam.x + bet.x);
                    y.y = mish_fast_f32((v[k].y - mean) * rstd * gam.y + bet.y);
                    y.z = mish_fast_f32((v[k].z - mean) * rstd * gam.z + bet.z);
                    y.w = mish_fast_f32((v[k].w - mean) * rstd * gam.w + bet.w);
                    {
                        const float4 tv = *reinterpret_cast<const float4*>(PRM + (i * 4 + 3) * C + col);   // (zeros: no time embedding)
                        y.x += tv.x; y.y += tv.y; y.z += tv.z; y.w += tv.w;
                    }
                    if (c.add_t0) {
                        const float4 rv = *reinterpret_cast<const float4*>(T0 + at[k]);
                        y.x += rv.x; y.y += rv.y; y.z += rv.z; y.w += rv.w;
                    }
                    *reinterpret_cast<float4*>(D + at[k]) = y;
                }
            }
        }
        __syncthreads();
        CHAIN_STAMP(4 + 2 * i);
    }
}

}  // namespace dad
