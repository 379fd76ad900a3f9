// train_bwd.hpp — the kernels of the backward pass that are not a convolution themselves.
//
// The reference trains with loss.backward() through TemporalUnet (m_diffuser/utils/training.py:144-156,
// m_diffuser/models/diffusion.py:253-290): autograd walks Conv1d / ConvTranspose1d / GroupNorm / Mish /
// Linear.  Here the data gradients of the convs run on the forward conv-GEMM kernel with transposed,
// tap-flipped weight images (dad_lib.hip, "backward plan"); this file holds the rest:
//
//   conv_wgrad<TAPS,TM,TN>   dW[m][c][k] = sum_{b,l} dH[b,l,m] * X[b, l*stride + k - pad, c]     (MFMA GEMM, K = B*L)
//   gn_mish_bwd        dH = d(conv + bias) of  Mish(GroupNorm(h)) (+ time embedding), per (sample, group),
//                      with the per-sample partial sums of d gamma, d beta, d bias and d(time projection)
//   row_partial_sums / col_sums_many / sum_slabs / add_inplace deterministic reductions (fixed order, no atomics)
//
// Everything is fp32 and bit-reproducible run to run.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_gemm.hpp"

namespace dad {

// ------------------------------------------------------------------------------------ wgrad
// GEMM view: D[m][c] (one per tap) = sum over rows n = (b, l) of  G[n][m] * Z[b][l*stride + tap - pad][c].
//   forward conv k (stride 1, pad k/2):   G = dH (Lg = L),      Z = the conv's input (Lz = L)
//   Downsample1d (k=3, s=2, p=1):          G = dH (Lg = L/2),    Z = input (Lz = L)
//   Upsample1d (ConvTranspose k=4,s=2,p=1): G = the INPUT (Lg = L), Z = dY (Lz = 2L), taps 4, stride 2, pad 1
//     -> D[ci][co][kk] = sum x[b,i,ci] dY[b, 2i-1+kk, co], the (in, out, k) layout of its weight
// Z may be a virtual channel concat [Z0 | Z1] (decoder blocks).  Output index (m * C + c) * TAPS + tap.
struct WgradParams {
    const float* G;  int32_t ldg, M;         // [B*Lg][ldg], columns [0, M) used
    const float* Z0; int32_t ldz0, C0;       // [B*Lz][ldz0], columns [0, C0)
    const float* Z1; int32_t ldz1, C1;       // second half of the concat, or nullptr / 0
    float* out;                              // ksplit == 1: the gradient tensor; else [ksplit][out_numel] partials
    long out_numel;
    int32_t B, Lg, Lz, lg_shift;             // Lg = 1 << lg_shift
    int32_t stride, pad;
    int32_t ksplit, samples_per_split, spc;  // batch split over blockIdx.z; samples per staged chunk
    const float* zero;                       // >= 16 bytes of zeros (what the LDS-DMA fetches for halo rows)
};

constexpr int WG_THREADS = 512;              // 8 waves: TM x TN wave tiles of 32 x 32 (x TAPS), the rest split K
constexpr int WG_ROWS = 64;                  // G rows per staged chunk: spc = max(1, 64 / Lg) whole samples
constexpr int WG_MAX_GROWS = 128, WG_MAX_ZROWS = 160;      // rows one chunk may stage (registers of chunk_load)

__host__ __device__ inline int wgrad_segz(int Lz, int taps, int pad) { return Lz + pad + (taps - 1 - pad); }
__host__ __device__ inline int wgrad_round64(int v) { return (v + 63) / 64 * 64; }
// LDS floats: two stages of a chunk (G rows [spc * Lg][32 TM], Z rows with halo [spc * SEGZ][32 TN], each part
// rounded up to whole 64-float4 wave-instructions of the LDS-DMA), and afterwards the K-group reduction tree, whose
// first round parks half of the block's accumulators: 4 waves x TAPS x 16 x 64.
__host__ __device__ inline size_t wgrad_lds_floats(int spc, int Lg, int Lz, int taps, int pad, int tm, int tn) {
    const size_t stage = 2 * ((size_t)wgrad_round64(spc * Lg * 8 * tm) * 4 + (size_t)wgrad_round64(spc * wgrad_segz(Lz, taps, pad) * 8 * tn) * 4);
    const size_t red = (size_t)4 * taps * 16 * 64;
    return stage > red ? stage : red;
}

// four consecutive columns of a row, zero beyond `ncols`; vector load when the row is 16-byte aligned
__device__ __forceinline__ float4 wg_load4(const float* row, int col, int ncols, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col + 3 < ncols && vec) return ldg4(row + col);
    if (col < ncols) v.x = row[col];
    if (col + 1 < ncols) v.y = row[col + 1];
    if (col + 2 < ncols) v.z = row[col + 2];
    if (col + 3 < ncols) v.w = row[col + 3];
    return v;
}

#ifdef DAD_WG_STAMPS
__device__ unsigned long long g_wg_stamps[32];
#define WG_STAMP(k) do { if (stamp_on && threadIdx.x == 0) g_wg_stamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WG_STAMP(k) do { } while (0)
#endif

// Block = 8 waves = (TM x TN) wave tiles x KG K-groups, KG = 8 / (TM TN): a 64 x 64 block tile with two K-groups
// for layers with many tiles, 64 x 32 with four and 32 x 32 with eight for layers with few (the batch then splits
// over fewer blocks: less slab traffic, and the chip still holds two waves per SIMD).  All waves stage a chunk
// together; K-group g runs the chunk's steps [g, g + 1) * steps / KG with its fragments read one step ahead; the
// groups' accumulators meet in a fixed-shape tree through LDS (bit-reproducible), group 0 stores.
//
// Staging.  Chunk i + 1 is on its way into the other LDS stage while chunk i is computed; one barrier per chunk.
// ALIGNED (every operand row is whole 16-byte-aligned float4s: all layers but those that touch the trajectory's td
// columns): LDS-DMA, `global_load_lds_dwordx4` — a wave-instruction moves 64 float4 to 1 KiB of LDS, no registers,
// nothing to wait for until the chunk's closing barrier.  Every chunk has the same shape, so what a thread fetches
// is decided ONCE: per float4 an element offset from the chunk's first row (-1: halo / beyond the tile -> a zero
// word in global memory) and its sample.  (Register staging held the loads of the next chunk in flight across the
// loop's back edge: hipcc put copies of the destination registers behind the loads, each with its `s_waitcnt` — a
// memory round trip per chunk, 12 us per chunk against 5 us of MFMAs by the in-kernel stamps.)
// !ALIGNED: registers, loaded before the chunk's MFMAs and written to the other stage behind them.
template <int TAPS, int TM, int TN, bool ALIGNED>
__global__ __launch_bounds__(WG_THREADS) void conv_wgrad(const WgradParams p) {
    constexpr int NT = TM * TN, KG = 8 / NT;
    constexpr int WMW = 32 * TM, WNW = 32 * TN;              // staged columns of G / Z
    constexpr int GQ = WMW / 4, ZQ = WNW / 4;                // float4 per staged row
    constexpr int WG_GI = (WG_MAX_GROWS * GQ + WG_THREADS - 1) / WG_THREADS;
    constexpr int WG_ZI = (WG_MAX_ZROWS * ZQ + WG_THREADS - 1) / WG_THREADS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave % NT, kg = wave / NT;
    const int wm = wt % TM, wn = wt / TM;
    const int l32 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * WMW, c0 = blockIdx.y * WNW;
    const int ks = blockIdx.z;
    const int Lg = p.Lg, Lz = p.Lz;
    const int SEGZ = wgrad_segz(Lz, TAPS, p.pad);
    const int spc = p.spc;
    // a stage = [G rows: spc * Lg][WMW] | [Z rows with halo: spc * SEGZ][WNW], each part rounded up to whole
    // wave-instructions of the LDS-DMA (64 float4)
    const int n_g = spc * Lg * GQ, n_z = spc * SEGZ * ZQ;    // float4s
    const int zs_at = wgrad_round64(n_g) * 4;
    const int stage_floats = zs_at + wgrad_round64(n_z) * 4;
    const int s_lo = ks * p.samples_per_split;
    const int s_hi = min(p.B, s_lo + p.samples_per_split);
    const int Ctot = p.C0 + p.C1;
#ifdef DAD_WG_STAMPS
    const bool stamp_on = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && p.M == 512 && Ctot == 512;
    int chunk_no = 0;
#endif
    WG_STAMP(0);

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // (no lambdas or macros over these arrays: a capture by reference puts them on a scratch segment)
    int g_off[WG_GI], z_off[WG_ZI];              // ALIGNED: element offsets from the chunk's first row, -1 = zero
    int g_smp[WG_GI], z_smp[WG_ZI];              // ALIGNED: sample of the chunk; Z: + 4096 when the source is Z1
    float4 gr[WG_GI], zr[WG_ZI];                 // !ALIGNED: the next chunk on its way
    if constexpr (ALIGNED) {
#pragma unroll
        for (int k = 0; k < WG_GI; ++k) {
            const int i = tid + k * WG_THREADS;
            const int row = i / GQ, q = i % GQ, col = m0 + 4 * q;
            g_smp[k] = row >> p.lg_shift;
            g_off[k] = (i < n_g && col < p.M) ? (row * p.ldg + col) : -1;       // (row = sample * Lg + l)
        }
#pragma unroll
        for (int k = 0; k < WG_ZI; ++k) {
            const int i = tid + k * WG_THREADS;
            const int row = i / ZQ, q = i % ZQ, c = c0 + 4 * q;
            const int sl = row / SEGZ, pz = row - sl * SEGZ - p.pad;
            const bool ok = i < n_z && pz >= 0 && pz < Lz && c < Ctot;
            const bool second = c >= p.C0;
            z_smp[k] = sl + (second ? 4096 : 0);
            z_off[k] = !ok ? -1 : second ? (sl * Lz + pz) * p.ldz1 + (c - p.C0) : (sl * Lz + pz) * p.ldz0 + c;
        }
    }
    const bool gvec = (p.ldg & 3) == 0, z0vec = (p.ldz0 & 3) == 0, z1vec = (p.ldz1 & 3) == 0;

    // fragments of step s (rows 2s, 2s + 1; lane half h takes row 2s + h): one G value, TAPS Z values
    auto frag = [&](const float* stage, int s, float& a, float (&z)[TAPS]) {
        const int row = 2 * s + h;
        const int sl = row >> p.lg_shift, l = row & (Lg - 1);
        a = stage[row * WMW + wm * 32 + l32];
        const float* zb = stage + zs_at + (sl * SEGZ + l * p.stride) * WNW + wn * 32 + l32;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) z[t] = zb[t * WNW];
    };
    const int spg = (spc * Lg) / (2 * KG);                   // steps of a chunk per K-group (host: divides, even)
    const int s_first = kg * spg, s_last = s_first + spg - 1;

    // iteration -1 only fetches chunk 0; iteration i fetches chunk i + 1 and computes chunk i
    int cur = 1;
    for (int sb = s_lo - spc; sb < s_hi; sb += spc, cur ^= 1) {
        const int nb = sb + spc;                             // first sample of the chunk to fetch
        float* const next = smem + (cur ^ 1) * stage_floats;
        if (nb < s_hi) {
            if constexpr (ALIGNED) {
                const float* const gb = p.G + (long)nb * Lg * p.ldg;
                const float* const zb0 = p.Z0 + (long)nb * Lz * p.ldz0;
                const float* const zb1 = p.Z1 != nullptr ? p.Z1 + (long)nb * Lz * p.ldz1 : zb0;
#pragma unroll
                for (int k = 0; k < WG_GI; ++k) {
                    const int at = wave * 64 + k * WG_THREADS;                  // the wave's first float4: uniform
                    if (at < n_g) {
                        const bool ok = g_off[k] >= 0 && nb + g_smp[k] < s_hi;
                        const float* src = ok ? gb + g_off[k] : p.zero;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)(next + 4 * at), 16, 0, 0);
                    }
                }
#pragma unroll
                for (int k = 0; k < WG_ZI; ++k) {
                    const int at = wave * 64 + k * WG_THREADS;
                    if (at < n_z) {
                        const bool ok = z_off[k] >= 0 && nb + (z_smp[k] & 4095) < s_hi;
                        const float* src = ok ? (z_smp[k] >= 4096 ? zb1 : zb0) + z_off[k] : p.zero;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)(next + zs_at + 4 * at), 16, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < WG_GI; ++k) {
                    const int i = tid + k * WG_THREADS;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (i < n_g) {
                        const int row = i / GQ, q = i % GQ;
                        const int smp = nb + (row >> p.lg_shift), l = row & (Lg - 1);
                        if (smp < s_hi) v = wg_load4(p.G + (long)(smp * Lg + l) * p.ldg, m0 + 4 * q, p.M, gvec);
                    }
                    gr[k] = v;
                }
#pragma unroll
                for (int k = 0; k < WG_ZI; ++k) {
                    const int i = tid + k * WG_THREADS;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (i < n_z) {
                        const int row = i / ZQ, q = i % ZQ;
                        const int sl = row / SEGZ, pz = row - sl * SEGZ - p.pad;
                        const int smp = nb + sl, c = c0 + 4 * q;
                        if (smp < s_hi && pz >= 0 && pz < Lz && c < Ctot) {
                            if (c < p.C0) v = wg_load4(p.Z0 + (long)(smp * Lz + pz) * p.ldz0, c, p.C0, z0vec);
                            else v = wg_load4(p.Z1 + (long)(smp * Lz + pz) * p.ldz1, c - p.C0, p.C1, z1vec);
                        }
                    }
                    zr[k] = v;
                }
            }
        }
        if (sb >= s_lo) {
            const float* const stage = smem + cur * stage_floats;
            // two steps per round on two register sets, each set's LDS reads issued a whole step ahead of its
            // MFMAs (hipcc otherwise sinks the reads to their use: one LDS round trip exposed per step)
            float a0, z0[TAPS], a1, z1[TAPS];
            frag(stage, s_first, a0, z0);
            for (int s = s_first; s < s_last; s += 2) {
                frag(stage, s + 1, a1, z1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, z0[t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                frag(stage, min(s + 2, s_last), a0, z0);     // (the last round re-reads a step it does not use)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, z1[t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (!ALIGNED) {
            if (nb < s_hi) {
#pragma unroll
                for (int k = 0; k < WG_GI; ++k) {
                    const int i = tid + k * WG_THREADS;
                    if (i < n_g) *reinterpret_cast<float4*>(next + 4 * i) = gr[k];
                }
#pragma unroll
                for (int k = 0; k < WG_ZI; ++k) {
                    const int i = tid + k * WG_THREADS;
                    if (i < n_z) *reinterpret_cast<float4*>(next + zs_at + 4 * i) = zr[k];
                }
            }
        }
        __syncthreads();                 // the fetched chunk has landed (the barrier's fence drains the LDS-DMA); stage cur is free
#ifdef DAD_WG_STAMPS
        if (chunk_no < 24) WG_STAMP(1 + chunk_no);
        ++chunk_no;
#endif
    }
    // ---- the K-groups meet: groups [half, 2 half) park their accumulators, groups [0, half) add them
    if constexpr (KG > 1) {
#pragma unroll
        for (int half = KG / 2; half >= 1; half >>= 1) {     // (the loop's last barrier freed the staging LDS)
            if (kg >= half && kg < 2 * half) {
                float* dst = smem + ((kg - half) * NT + wt) * (TAPS * 16 * 64) + lane;
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[(t * 16 + r) * 64] = acc[t][r];
            }
            __syncthreads();
            if (kg < half) {
                const float* src = smem + (kg * NT + wt) * (TAPS * 16 * 64) + lane;
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += src[(t * 16 + r) * 64];
            }
            if (half > 1) __syncthreads();
        }
        WG_STAMP(28);
        if (kg != 0) return;
    }
    // ---- store: D row = m (first operand), column = c
    float* const out = p.out + (p.ksplit > 1 ? (long)ks * p.out_numel : 0L);
    const int c = c0 + wn * 32 + l32;
    if (c < Ctot) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= p.M) continue;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) out[((long)m * Ctot + c) * TAPS + t] = acc[t][r];
        }
    }
    WG_STAMP(29);
}

// ------------------------------------------------------------------ device-side weight re-packing
// dad_model_refresh_weights: a training loop changes the parameters every step; the packed images
// (csrc/host_plan.hpp pack_op / pack_bwd_op) are rebuilt ON THE DEVICE from the parameter tensors in the
// reference's layouts, one thread per element of the image (padding included, so no memset):
//   image index = (((ci / kg) * wtaps + slot) * M + o) * kg + ci % kg
enum RepackMode {
    RP_FWD = 0,        // Conv1d (co, ci, K) [+ riding 1x1 conv as slot K]
    RP_FWD_UP = 1,     // ConvTranspose1d (ci, co, 4) as two 2-tap phases, M = 2 co
    RP_BWD_CONV = 2,   // data gradient of Conv1d: (m, c, K-1-slot) <- W[c][c_lo + m][.]
    RP_BWD_DOWN = 3,   // data gradient of Downsample1d as a transposed conv whose 4th tap is zero
    RP_BWD_UP = 4,     // data gradient of Upsample1d as a 5-tap stride-2 conv whose first tap is zero
    RP_BWD_FINAL = 5,  // data gradient of final_conv[1]
};
struct RepackParams {
    float* dst; const float* w; const float* ride;     // ride: the 1x1 residual conv's weight, or nullptr
    long n;                                            // elements of the image
    int32_t mode, kg, wtaps, M;
    int32_t CO, CI, K;                                 // the SOURCE tensor's dims as the mode reads them
    int32_t c_lo, c_n;                                 // RP_BWD_CONV: input-channel range of the forward conv
};
__global__ void repack_kernel(const RepackParams p) {
    const long d = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= p.n) return;
    const int j = (int)(d % p.kg);
    long r = d / p.kg;
    const int o = (int)(r % p.M);
    r /= p.M;
    const int slot = (int)(r % p.wtaps);
    const int ci = (int)(r / p.wtaps) * p.kg + j;
    float v = 0.0f;
    switch (p.mode) {
        case RP_FWD:
            if (ci < p.CI) {
                if (slot < p.K) v = p.w[((long)o * p.CI + ci) * p.K + slot];
                else if (p.ride != nullptr) v = p.ride[(long)o * p.CI + ci];
            }
            break;
        case RP_FWD_UP: case RP_BWD_DOWN: {
            const int co = p.M >> 1, half = o >= co, oo = o - half * co;
            const int kk = half == 0 ? (slot == 0 ? 3 : 1) : (slot == 0 ? 2 : 0);
            if (p.mode == RP_FWD_UP) { if (ci < p.CI) v = p.w[((long)ci * co + oo) * 4 + kk]; }
            else if (ci < p.CO && kk < 3) v = p.w[((long)ci * p.CI + oo) * 3 + kk];       // W (co_f = ci, ci_f = oo, k)
            break;
        }
        case RP_BWD_CONV:
            if (o < p.c_n && ci < p.CO) v = p.w[((long)ci * p.CI + p.c_lo + o) * p.K + (p.K - 1 - slot)];
            break;
        case RP_BWD_UP:
            if (slot >= 1 && o < p.CI && ci < p.CO) v = p.w[((long)o * p.CO + ci) * 4 + (slot - 1)];   // Wt (ci_f = o, co_f = ci, kk)
            break;
        case RP_BWD_FINAL:
            if (ci < p.CO) v = p.w[(long)ci * p.CI + o];                                      // Wf (td = ci, dim = o)
            break;
    }
    p.dst[d] = v;
}

// Every image of a refresh in ONE launch: descriptor table + block ranges in device memory (cached by the caller
// while the parameter tensors keep their addresses), block b finds its descriptor by bisection over first[].
__global__ __launch_bounds__(256) void repack_many_kernel(const RepackParams* descs, const int* first, int ndesc) {
    int lo = 0, hi = ndesc - 1;                  // first[lo] <= blockIdx.x < first[lo + 1]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const RepackParams p = descs[lo];
    const long d = (long)((int)blockIdx.x - first[lo]) * 256 + threadIdx.x;
    if (d >= p.n) return;
    const int j = (int)(d % p.kg);
    long r = d / p.kg;
    const int o = (int)(r % p.M);
    r /= p.M;
    const int slot = (int)(r % p.wtaps);
    const int ci = (int)(r / p.wtaps) * p.kg + j;
    float v = 0.0f;
    switch (p.mode) {
        case RP_FWD:
            if (ci < p.CI) {
                if (slot < p.K) v = p.w[((long)o * p.CI + ci) * p.K + slot];
                else if (p.ride != nullptr) v = p.ride[(long)o * p.CI + ci];
            }
            break;
        case RP_FWD_UP: case RP_BWD_DOWN: {
            const int co = p.M >> 1, half = o >= co, oo = o - half * co;
            const int kk = half == 0 ? (slot == 0 ? 3 : 1) : (slot == 0 ? 2 : 0);
            if (p.mode == RP_FWD_UP) { if (ci < p.CI) v = p.w[((long)ci * co + oo) * 4 + kk]; }
            else if (ci < p.CO && kk < 3) v = p.w[((long)ci * p.CI + oo) * 3 + kk];
            break;
        }
        case RP_BWD_CONV:
            if (o < p.c_n && ci < p.CO) v = p.w[((long)ci * p.CI + p.c_lo + o) * p.K + (p.K - 1 - slot)];
            break;
        case RP_BWD_UP:
            if (slot >= 1 && o < p.CI && ci < p.CO) v = p.w[((long)o * p.CO + ci) * 4 + (slot - 1)];
            break;
        case RP_BWD_FINAL:
            if (ci < p.CO) v = p.w[(long)ci * p.CI + o];
            break;
    }
    p.dst[d] = v;
}

// The small tensors of a refresh (biases, GroupNorm parameters, time-MLP tensors) in one launch: blockIdx.y picks
// the copy, descriptors travel as kernel arguments.
constexpr int COPY_MAX = 150;
struct CopyMany { float* dst[COPY_MAX]; const float* src[COPY_MAX]; int32_t n[COPY_MAX]; };
__global__ __launch_bounds__(256) void copy_many_kernel(const CopyMany a) {
    const int n = a.n[blockIdx.y];
    const float* src = a.src[blockIdx.y];
    float* dst = a.dst[blockIdx.y];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = src[i];
}

// out[i] = sum_k slab[k][i].  Thread (i, g) of a block adds slabs g, g + 4, g + 8, ... of float4 i (four loads in
// flight per thread, four times the threads of a one-thread-per-element loop: a level-0 conv of PointMaze has 64
// slabs of 82 k floats), the four partial sums meet in LDS and are added in group order — fixed order, no atomics.
// n4 = n / 4 (n is a multiple of 4: one of M, C is a multiple of 32).  grid = ceil(n4 / 64), 256 threads.
__global__ __launch_bounds__(256) void sum_slabs_kernel(float* out, const float* slab, long n4, int ks) {
    __shared__ float4 red[3][64];
    const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + col;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (i < n4) {
        const float4* s4 = reinterpret_cast<const float4*>(slab) + i;
        int k = g;
        for (; k + 4 < ks; k += 8) {
            const float4 u = s4[(long)k * n4], v = s4[(long)(k + 4) * n4];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
        }
        if (k < ks) { const float4 u = s4[(long)k * n4]; a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (g > 0) red[g - 1][col] = a;
    __syncthreads();
    if (g == 0 && i < n4) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { const float4 v = red[q][col]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        reinterpret_cast<float4*>(out)[i] = a;
    }
}

// dst[r][0..c) (+)= src[r][off .. off + c)   (src rows of ld floats; c, off, ld multiples of 4): one side of the gradient
// of an identity residual over a channel concat — d [x | skip] = [d out[:, :c0] | d out[:, c0:]]
__global__ void take_cols_kernel(float* dst, const float* src, long rows, int c, int ld, int off, int have) {
    const int q = c >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * q) return;
    const long r = i / q;
    const int j = (int)(i - r * q) * 4;
    float4 v = *reinterpret_cast<const float4*>(src + r * ld + off + j);
    float4* d = reinterpret_cast<float4*>(dst + r * c + j);
    if (have) { const float4 o = *d; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    *d = v;
}

// y[i] += x[i]   (n4 float4s)
__global__ void add_inplace_kernel(float* y, const float* x, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 a = reinterpret_cast<float4*>(y)[i];
    const float4 b = reinterpret_cast<const float4*>(x)[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    reinterpret_cast<float4*>(y)[i] = a;
}

// part[b][c] = sum_l g[(b*L + l) * ld + c]   (bias gradients of convs without GroupNorm), grid = B
__global__ void row_partial_sums_kernel(float* part, const float* g, int L, int ld, int C) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.0f;
        for (int l = 0; l < L; ++l) acc += g[((long)b * L + l) * ld + c];
        part[(long)b * C + c] = acc;
    }
}

// out[c] = sum_b part[b * stride + c].  Block = 32 columns x 8 row groups: thread (col, rg) adds rows
// rg, rg + 8, ... (four independent chains of loads in flight), the eight partials meet in LDS and are
// added in group order — fixed order, bit-reproducible.  grid = ceil(C / 32), 256 threads.
__global__ __launch_bounds__(256) void col_sums_kernel(float* out, const float* part, int B, int stride, int C) {
    __shared__ float red[8][33];
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + col;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < C) {
        int b = rg;
        for (; b + 24 < B; b += 32) {
            a0 += part[(long)b * stride + c];
            a1 += part[(long)(b + 8) * stride + c];
            a2 += part[(long)(b + 16) * stride + c];
            a3 += part[(long)(b + 24) * stride + c];
        }
        for (; b < B; b += 8) a0 += part[(long)b * stride + c];
    }
    red[rg][col] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rg == 0 && c < C) {
        float v = red[0][col];
#pragma unroll
        for (int k = 1; k < 8; ++k) v += red[k][col];
        out[c] = v;
    }
}

// The same for MANY [B][C] arrays in one launch (blockIdx.y picks the array): every layer of a backward pass parks
// its per-sample partial sums (d gamma, d beta, d bias) in a region of its own and ONE launch at the end of the pass
// reduces them all — 45 launches of ~5 us each on a PointMaze step become one.  The descriptors travel as kernel
// arguments (no upload): at most COLS_MAX per launch.
constexpr int COLS_MAX = 120;
struct ColSumsMany { float* out[COLS_MAX]; const float* part[COLS_MAX]; int32_t C[COLS_MAX]; };
__global__ __launch_bounds__(256) void col_sums_many_kernel(const ColSumsMany a, int B) {
    __shared__ float red[8][33];
    const int C = a.C[blockIdx.y];
    if ((int)blockIdx.x * 32 >= C) return;               // (whole blocks leave)
    const float* part = a.part[blockIdx.y];
    float* out = a.out[blockIdx.y];
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + col;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < C) {
        int b = rg;
        for (; b + 24 < B; b += 32) {
            a0 += part[(long)b * C + c];
            a1 += part[(long)(b + 8) * C + c];
            a2 += part[(long)(b + 16) * C + c];
            a3 += part[(long)(b + 24) * C + c];
        }
        for (; b < B; b += 8) a0 += part[(long)b * C + c];
    }
    red[rg][col] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rg == 0 && c < C) {
        float v = red[0][col];
#pragma unroll
        for (int k = 1; k < 8; ++k) v += red[k][col];
        out[c] = v;
    }
}

// --------------------------------------------------------------------- GroupNorm + Mish backward
// Forward (temporal_unet.py:57-76,106-122):  h = conv(x) + bias;  xh = (h - mean) * rstd;  u = gamma xh + beta;
// a = Mish(u) (+ temb[b, c]).  Given dA:
//   dU = dA * mish'(u);  d gamma_c += sum dU xh;  d beta_c += sum dU;  d temb[b, c] = sum_l dA
//   dxh = dU gamma;  dH = rstd * (dxh - mean_pair(dxh) - xh * mean_pair(dxh * xh));  d bias_c += sum dH
// One block per (sample, group).  Per-channel sums are written per sample (part_*[b][C]) and reduced over
// the batch by col_sums_kernel: fixed order, no atomics.
struct GnBwdParams {
    const float* dA;      // [B*L][C]
    const float* h;       // [B*L][C]  pre-normalisation output saved by the training forward
    const float* stats;   // [B][8][2] mean, rstd
    const float* gamma; const float* beta;
    float* dH;            // [B*L][C]
    float* part_dgamma; float* part_dbeta; float* part_dbias;     // [B][C]
    float* dtemb;         // [B][temb_stride] + temb_off, or nullptr
    int32_t temb_stride;
    int32_t C, L, cpg;
    int32_t B;
    int32_t lreal;        // > 0: only the first lreal positions of a sample exist (zero-padded horizon): the rest take no part
                          // in the sums and get dH = 0
    int32_t cpg_real;     // > 0: channels of a group that exist (zero-padded groups: gamma = 0 on the padding keeps it out of
                          // the two pair sums; only their divisor changes.  dH on the padding is not zero and meets zero weights)
};

__device__ __forceinline__ float mish_grad_f32(float u) {
    // d/du [u tanh(softplus(u))] = T + u * sigmoid(u) * (1 - T^2),  T = w / (w + 2), w = e^u (e^u + 2)
    //                            = T + u * 4 n (n + 1) / (w + 2)^2,  n = e^u
    if (u > 20.0f) return 1.0f;
    const float n = expf(u);
    const float w = n * (n + 2.0f);
    const float d = w + 2.0f;
    return w / d + u * (4.0f * n * (n + 1.0f)) / (d * d);
}

constexpr int GNB_THREADS = 256;
__global__ __launch_bounds__(GNB_THREADS) void gn_mish_bwd_kernel(const GnBwdParams p) {
    __shared__ float red[GNB_THREADS * 16];
    const int b = blockIdx.x, g = blockIdx.y;
    const int tid = threadIdx.x;
    const int C = p.C, L = p.L, cpg = p.cpg;
    const int nq = cpg >> 2;                       // channel quads of the group (a power of two <= 64)
    const int q = tid & (nq - 1), lg = tid / nq;   // this thread's quad and first position
    const int nlg = GNB_THREADS / nq;              // positions handled concurrently
    const int cbase = g * cpg + 4 * q;
    const float mean = p.stats[((long)b * 8 + g) * 2], rstd = p.stats[((long)b * 8 + g) * 2 + 1];
    const float4 gam = ldg4(p.gamma + cbase), bet = ldg4(p.beta + cbase);
    const int Lr = p.lreal > 0 ? p.lreal : L;
    const float inv_n = 1.0f / (float)((p.cpg_real > 0 ? p.cpg_real : cpg) * Lr);

    // pass 1: the two pair sums
    float s1 = 0.0f, s2 = 0.0f;
    for (int l = lg; l < Lr; l += nlg) {
        const long off = ((long)b * L + l) * C + cbase;
        const float4 hv = ldg4(p.h + off), da = ldg4(p.dA + off);
        const float xh[4] = {(hv.x - mean) * rstd, (hv.y - mean) * rstd, (hv.z - mean) * rstd, (hv.w - mean) * rstd};
        const float gm[4] = {gam.x, gam.y, gam.z, gam.w}, bt[4] = {bet.x, bet.y, bet.z, bet.w};
        const float dav[4] = {da.x, da.y, da.z, da.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dxh = dav[j] * mish_grad_f32(gm[j] * xh[j] + bt[j]) * gm[j];
            s1 += dxh;
            s2 += dxh * xh[j];
        }
    }
    red[tid] = s1; red[GNB_THREADS + tid] = s2;
    __syncthreads();
    for (int w = GNB_THREADS / 2; w > 0; w >>= 1) {            // fixed-shape tree: deterministic
        if (tid < w) { red[tid] += red[tid + w]; red[GNB_THREADS + tid] += red[GNB_THREADS + tid + w]; }
        __syncthreads();
    }
    const float m1 = red[0] * inv_n, m2 = red[GNB_THREADS] * inv_n;
    __syncthreads();

    // pass 2: dH and the per-channel sums of this sample
    float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f}, pbias[4] = {0.f, 0.f, 0.f, 0.f},
          pt[4] = {0.f, 0.f, 0.f, 0.f};
    for (int l = Lr + lg; l < L; l += nlg)        // zero-padded horizon: the padding's gradient is zero
        *reinterpret_cast<float4*>(p.dH + ((long)b * L + l) * C + cbase) = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = lg; l < Lr; l += nlg) {
        const long off = ((long)b * L + l) * C + cbase;
        const float4 hv = ldg4(p.h + off), da = ldg4(p.dA + off);
        const float xh[4] = {(hv.x - mean) * rstd, (hv.y - mean) * rstd, (hv.z - mean) * rstd, (hv.w - mean) * rstd};
        const float gm[4] = {gam.x, gam.y, gam.z, gam.w}, bt[4] = {bet.x, bet.y, bet.z, bet.w};
        const float dav[4] = {da.x, da.y, da.z, da.w};
        float dh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float du = dav[j] * mish_grad_f32(gm[j] * xh[j] + bt[j]);
            const float dxh = du * gm[j];
            dh[j] = rstd * (dxh - m1 - xh[j] * m2);
            pg[j] += du * xh[j];
            pb[j] += du;
            pbias[j] += dh[j];
            pt[j] += dav[j];
        }
        *reinterpret_cast<float4*>(p.dH + off) = make_float4(dh[0], dh[1], dh[2], dh[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[tid * 16 + j] = pg[j]; red[tid * 16 + 4 + j] = pb[j];
        red[tid * 16 + 8 + j] = pbias[j]; red[tid * 16 + 12 + j] = pt[j];
    }
    __syncthreads();
    if (lg == 0) {                                           // one thread per quad adds its nlg partials in order
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = red[q * 16 + j];
        for (int k = 1; k < nlg; ++k)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] += red[(k * nq + q) * 16 + j];
        const long po = (long)b * C + cbase;
        *reinterpret_cast<float4*>(p.part_dgamma + po) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(p.part_dbeta + po) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        *reinterpret_cast<float4*>(p.part_dbias + po) = make_float4(acc[8], acc[9], acc[10], acc[11]);
        if (p.dtemb != nullptr) {
            float* t = p.dtemb + (long)b * p.temb_stride + cbase;
            t[0] = acc[12]; t[1] = acc[13]; t[2] = acc[14]; t[3] = acc[15];
        }
    }
}

// The same with ONE WAVE per (sample, group) pair, for pairs of up to 16 x 64 float4 (every layer of the three
// BASELINE nets at horizon 32): both tensors are read once and stay in registers between the two passes, the pair
// sums are wave reductions (xor butterflies: a fixed tree, bit-reproducible) — no LDS, no block barrier.  Lane
// mapping: float4 e = v * 64 + lane covers channel quad q = e % nq at position l = e / nq (nq = cpg / 4, a power of
// two <= 64, so a lane keeps its quad for every v); per-channel sums meet across the lanes that share q.
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s, 64);
    return v;
}
template <int NV>
__global__ __launch_bounds__(256) void gn_mish_bwd_wave_kernel(const GnBwdParams p) {
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= p.B * 8) return;                   // (whole waves leave: no barrier below)
    const int b = pair >> 3, g = pair & 7;
    const int C = p.C, L = p.L, cpg = p.cpg;
    const int nq = cpg >> 2, count = nq * L;
    const int nq_shift = 31 - __builtin_clz(nq);
    const int q = lane & (nq - 1);
    const int cbase = g * cpg + 4 * q;
    const float mean = p.stats[((long)b * 8 + g) * 2], rstd = p.stats[((long)b * 8 + g) * 2 + 1];
    const float4 gam = ldg4(p.gamma + cbase), bet = ldg4(p.beta + cbase);
    const float gm[4] = {gam.x, gam.y, gam.z, gam.w}, bt[4] = {bet.x, bet.y, bet.z, bet.w};
    const int Lr = p.lreal > 0 ? p.lreal : L;      // positions that exist (zero-padded horizon)
    const float inv_n = 1.0f / (float)((p.cpg_real > 0 ? p.cpg_real : cpg) * Lr);

    float4 hv[NV], da[NV];
    long off[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {                 // unconditional loads (clamped index), masked at the use
        const int e = min(v * 64 + lane, count - 1);
        off[v] = ((long)b * L + (e >> nq_shift)) * C + cbase;
        hv[v] = ldg4(p.h + off[v]);
        da[v] = ldg4(p.dA + off[v]);
    }
    // pass 1: du = dA mish'(u) and xh replace dA and h in the registers; the two pair sums; sum_l dA
    float s1 = 0.0f, s2 = 0.0f;
    float pt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const bool live = v * 64 + lane < count && ((v * 64 + lane) >> nq_shift) < Lr;
        float x[4] = {hv[v].x, hv[v].y, hv[v].z, hv[v].w}, d[4] = {da[v].x, da[v].y, da[v].z, da[v].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (x[j] - mean) * rstd;
            const float du = live ? d[j] * mish_grad_f32(gm[j] * xh + bt[j]) : 0.0f;
            pt[j] += live ? d[j] : 0.0f;
            const float dxh = du * gm[j];
            s1 += dxh;
            s2 += dxh * xh;
            x[j] = xh; d[j] = du;
        }
        hv[v] = make_float4(x[0], x[1], x[2], x[3]);
        da[v] = make_float4(d[0], d[1], d[2], d[3]);
    }
    const float m1 = wave_sum_all(s1) * inv_n, m2 = wave_sum_all(s2) * inv_n;
    // pass 2: dH and this sample's per-channel sums
    float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f}, pbias[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const bool inside = v * 64 + lane < count;                                 // an element of the pair
        const bool live = inside && ((v * 64 + lane) >> nq_shift) < Lr;            // ... at a position that exists
        const float x[4] = {hv[v].x, hv[v].y, hv[v].z, hv[v].w}, d[4] = {da[v].x, da[v].y, da[v].z, da[v].w};
        float dh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dh[j] = live ? rstd * (d[j] * gm[j] - m1 - x[j] * m2) : 0.0f;
            pg[j] += d[j] * x[j];
            pb[j] += d[j];
            pbias[j] += dh[j];
        }
        if (inside) *reinterpret_cast<float4*>(p.dH + off[v]) = make_float4(dh[0], dh[1], dh[2], dh[3]);   // (padding: zeros)
    }
    for (int s = nq; s < 64; s <<= 1)              // lanes q, q + nq, q + 2 nq, ... hold the same channels
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pg[j] += __shfl_xor(pg[j], s, 64);
            pb[j] += __shfl_xor(pb[j], s, 64);
            pbias[j] += __shfl_xor(pbias[j], s, 64);
            pt[j] += __shfl_xor(pt[j], s, 64);
        }
    if (lane < nq) {
        const long po = (long)b * C + cbase;
        *reinterpret_cast<float4*>(p.part_dgamma + po) = make_float4(pg[0], pg[1], pg[2], pg[3]);
        *reinterpret_cast<float4*>(p.part_dbeta + po) = make_float4(pb[0], pb[1], pb[2], pb[3]);
        *reinterpret_cast<float4*>(p.part_dbias + po) = make_float4(pbias[0], pbias[1], pbias[2], pbias[3]);
        if (p.dtemb != nullptr) {
            float* t = p.dtemb + (long)b * p.temb_stride + cbase;
            t[0] = pt[0]; t[1] = pt[1]; t[2] = pt[2]; t[3] = pt[3];
        }
    }
}

}  // namespace dad
