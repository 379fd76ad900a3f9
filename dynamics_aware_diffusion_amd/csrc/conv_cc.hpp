// conv_cc.hpp — the small-batch ("consumer-combine") form of the TemporalUnet convs.
//
// At batch 1..8 a conv layer has a few hundred rows at most and the step is a chain of ~30
// all-to-all seams: every launch costs a kernel boundary plus dependent memory round trips, the
// arithmetic is noise.  The batch-256 kernel (conv_gemm.hpp) splits K over blocks and lets the last
// arriver of a tile combine the partial tiles and run the GroupNorm / Mish epilogue; at batch 1 that
// in-kernel seam (release fence, ticket, acquire fence, serial slab reads) is most of each launch
// (19 us for a 512->512 layer, rocprofv3).  This file turns the seam into the kernel boundary that
// is there anyway:
//
//   * a conv launch ONLY produces partial sums: block (K slice, M tile, N tile) multiplies its
//     slice of input channels and stores a raw partial tile — no bias, no norm, no ticket;
//   * the CONSUMER of a tensor finishes it while staging its input: it adds the producer's partial
//     slabs in slice order, the bias, normalises each (sample, group) with GroupNorm statistics it
//     computes itself (two-pass, fp32), applies Mish, the time embedding and the residual, and keeps
//     the result in LDS as its A operand.  The blocks of M tile 0 also write that finished slice to
//     the tensor's ordinary activation buffer, which later readers (residual adds, skip
//     connections) use as is.
//
// Same arithmetic as the reference chain F.conv1d -> F.group_norm -> F.mish -> adds
// (m_diffuser/models/temporal_unet.py:57-122, :35-54), fp32 throughout, fixed summation order
// (bit-reproducible run to run); only the order of fp32 additions differs from conv_gemm.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_gemm.hpp"
#include "pointwise.hpp"

namespace dad {

// One input of a CC conv: either a finished tensor (nsl == 0) or a tensor still in pieces.
struct CcSrc {
    const float* data;     // nsl == 0: [rows][C] finished tensor;  nsl > 0: [nsl][rows][C] partial sums
    int32_t nsl;           // partial slabs to add (<= 2 * CC_MAX_SLABS; more than CC_MAX_SLABS cost a second round trip)
    int32_t C;             // channels (row stride) of this tensor
    int32_t rows;          // rows of the whole tensor (B * L): slab stride = rows * C
    int32_t cpg;           // channels per GroupNorm group (when gamma != nullptr)
    const float* bias;     // [C]
    const float* gamma;    // [C] GroupNorm weight -> normalise + Mish; nullptr: plain sum + bias
    const float* beta;     // [C]
    const float* temb;     // [C] time-embedding row added after Mish, or nullptr
    const float* res;      // [rows][C] finished residual tensor added last, or nullptr
    const float* rslab;    // [nrs][rows][C] partial sums of the riding 1x1 residual conv, or nullptr
    const float* rbias;    // [C]
    int32_t nrs;
    int32_t pad_;
    float* mat;            // [rows][C] where M-tile-0 blocks store the finished values, or nullptr
};

struct CcParams {
    CcSrc src0, src1;      // virtual channel concat [src0 | src1]; src1.C == 0: none
    const float* w;        // packed [cin_pad/16][wtaps][M][16]
    int32_t wtaps;         // tap slots of the weight image
    int32_t cin0, cin1;    // channels taken from src0 / src1
    int32_t M;             // GEMM columns (2 * C_out for the transposed conv)
    int32_t B, Lin, Lout;  // batch rows, per-sample GEMM lengths
    int32_t lshift, lshift_in;
    int32_t interleave;    // transposed conv: column m < M/2 -> row 2l, else row 2l + 1
    int32_t slice_ch;      // input channels per K slice: multiple of 32 and of the source's group width
    float* oslab;          // [kslices][out_rows][out_cols] this conv's partial sums
    float* orslab;         // same for the riding 1x1 conv, or nullptr
    int32_t out_rows;      // rows of the output tensor (B * L_final)
#ifdef DAD_STAMPS
    unsigned long long* stamps;   // diagnostic build only: [8] realtime stamps of block 0
#endif
};
#ifdef DAD_STAMPS
#define CC_STAMP(i) do { if (p.stamps != nullptr && threadIdx.x == 0 && blockIdx.x + blockIdx.y + blockIdx.z == 0) p.stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CC_STAMP(i) do {} while (0)
#endif

#ifdef DAD_STAMPS
__device__ unsigned long long* g_cc_gn_stamps = nullptr;     // set per launch by lane 0 of block 0
#define CC_GN_STAMP(i) do { if (g_cc_gn_stamps != nullptr && lane == 0 && wave == 0 && blockIdx.x + blockIdx.y + blockIdx.z == 0 && pr == wave) g_cc_gn_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CC_GN_STAMP(i) do {} while (0)
#endif
constexpr int CC_MAX_SLABS = 8;
constexpr int CC_THREADS = 512;

// Kernel arguments are fetched from the kernarg segment by scalar loads that hipcc places in the
// basic block of their FIRST USE, each batch followed by its own s_waitcnt: a kernel that first
// touches an argument deep in its prologue pays a scalar-memory round trip there (the ISA of
// conv_cc had six such batches in series before its first MFMA; -DDAD_STAMPS showed 1.8 us between
// issuing the weight loads and the first load of the input).  Naming every argument as an SGPR
// operand of an empty asm in the entry block makes all of them part of ONE batch at kernel entry.
#define CC_PIN_SRC(S)                                                                             \
    asm volatile("" ::"s"(S.data), "s"(S.nsl), "s"(S.C), "s"(S.rows), "s"(S.cpg), "s"(S.bias),      \
                 "s"(S.gamma), "s"(S.beta), "s"(S.temb), "s"(S.res), "s"(S.rslab), "s"(S.rbias),    \
                 "s"(S.nrs), "s"(S.mat))

// Sum over the 64 lanes, every lane gets it: DPP permutes inside each 16-lane row, v_readlane across
// rows (no LDS crossbar hops: the shuffle form of this cost ~0.4 us per reduction at this clock).
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);           // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);           // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);          // row_half_mirror
    v += dpp_f32<0x140>(v);          // row_mirror
    return rows_sum(v, 64, 0);
}

// The GroupNorm'd form of the same thing: one (sample, group) pair per wave at a time, entirely in
// registers.  A pair has cnt = L * cpg <= CC_MAX_PAIR elements = at most CC_F4 float4 per lane; every
// global load of the pair (partial slabs, bias, gamma, beta, time embedding, residual or the riding
// conv's partial slabs) is issued before the first use, the two-pass mean / biased variance runs on
// wave shuffles, and the finished values go to LDS (and to s.mat).  No barrier inside; the caller
// synchronises the block afterwards.
constexpr int CC_F4 = 4;
constexpr int CC_MAX_PAIR = CC_F4 * 4 * 64;
// BIG: tensors with 9..16 partial slabs (1024-channel inputs) add them in a second, dependent batch
// of loads; a kernel-level template parameter, so that the common kernels keep their register
// budget (with both forms in one kernel the allocator spilled the staged weights).
// Windowed tiles (layers of more than 32 positions: horizon 64 on the get_action path): the tile covers rows
// [wl0, wl0 + seg - 2 pad) of ONE sample; the pair statistics still run over the whole sample, but only the rows
// of the window and its halo go to LDS (at row l - wl0 + pad) and only the window's own rows are published.
// wl0 < 0: whole samples per tile (every row is staged at smp * seg + pad + l).
// WINOK: the kernel can be launched with windowed tiles at all (5-tap stride-1 convs on 32-row tiles with at most 8
// input slabs); elsewhere wl0 is -1 at compile time and none of this costs a register (the 9..16-slab variants sit
// at 253 VGPRs).
template <bool RIDE, bool BIG, bool WINOK>
__device__ __forceinline__ void cc_build_input_gn(const CcSrc& s, float* dst, int ld, int r0, int nrows_valid,
                                                  int L, int lshiftL, int seg, int pad, int c0, int nch,
                                                  bool publish, int lane, int wave, int wl0_arg) {
    const int wl0 = WINOK ? wl0_arg : -1;
    const int cpg = s.cpg;
    const int groups = nch / cpg;                           // whole groups (host guarantees)
    const int nsmp = nrows_valid >> lshiftL;
    const int cnt = L * cpg;
    const int cnt4 = cnt >> 2;
    const float inv_cnt = 1.0f / (float)cnt;
    const int cq = cpg >> 2;                                // float4 per row of the pair
    const long sstride = (long)s.rows * s.C;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int pr = wave; pr < nsmp * groups; pr += CC_THREADS / 64) {
        const int smp = pr / groups, g = pr - smp * groups;
        float4 v[CC_F4], ex[CC_F4], gam[CC_F4], bet[CC_F4];
        long off[CC_F4];
        int lo[CC_F4];
        unsigned onmask = 0, pubmask = 0;              // bit k: float4 k belongs to the pair / is published by this tile
        CC_GN_STAMP(8);
#pragma unroll
        for (int k = 0; k < CC_F4; ++k) {
            v[k] = zero4; ex[k] = zero4; gam[k] = zero4; bet[k] = zero4; off[k] = 0; lo[k] = -1;
            if (k * 64 >= cnt4) continue;                   // wave-uniform: this pair has fewer float4
            // keep the loads of at most two float4 per lane in flight (register budget)
            if (k == 2) __builtin_amdgcn_sched_barrier(0);
            const int j = lane + k * 64;                    // float4 index inside the pair
            const bool on = j < cnt4;
            const int jj = on ? j : 0;
            const int l = jj / cq, cl = (jj - l * cq) * 4;
            const int c = c0 + g * cpg + cl;
            off[k] = (long)(r0 + smp * L + l) * s.C + c;
            if constexpr (!WINOK) {
                lo[k] = on ? (smp * seg + pad + l) * ld + g * cpg + cl : -1;     // (lo >= 0 <=> part of the pair)
            } else {
                const int wr = l - wl0 + pad;          // row of the window stage
                lo[k] = (on && wr >= 0 && wr < seg) ? wr * ld + g * cpg + cl : -1;
                if (on && wr >= pad && wr < seg - pad) pubmask |= 1u << k;
                if (on) onmask |= 1u << k;
            }
            // every load below is unconditional (lanes past the pair re-read element 0; slabs that do
            // not exist re-read the last one and are not added): they all fly together
            // (absent operands re-read something valid and are masked afterwards: a load under a
            // branch would make the compiler drain every outstanding load at the branch)
            float4 part[CC_MAX_SLABS], rp[CC_MAX_SLABS];
#pragma unroll
            for (int q = 0; q < CC_MAX_SLABS; ++q) part[q] = ldg4(s.data + (long)min(q, s.nsl - 1) * sstride + off[k]);
            const float4 b = ldg4(s.bias + c);
            gam[k] = ldg4(s.gamma + c);
            bet[k] = ldg4(s.beta + c);
            const bool has_t = s.temb != nullptr, has_r = s.res != nullptr;
            const float4 tv = ldg4((has_t ? s.temb : s.bias) + c);
            const float4 rv = ldg4((has_r ? s.res : s.data) + off[k]);
            const int nrs = RIDE ? s.nrs : 1;
            float4 rb = zero4;
#pragma unroll
            for (int q = 0; q < CC_MAX_SLABS; ++q) rp[q] = zero4;
            if constexpr (RIDE) {
#pragma unroll
                for (int q = 0; q < CC_MAX_SLABS; ++q) rp[q] = ldg4(s.rslab + (long)min(q, nrs - 1) * sstride + off[k]);
                rb = ldg4(s.rbias + c);
            }
            // selects instead of branches (no load can be sunk under a condition) and instead of 0 / 1
            // multipliers (0 * Inf of a stand-in operand would poison the sum)
            float4 r = rp[0];
#pragma unroll
            for (int q = 1; q < CC_MAX_SLABS; ++q) {
                const float mq = q < nrs ? 1.0f : 0.0f;
                r.x = fmaf(rp[q].x, mq, r.x); r.y = fmaf(rp[q].y, mq, r.y);
                r.z = fmaf(rp[q].z, mq, r.z); r.w = fmaf(rp[q].w, mq, r.w);
            }
            if constexpr (RIDE && BIG) {
                if (nrs > CC_MAX_SLABS) {                   // 9..16 ride slabs: second batch
#pragma unroll
                    for (int q = 0; q < CC_MAX_SLABS; ++q) rp[q] = ldg4(s.rslab + (long)min(CC_MAX_SLABS + q, nrs - 1) * sstride + off[k]);
#pragma unroll
                    for (int q = 0; q < CC_MAX_SLABS; ++q) {
                        const float mq = CC_MAX_SLABS + q < nrs ? 1.0f : 0.0f;
                        r.x = fmaf(rp[q].x, mq, r.x); r.y = fmaf(rp[q].y, mq, r.y);
                        r.z = fmaf(rp[q].z, mq, r.z); r.w = fmaf(rp[q].w, mq, r.w);
                    }
                }
            }
            const float4 ts = has_t ? tv : zero4, rs = has_r ? rv : zero4;
            const float4 ds = RIDE ? make_float4(r.x + rb.x, r.y + rb.y, r.z + rb.z, r.w + rb.w) : zero4;
            float4 e;
            e.x = (ts.x + rs.x) + ds.x;
            e.y = (ts.y + rs.y) + ds.y;
            e.z = (ts.z + rs.z) + ds.z;
            e.w = (ts.w + rs.w) + ds.w;
            ex[k] = e;
            float4 a = part[0];
#pragma unroll
            for (int q = 1; q < CC_MAX_SLABS; ++q) {
                const float mq = q < s.nsl ? 1.0f : 0.0f;   // x * 1 + a is exact: same sums as a branch
                a.x = fmaf(part[q].x, mq, a.x); a.y = fmaf(part[q].y, mq, a.y);
                a.z = fmaf(part[q].z, mq, a.z); a.w = fmaf(part[q].w, mq, a.w);
            }
            if (BIG && s.nsl > CC_MAX_SLABS) {              // 9..16 slabs (1024-channel inputs): a second,
#pragma unroll                                              // dependent batch of loads, in slice order
                for (int q = 0; q < CC_MAX_SLABS; ++q) part[q] = ldg4(s.data + (long)min(CC_MAX_SLABS + q, s.nsl - 1) * sstride + off[k]);
#pragma unroll
                for (int q = 0; q < CC_MAX_SLABS; ++q) {
                    const float mq = CC_MAX_SLABS + q < s.nsl ? 1.0f : 0.0f;
                    a.x = fmaf(part[q].x, mq, a.x); a.y = fmaf(part[q].y, mq, a.y);
                    a.z = fmaf(part[q].z, mq, a.z); a.w = fmaf(part[q].w, mq, a.w);
                }
            }
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            v[k] = on ? a : zero4;
        }
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < CC_F4; ++k) sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        CC_GN_STAMP(9);
        const float mean = wave_sum(sum) * inv_cnt;
        float sq = 0.0f;
#pragma unroll
        for (int k = 0; k < CC_F4; ++k)
            if (WINOK ? (onmask >> k & 1) != 0 : lo[k] >= 0) {
                const float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
                sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_cnt + 1e-5f);
        CC_GN_STAMP(10);
#pragma unroll
        for (int k = 0; k < CC_F4; ++k) {
            if (lo[k] < 0) continue;
            float4 y;
            y.x = mish_fast_f32((v[k].x - mean) * rstd * gam[k].x + bet[k].x) + ex[k].x;
            y.y = mish_fast_f32((v[k].y - mean) * rstd * gam[k].y + bet[k].y) + ex[k].y;
            y.z = mish_fast_f32((v[k].z - mean) * rstd * gam[k].z + bet[k].z) + ex[k].z;
            y.w = mish_fast_f32((v[k].w - mean) * rstd * gam[k].w + bet[k].w) + ex[k].w;
            *reinterpret_cast<float4*>(dst + lo[k]) = y;
            if (publish && s.mat != nullptr && (!WINOK || (pubmask >> k & 1))) *reinterpret_cast<float4*>(s.mat + off[k]) = y;
        }
        CC_GN_STAMP(11);
    }
}

// Finished values of rows [r0, r0 + nrows) x channels [c0, c0 + nch) of `s` into LDS:
//   dst[(row_map(r)) * ld + (c - c0)]   with row_map(r) = (r / L) * seg + pad + (r % L)
// (seg = L + 2 pad: zero halo rows around every sample; pass seg = L, pad = 0 for none).
// nch is a multiple of 4 except for the ragged external trajectory (C = transition_dim), whose
// channels beyond C read as zero.  All threads of the block take part; ends with a barrier.
// `publish`: also store the finished values to s.mat.
template <bool BIG, bool WINOK = false>
__device__ __forceinline__ void cc_build_input(const CcSrc& s, float* dst, int ld, int r0, int nrows_valid,
                                               int nrows_tile, int L, int lshiftL, int seg, int pad, int c0,
                                               int nch, bool publish, int tid, int lane, int wave, int wl0_arg = -1) {
    const int wl0 = WINOK ? wl0_arg : -1;
    const int q4 = nch >> 2;
    const long sstride = (long)s.rows * s.C;
    const bool plain = s.nsl == 0;
    const bool gn = !plain && s.gamma != nullptr;
    if (wl0 >= 0) {
        // windowed tile: ONE sample (rows r0 .. r0 + L), stage rows [wl0 - pad, wl0 + seg - pad) of it; the caller
        // zeroed the stage (rows of the halo that fall outside the sample stay zero)
        if (gn) {
            if (s.rslab != nullptr) cc_build_input_gn<true, BIG, WINOK>(s, dst, ld, r0, L, L, lshiftL, seg, pad, c0, nch, publish, lane, wave, wl0);
            else cc_build_input_gn<false, BIG, WINOK>(s, dst, ld, r0, L, L, lshiftL, seg, pad, c0, nch, publish, lane, wave, wl0);
            __syncthreads();
            return;
        }
        for (int i = tid; i < seg * q4; i += CC_THREADS) {
            const int wr = i / q4, q = i - wr * q4;
            const int l = wl0 - pad + wr;
            const int c = c0 + 4 * q;
            if (l < 0 || l >= L || c >= s.C) continue;
            const long off = (long)(r0 + l) * s.C + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (plain) {
                if ((s.C & 3) == 0) {
                    v = ldg4(s.data + off);
                } else {                                    // external trajectory: C = transition_dim
                    const float* g = s.data + off;
                    const int left = s.C - c;
                    v.x = g[0];
                    if (left > 1) v.y = g[1];
                    if (left > 2) v.z = g[2];
                    if (left > 3) v.w = g[3];
                }
            } else {
                float4 part[CC_MAX_SLABS];
#pragma unroll
                for (int k = 0; k < CC_MAX_SLABS; ++k)
                    part[k] = ldg4(s.data + (long)min(k, s.nsl - 1) * sstride + off);
                v = part[0];
#pragma unroll
                for (int k = 1; k < CC_MAX_SLABS; ++k)
                    if (k < s.nsl) { v.x += part[k].x; v.y += part[k].y; v.z += part[k].z; v.w += part[k].w; }
                for (int k = CC_MAX_SLABS; k < s.nsl; ++k) {
                    const float4 u = ldg4(s.data + (long)k * sstride + off);
                    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                }
                const float4 b = ldg4(s.bias + c);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                if (publish && s.mat != nullptr && wr >= pad && wr < seg - pad) *reinterpret_cast<float4*>(s.mat + off) = v;
            }
            *reinterpret_cast<float4*>(dst + wr * ld + 4 * q) = v;
        }
        __syncthreads();
        return;
    }
    // zero halo rows (LDS only)
    if (pad > 0) {
        const int nsmp = nrows_tile >> lshiftL;
        for (int i = tid; i < nsmp * 2 * pad * q4; i += CC_THREADS) {
            const int hr = i / q4, q = i - hr * q4;
            const int smp = hr / (2 * pad), j = hr - smp * (2 * pad);
            const int row = smp * seg + (j < pad ? j : L + j);
            *reinterpret_cast<float4*>(dst + row * ld + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (gn) {
        // GroupNorm'd tensors.  This side of the (uniform) branch issues no global load before the
        // pairs' own: a loop with loads in front of them made hipcc drain the weight loads first —
        // two memory round trips in series instead of one (1.8 us of every such launch).
        for (int i = tid + nrows_valid * q4; i < nrows_tile * q4; i += CC_THREADS) {     // absent samples: zero
            const int r = i / q4, q = i - r * q4;
            *reinterpret_cast<float4*>(dst + ((r >> lshiftL) * seg + pad + (r & (L - 1))) * ld + 4 * q) =
                make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (s.rslab != nullptr) cc_build_input_gn<true, BIG, false>(s, dst, ld, r0, nrows_valid, L, lshiftL, seg, pad, c0, nch, publish, lane, wave, -1);
        else cc_build_input_gn<false, BIG, false>(s, dst, ld, r0, nrows_valid, L, lshiftL, seg, pad, c0, nch, publish, lane, wave, -1);
        __syncthreads();
        return;
    }
    // ---- no norm: partial sums (+ bias), or the finished tensor, into LDS ------------------------
    for (int i = tid; i < nrows_tile * q4; i += CC_THREADS) {
        const int r = i / q4, q = i - r * q4;
        const int smp = r >> lshiftL, l = r & (L - 1);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = c0 + 4 * q;
        if (r < nrows_valid && c < s.C) {
            const long off = (long)(r0 + r) * s.C + c;
            if (plain) {
                if ((s.C & 3) == 0) {
                    v = ldg4(s.data + off);
                } else {                                    // external trajectory: C = transition_dim
                    const float* g = s.data + off;
                    const int left = s.C - c;
                    v.x = g[0];
                    if (left > 1) v.y = g[1];
                    if (left > 2) v.z = g[2];
                    if (left > 3) v.w = g[3];
                }
            } else {
                // all CC_MAX_SLABS loads are issued unconditionally (slabs that do not exist re-read
                // the last one and are not added): branch-free, so they fly together
                float4 part[CC_MAX_SLABS];
#pragma unroll
                for (int k = 0; k < CC_MAX_SLABS; ++k)
                    part[k] = ldg4(s.data + (long)min(k, s.nsl - 1) * sstride + off);
                v = part[0];
#pragma unroll
                for (int k = 1; k < CC_MAX_SLABS; ++k)
                    if (k < s.nsl) { v.x += part[k].x; v.y += part[k].y; v.z += part[k].z; v.w += part[k].w; }
                for (int k = CC_MAX_SLABS; k < s.nsl; ++k) {            // 9..16 slabs
                    const float4 u = ldg4(s.data + (long)k * sstride + off);
                    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                }
                const float4 b = ldg4(s.bias + c);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                // (down / up-sampling convs: bias only) finished here
                if (publish && s.mat != nullptr) *reinterpret_cast<float4*>(s.mat + off) = v;
            }
        }
        *reinterpret_cast<float4*>(dst + (smp * seg + pad + l) * ld + 4 * q) = v;
    }
    __syncthreads();
}

// LDS floats of one conv_cc block: [X rows][slice + 4] + [weight taps][32][slice + 4], or the
// exchange tile [8 waves][32][36] (+ the ride's) after the K loop.
__host__ __device__ inline size_t cc_lds_floats(int slice_ch, int taps, int wtaps, int Lin, int Lout, int nr) {
    const size_t xs = slice_ch + 4;
    const size_t k = (size_t)cc_xrows(taps, Lin, Lout, nr) * xs + (size_t)wtaps * 32 * xs;
    const size_t e = (size_t)2 * 8 * nr * 36;
    return k > e ? k : e;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// grid = (K slices, M / 32, N tiles of NR GEMM rows); 8 waves: every wave owns the whole NR x 32 tile
// and takes every 8th (tap, channel group) unit of the slice (intra-block split-K over all waves).
// NR = 32: v_mfma_f32_32x32x2_f32, units of 8 channels.  NR = 16 (layers of at most 16 positions: at
// batch 1 a 32-row tile would be at most half full): v_mfma_f32_16x16x4_f32 on two 16-channel halves,
// units of 16 channels — the same matrix-pipe cycles per unit for half the padded rows.
// WPT: float4 of weights each thread stages = slice * weight taps * 32 rows / 4 / 512 threads, rounded
// up: 6 covers the widest slice (64 channels, 6 taps).
// WINDOWED: its own instantiation (layers of more than 32 positions), so that the kernels of the horizon-32 plans
// stay exactly what they were (with the window logic as a runtime branch of one kernel the PointMaze batch-1 step
// went from 211 to 224 us).
template <int TAPS, int STRIDE, bool RES, bool BIG, int NR, int WPT = 6, bool WINDOWED = false>
__global__ __launch_bounds__(CC_THREADS) void conv_cc(const CcParams p) {
    static_assert(NR == 16 || NR == 32, "tile rows");
    static_assert(!WINDOWED || (TAPS == 5 && STRIDE == 1 && NR == 32 && !BIG), "windowed tiles: 5-tap stride-1 convs, 32 rows, <= 8 slabs");
    constexpr int PAD = TAPS / 2;
    constexpr int WTAPS = TAPS + (RES ? 1 : 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* const smem4 = reinterpret_cast<float4*>(smem);
    CC_STAMP(0);
    CC_PIN_SRC(p.src0);
    CC_PIN_SRC(p.src1);
    asm volatile("" ::"s"(p.w), "s"(p.wtaps), "s"(p.cin0), "s"(p.cin1), "s"(p.M), "s"(p.B), "s"(p.Lin), "s"(p.Lout),
                 "s"(p.lshift), "s"(p.lshift_in), "s"(p.interleave), "s"(p.slice_ch), "s"(p.oslab), "s"(p.orslab),
                 "s"(p.out_rows));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, h = lane >> 5;
    const int kb = blockIdx.x, mt = blockIdx.y, nt = blockIdx.z;
    const int Lin = p.Lin, Lout = p.Lout, M = p.M;
    // windowed tiles (stride-1 layers of more than NR positions): tile nt = rows [wl0, wl0 + NR) of sample nt / tps
    constexpr bool WINOK = WINDOWED;
    constexpr bool WIN = WINDOWED;
    const int tps = WIN ? Lout / NR : 1;
    const int wl0 = WIN ? (nt % tps) * NR : -1;
    const int SPT = WIN ? 1 : NR >> p.lshift;          // whole samples per tile
    const int SEG = WIN ? NR + 2 * PAD : Lin + 2 * PAD;
    const int XROWS = SPT * SEG;
    const int s0 = WIN ? nt / tps : nt * SPT;
    const int nvalid = min(SPT, p.B - s0);
    const int m0 = mt * 32;
    const int SL = p.slice_ch;
    const int XS = SL + 4;                             // LDS row stride: odd number of 16-byte slots
    const int XS4 = XS >> 2;
    float* const Xb = smem;
    float* const Wb = smem + XROWS * XS;

    // ---- weights of this (K slice, M tile): global -> registers -> LDS, all in flight at once --
    const int c0 = kb * SL;                            // first input channel of the slice
    const int ngr = SL >> 4;                           // 16-channel granules in the slice
    const int n_w4 = ngr * WTAPS * 32 * 4;             // float4 to stage
    float4 wreg[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        // unconditional (threads past the end re-read the last float4; only the LDS store below is
        // predicated): a load under `if (e < n_w4)` got a register copy behind it at the end of the
        // conditional block, with an s_waitcnt vmcnt(0) in front — a full memory round trip between
        // the second and the third weight load of every launch
        const int e = min(tid + i * CC_THREADS, n_w4 - 1);
        const int q = e & 3, mm = (e >> 2) & 31, gt = e >> 7;           // gt = gr * WTAPS + tap
        const int gr = gt / WTAPS, tap = gt - gr * WTAPS;
        wreg[i] = ldg4(p.w + ((long)(((c0 >> 4) + gr) * p.wtaps + tap) * M + m0 + mm) * 16 + q * 4);
    }

    CC_STAMP(1);
#ifdef DAD_STAMPS
    if (threadIdx.x == 0 && blockIdx.x + blockIdx.y + blockIdx.z == 0) g_cc_gn_stamps = p.stamps;
#endif
    // ---- input slice: finish the producer's tensor into LDS (see file header) ----------------
    const bool second = p.cin1 > 0 && c0 >= p.cin0;
    const CcSrc& src = second ? p.src1 : p.src0;
    const int cs0 = second ? c0 - p.cin0 : c0;         // first channel inside that source
    const int cin_src = second ? p.cin1 : p.cin0;
    const int nch = min(SL, ((cin_src - cs0) + 3) & ~3);   // channels to stage (rest of the slice: zero)
    if (nch < SL || WIN) {                             // zero the columns / halo rows the staging leaves untouched
        for (int i = tid; i < XROWS * XS4; i += CC_THREADS) smem4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    }
    cc_build_input<BIG, WINOK>(src, Xb, XS, s0 * Lin, nvalid * Lin, SPT * Lin, Lin, p.lshift_in, SEG, PAD, cs0, nch,
                               mt == 0, tid, lane, wave, wl0);
    CC_STAMP(2);
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int e = tid + i * CC_THREADS;
        if (e < n_w4) {
            const int q = e & 3, mm = (e >> 2) & 31, gt = e >> 7;
            const int gr = gt / WTAPS, tap = gt - gr * WTAPS;
            *reinterpret_cast<float4*>(Wb + (tap * 32 + mm) * XS + gr * 16 + q * 4) = wreg[i];
        }
    }
    __syncthreads();
    CC_STAMP(3);

    // ---- K loop: no barriers, every operand is resident ---------------------------------------
    const int phase_shift = (TAPS == 2 && p.interleave && m0 >= (M >> 1)) ? 1 : 0;
    constexpr int ES = 36;
    float* const E = smem;                             // [8][NR][ES]
    float* const ER = smem + 8 * NR * ES;              // the ride's
    if constexpr (NR == 32) {
        f32x16 acc, acc2, acc3, acc4, accr, accr2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; acc4[r] = 0.f; accr[r] = 0.f; accr2[r] = 0.f; }
        // (windowed tile: l32 >> lshift = 0 and l32 & (Lout - 1) = l32 — the window's rows in order)
        const int arow4 = (((l32 >> p.lshift) * SEG + (l32 & (Lout - 1)) * STRIDE + phase_shift) * XS + 4 * h) >> 2;
        const int brow4 = (XROWS * XS + l32 * XS + 4 * h) >> 2;
        const int G = SL >> 3;                         // 8-channel groups in the slice
        const int U = WTAPS * G;
        // fragments of unit u + 8 are read while unit u's MFMAs run
        auto frag = [&](int u, float4& a, float4& b) {
            const int wtap = u / G, g = u - wtap * G;
            const int tap = (RES && wtap == TAPS) ? PAD : wtap;
            a = smem4[arow4 + tap * XS4 + g * 2];
            b = smem4[brow4 + wtap * 32 * XS4 + g * 2];
        };
        float4 a, b;
        if (wave < U) frag(wave, a, b);
        for (int u = wave; u < U; u += CC_THREADS / 64) {
            const float4 ca = a, cb = b;
            const int un = u + CC_THREADS / 64;
            if (un < U) frag(un, a, b);
            if (RES && u / G == TAPS) {
                accr = __builtin_amdgcn_mfma_f32_32x32x2f32(ca.x, cb.x, accr, 0, 0, 0);
                accr2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ca.y, cb.y, accr2, 0, 0, 0);
                accr = __builtin_amdgcn_mfma_f32_32x32x2f32(ca.z, cb.z, accr, 0, 0, 0);
                accr2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ca.w, cb.w, accr2, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca.x, cb.x, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ca.y, cb.y, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(ca.z, cb.z, acc3, 0, 0, 0);
                acc4 = __builtin_amdgcn_mfma_f32_32x32x2f32(ca.w, cb.w, acc4, 0, 0, 0);
            }
        }
        CC_STAMP(4);
        __syncthreads();                               // all fragment reads done: LDS becomes the exchange tile
        // ---- the 8 waves' partial tiles meet in LDS; one float4 of the block's tile per thread --
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            E[(wave * 32 + row) * ES + l32] = (acc[r] + acc2[r]) + (acc3[r] + acc4[r]);
            if (RES) ER[(wave * 32 + row) * ES + l32] = accr[r] + accr2[r];
        }
    } else {
        // 16-row tile: lane = (row or channel l16, k group g4); a unit is 16 channels of one tap: the
        // float4 at channels 4*g4.. is component j of MFMA j's k index g4 — the same bijection of K
        // for both operands, so any order is a valid summation order.
        const int l16 = lane & 15, g4 = lane >> 4;
        f32x4 c0a = {0.f, 0.f, 0.f, 0.f}, c0b = c0a, c1a = c0a, c1b = c0a, r0 = c0a, r1 = c0a;
        const int arow4 = (((l16 >> p.lshift) * SEG + (l16 & (Lout - 1)) * STRIDE + phase_shift) * XS + 4 * g4) >> 2;
        const int brow4 = (XROWS * XS + l16 * XS + 4 * g4) >> 2;
        const int G = SL >> 4;                         // 16-channel groups in the slice
        const int U = WTAPS * G;
        auto frag = [&](int u, float4& a, float4& b0, float4& b1) {
            const int wtap = u / G, g = u - wtap * G;
            const int tap = (RES && wtap == TAPS) ? PAD : wtap;
            a = smem4[arow4 + tap * XS4 + g * 4];
            b0 = smem4[brow4 + wtap * 32 * XS4 + g * 4];
            b1 = smem4[brow4 + (wtap * 32 + 16) * XS4 + g * 4];
        };
        float4 a, b0, b1;
        if (wave < U) frag(wave, a, b0, b1);
        for (int u = wave; u < U; u += CC_THREADS / 64) {
            const float4 ca = a, cb0 = b0, cb1 = b1;
            const int un = u + CC_THREADS / 64;
            if (un < U) frag(un, a, b0, b1);
            if (RES && u / G == TAPS) {
                r0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.x, cb0.x, r0, 0, 0, 0);
                r1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.x, cb1.x, r1, 0, 0, 0);
                r0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.y, cb0.y, r0, 0, 0, 0);
                r1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.y, cb1.y, r1, 0, 0, 0);
                r0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.z, cb0.z, r0, 0, 0, 0);
                r1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.z, cb1.z, r1, 0, 0, 0);
                r0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.w, cb0.w, r0, 0, 0, 0);
                r1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.w, cb1.w, r1, 0, 0, 0);
            } else {
                c0a = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.x, cb0.x, c0a, 0, 0, 0);
                c1a = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.x, cb1.x, c1a, 0, 0, 0);
                c0b = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.y, cb0.y, c0b, 0, 0, 0);
                c1b = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.y, cb1.y, c1b, 0, 0, 0);
                c0a = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.z, cb0.z, c0a, 0, 0, 0);
                c1a = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.z, cb1.z, c1a, 0, 0, 0);
                c0b = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.w, cb0.w, c0b, 0, 0, 0);
                c1b = __builtin_amdgcn_mfma_f32_16x16x4f32(ca.w, cb1.w, c1b, 0, 0, 0);
            }
        }
        CC_STAMP(4);
        __syncthreads();
        // D[row = 4 * g4 + r][col = l16 (+ 16 for the second channel half)]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * g4 + r;
            E[(wave * 16 + row) * ES + l16] = c0a[r] + c0b[r];
            E[(wave * 16 + row) * ES + 16 + l16] = c1a[r] + c1b[r];
            if (RES) {
                ER[(wave * 16 + row) * ES + l16] = r0[r];
                ER[(wave * 16 + row) * ES + 16 + l16] = r1[r];
            }
        }
    }
    __syncthreads();
    CC_STAMP(5);
    const int which = tid >> 8;                        // 0: the conv, 1: the riding 1x1 conv
    if (which == 1 && !RES) return;
    const int t8 = tid & 255;
    const int row = t8 >> 3, col = (t8 & 7) * 4;
    if (row >= NR) return;
    const float* q = (which ? ER : E) + row * ES + col;
    float4 v = *reinterpret_cast<const float4*>(q);
#pragma unroll
    for (int w = 1; w < 8; ++w) {
        const float4 u = *reinterpret_cast<const float4*>(q + w * NR * ES);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const int smp = row >> p.lshift, l = (row & (Lout - 1)) + (WIN ? wl0 : 0);
    if (smp >= nvalid) return;
    const int em = m0 + col;
    long off;
    int ocols;
    if (!p.interleave) {
        ocols = M;
        off = (long)((s0 + smp) * Lout + l) * M + em;
    } else {
        const int half = M >> 1, ph = em >= half;
        ocols = half;
        off = (long)((s0 + smp) * (2 * Lout) + 2 * l + ph) * half + (em - ph * half);
    }
    float* out = (which ? p.orslab : p.oslab) + (long)kb * p.out_rows * ocols + off;
    *reinterpret_cast<float4*>(out) = v;
    CC_STAMP(6);
}

// ------------------------------------------------------------------ final conv + posterior, CC form
// One block per sample: finishes final_conv[0]'s output (GroupNorm over whole samples) into LDS,
// then the 1x1 output conv and the posterior update exactly as final_posterior_kernel does.
struct FinalCcParams {
    CcSrc src;               // final_conv[0]'s output, still in pieces
    FinalParams f;           // everything else (f.act unused)
};

__host__ __device__ inline size_t final_cc_lds_floats(int td, int dim, int H) {
    return (size_t)td * dim + ((td + 3) & ~3) + (size_t)H * (dim + 4);
}

__global__ __launch_bounds__(CC_THREADS) void final_cc_kernel(const FinalCcParams pp) {
    extern __shared__ __attribute__((aligned(16))) float ws[];
    const FinalParams& p = pp.f;
    const int td = p.td, dim = p.dim, H = p.H;
    const int rs = dim + 4;
    float* wl = ws;                        // [td][dim]
    float* bl = wl + td * dim;             // [td] (+ pad to 4)
    float* tile = bl + ((td + 3) & ~3);    // [H][dim + 4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int dq = dim >> 2;
    // blockIdx.y owns a contiguous range of output columns (wide transitions at batch 1 would
    // otherwise leave the whole 1x1 conv to one CU)
    const int jlo = (td * (int)blockIdx.y) / (int)gridDim.y, jhi = (td * ((int)blockIdx.y + 1)) / (int)gridDim.y;
    for (int i = tid; i < (jhi - jlo) * dq; i += CC_THREADS)
        *reinterpret_cast<float4*>(wl + i * 4) = ldg4(p.w + (long)jlo * dim + i * 4);
    for (int i = tid; i < td; i += CC_THREADS) bl[i] = p.bias[i];
    cc_build_input<false>(pp.src, tile, rs, b * H, H, H, H, 31 - __clz(H), H, 0, 0, dim, false, tid, lane, wave);
    __syncthreads();
    // neighbouring lanes take neighbouring positions of one column: the tile rows have an odd
    // 16-byte stride (conflict-free) and the weight row is a broadcast (the transposed mapping read
    // td different weight rows, all on one bank)
    for (int o = tid; o < H * (jhi - jlo); o += CC_THREADS) {
        const int jj = o / H, l = o - jj * H, j = jlo + jj;
        const float* wr = wl + jj * dim;
        const float* arow = tile + l * rs;
        const long idx = ((long)b * H + l) * td + j;
        const float xv = p.x[idx];                               // ahead of the dot product: its latency hides there
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;       // four chains, summed pairwise
        for (int c = 0; c < dim; c += 4) {
            const float4 wv = *reinterpret_cast<const float4*>(wr + c);
            const float4 av = *reinterpret_cast<const float4*>(arow + c);
            a0 = fmaf(wv.x, av.x, a0); a1 = fmaf(wv.y, av.y, a1);
            a2 = fmaf(wv.z, av.z, a2); a3 = fmaf(wv.w, av.w, a3);
        }
        const float out = ((a0 + a1) + (a2 + a3)) + bl[j];
        if (p.eps_out != nullptr) p.eps_out[idx] = out;
        if (p.x_out_disabled && p.mean_out == nullptr) continue;
        float x0 = p.predict_epsilon ? p.c_recip * xv - p.c_recipm1 * out : out;
        if (p.clip_denoised) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        float mean = p.coef1 * x0 + p.coef2 * xv;
        if (p.guide != nullptr) mean = mean + p.guide_scale * p.guide[idx];
        if (p.mean_out != nullptr) p.mean_out[idx] = mean;
        if (p.x_out_disabled) continue;
        const float z = (p.noise != nullptr)
                            ? p.noise[idx]
                            : philox_normal(p.elem_offset + (uint64_t)idx, p.draw,
                                            p.seed_dev != nullptr ? (uint64_t)*p.seed_dev : p.seed);
        float xn = mean + p.sigma * z;
        if (l == 0 && p.cond0 != nullptr) xn = p.cond0[(p.cond_per_row ? (long)b * td : 0) + j];
        p.x[idx] = xn;
    }
}

}  // namespace dad
