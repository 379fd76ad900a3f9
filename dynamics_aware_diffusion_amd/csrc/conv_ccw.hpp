// conv_ccw.hpp — the consumer-combine conv (conv_cc.hpp) for WIDE layers at small batch.
//
// A 2048 -> 2048 k=5 layer of the HalfCheetah / Door denoisers holds 84 MB of weights; at batch 1 a
// denoise step streams ~1.2 GB of weights through a few hundred MFLOP: the step is bound by the
// HBM weight stream, and by what sits between the streams.  conv_cc.hpp keeps a block's whole
// weight slice in LDS (slices of <= 64 channels); here a block owns a K slice of up to 512 channels
// (whole GroupNorm groups of the tensor it finishes: 128 / 256 channels per group in these nets) and
//
//   * weights never touch LDS: every wave loads the MFMA B fragments of its (16-channel granule,
//     tap) units global -> registers, DEPTH units ahead, and rolls the ring as units are consumed
//     (the packed image [C_in/16][taps][M][16] makes a wave's fragment load one contiguous 2 KiB);
//   * the input slice is finished by ALL 512 threads, element-wise: partial slabs + bias (+ the
//     additive terms: time embedding, residual, riding 1x1 conv) are loaded FIRST, before the weight
//     ring is filled, so they return ahead of it and the ring streams while the block computes
//     GroupNorm statistics (one wave per (sample, group) pair, from LDS), Mish and the adds;
//   * partial tiles are stored raw, exactly like conv_cc: the next consumer finishes them.
//
// Same arithmetic contract as conv_cc.hpp (fp32 throughout, fixed summation order, reference chain
// F.conv1d -> F.group_norm -> F.mish -> adds, m_diffuser/models/temporal_unet.py:57-122).
#pragma once
#include "conv_cc.hpp"

namespace dad {

constexpr int CCW_DEPTH = 8;          // (granule, tap) units of weights in flight per wave: 16 float4 per lane
constexpr int CCW_MAX_PAIRS = kCcwMaxPairs;

// x / d for a wave-uniform d that is a power of two in every net this path was built for (channel counts
// of 128 .. 2048): a shift then, the ~25-instruction integer division otherwise (one uniform branch).
struct CcwDiv {
    int d, sh;
    __device__ __forceinline__ explicit CcwDiv(int d_) : d(d_), sh((d_ & (d_ - 1)) == 0 ? 31 - __clz(d_) : -1) {}
    __device__ __forceinline__ int operator()(int x) const { return sh >= 0 ? x >> sh : x / d; }
};

// Loads of one float4 of the input slice, issued together and summed later (registers only).
template <bool RIDE>
struct CcwElem {
    float4 part[CC_MAX_SLABS];
    float4 b, tv, rv;
    float4 rp[RIDE ? CC_MAX_SLABS : 1];
    float4 rb;
};

// element i of the slice: row r = i / q4 of the tile, channel quad q = i % q4
template <bool RIDE>
__device__ __forceinline__ void ccw_issue(const CcSrc& s, CcwElem<RIDE>& e, int i, int n4, int q4, const CcwDiv& by_q4,
                                          int r0, int nrows_valid, int c0) {
    const int ii = min(i, n4 - 1);
    const int rr = by_q4(ii);
    const int r = min(rr, nrows_valid - 1), q = ii - rr * q4;
    const int c = c0 + 4 * q;
    const long off = (long)(r0 + r) * s.C + c;
    const long sstride = (long)s.rows * s.C;
    const int nsl = max(s.nsl, 1);
    // slabs that do not exist are neither loaded nor initialised (ccw_reduce skips them under the
    // same wave-uniform condition).  These loads precede the weight ring, so the stricter waits a
    // conditional load implies only cover loads of this same batch.
    e.part[0] = ldg4(s.data + off);
#pragma unroll
    for (int k = 1; k < CC_MAX_SLABS; ++k)
        if (k < nsl) e.part[k] = ldg4(s.data + (long)k * sstride + off);
    const float* some = s.bias != nullptr ? s.bias : s.data;
    e.b = ldg4((s.bias != nullptr ? s.bias : s.data) + (s.bias != nullptr ? c : 0));
    e.tv = ldg4(s.temb != nullptr ? s.temb + c : some);
    e.rv = ldg4(s.res != nullptr ? s.res + off : some);
    if constexpr (RIDE) {
        const int nrs = max(s.nrs, 1);
        const bool hr = s.rslab != nullptr;              // (the other source of a concat may have none)
        e.rp[0] = ldg4(hr ? s.rslab + off : some);
#pragma unroll
        for (int k = 1; k < CC_MAX_SLABS; ++k)
            if (hr && k < nrs) e.rp[k] = ldg4(s.rslab + (long)k * sstride + off);
        e.rb = ldg4(hr ? s.rbias + c : some);
    } else {
        e.rp[0] = make_float4(0.f, 0.f, 0.f, 0.f);
        e.rb = e.rp[0];
    }
}

// v = sum of slabs in slice order + bias;  ex = temb + residual (+ riding conv), as conv_cc.hpp
template <bool RIDE>
__device__ __forceinline__ void ccw_reduce(const CcSrc& s, const CcwElem<RIDE>& e, float4& v, float4& ex) {
    float4 a = e.part[0];
#pragma unroll
    for (int k = 1; k < CC_MAX_SLABS; ++k)
        if (k < s.nsl) { a.x += e.part[k].x; a.y += e.part[k].y; a.z += e.part[k].z; a.w += e.part[k].w; }
    // absent operands were loaded from some valid address: cancelled with selects (v_cndmask), not
    // multiplied by 0 — 0 * Inf would carry a non-finite word of an unrelated tensor into the sums
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bb = s.bias != nullptr ? e.b : zero4;
    v.x = bb.x + a.x; v.y = bb.y + a.y; v.z = bb.z + a.z; v.w = bb.w + a.w;
    const float4 tv = s.temb != nullptr ? e.tv : zero4, rv = s.res != nullptr ? e.rv : zero4;
    float4 r = zero4;
    bool has_d = false;
    if constexpr (RIDE) {
        has_d = s.rslab != nullptr;
        r = e.rp[0];
#pragma unroll
        for (int k = 1; k < CC_MAX_SLABS; ++k)
            if (s.rslab != nullptr && k < s.nrs) { r.x += e.rp[k].x; r.y += e.rp[k].y; r.z += e.rp[k].z; r.w += e.rp[k].w; }
    }
    const float4 d = has_d ? make_float4(r.x + e.rb.x, r.y + e.rb.y, r.z + e.rb.z, r.w + e.rb.w) : zero4;
    ex.x = (tv.x + rv.x) + d.x;
    ex.y = (tv.y + rv.y) + d.y;
    ex.z = (tv.z + rv.z) + d.z;
    ex.w = (tv.w + rv.w) + d.w;
}

// grid = (K slices, M / 32, N tiles of NR rows), 8 waves.  Host contract (cc_plan): weight image in
// 16-channel granules, slice a multiple of 32 and of the finished tensor's group width, at most
// CC_MAX_SLABS partial slabs per input, channel counts multiples of 4, pairs of a block <= CCW_MAX_PAIRS
// with at most 8192 elements each (up to 2048 stay in registers between the two passes).
template <int TAPS, int STRIDE, bool RES, bool RIDE, int NR>
__global__ __launch_bounds__(CC_THREADS) void conv_ccw(const CcParams p) {
    static_assert(NR == 16 || NR == 32, "tile rows");
    constexpr int PAD = TAPS / 2;
    constexpr int WTAPS = TAPS + (RES ? 1 : 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* const smem4 = reinterpret_cast<float4*>(smem);
    CC_PIN_SRC(p.src0);
    CC_PIN_SRC(p.src1);
    asm volatile("" ::"s"(p.w), "s"(p.wtaps), "s"(p.cin0), "s"(p.cin1), "s"(p.M), "s"(p.B), "s"(p.Lin), "s"(p.Lout),
                 "s"(p.lshift), "s"(p.lshift_in), "s"(p.interleave), "s"(p.slice_ch), "s"(p.oslab), "s"(p.orslab),
                 "s"(p.out_rows));
    CC_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = blockIdx.x, mt = blockIdx.y, nt = blockIdx.z;
    const int Lin = p.Lin, Lout = p.Lout, M = p.M;
    const int SPT = NR >> p.lshift;
    const int SEG = Lin + 2 * PAD;
    const int XROWS = SPT * SEG;
    const int s0 = nt * SPT;
    const int nvalid = min(SPT, p.B - s0);
    const int m0 = mt * 32;
    const int SL = p.slice_ch;
    const int XS = SL + 4, XS4 = XS >> 2;
    float* const Xb = smem;                            // [XROWS][XS]
    float* const EXb = Xb + XROWS * XS;                // [SPT * Lin][XS]   additive terms
    float* const GBb = EXb + SPT * Lin * XS;           // [2][SL]           gamma, beta
    float* const STb = GBb + 2 * SL;                   // [pairs][2]        mean, rstd

    const int c0 = kb * SL;
    const bool second = p.cin1 > 0 && c0 >= p.cin0;
    // field-by-field scalar selects: a reference picked at run time makes hipcc park both structs in
    // scratch and index them
    CcSrc src;
#define CCW_PICK(f) src.f = second ? p.src1.f : p.src0.f
    CCW_PICK(data); CCW_PICK(nsl); CCW_PICK(C); CCW_PICK(rows); CCW_PICK(cpg); CCW_PICK(bias); CCW_PICK(gamma);
    CCW_PICK(beta); CCW_PICK(temb); CCW_PICK(res); CCW_PICK(rslab); CCW_PICK(rbias); CCW_PICK(nrs); CCW_PICK(mat);
#undef CCW_PICK
    src.pad_ = 0;
    const int cs0 = second ? c0 - p.cin0 : c0;
    const int cin_src = second ? p.cin1 : p.cin0;
    const int nch = min(SL, cin_src - cs0);            // channels to stage (multiple of 4; rest of the slice: zero)
    const int q4 = nch >> 2;
    const int rows_tile = SPT * Lin, rows_valid = nvalid * Lin, r0 = s0 * Lin;
    const int n4 = rows_valid * q4;                    // float4 elements to finish
    const CcwDiv by_q4(q4);
    const bool gn = src.gamma != nullptr;
    const bool publish = mt == 0 && src.nsl > 0 && src.mat != nullptr;

    // ---- 1. the first element of every thread: loads issued before anything else ---------------
    CcwElem<RIDE> e0;
    ccw_issue<RIDE>(src, e0, tid, n4, q4, by_q4, r0, rows_valid, cs0);
    float4 gam0 = make_float4(0.f, 0.f, 0.f, 0.f), bet0 = gam0;
    {
        const int cq = min(tid, q4 - 1) * 4;
        const float* gp = gn ? src.gamma : src.data;
        const float* bp = gn ? src.beta : src.data;
        gam0 = ldg4(gp + (gn ? cs0 + cq : 0));
        bet0 = ldg4(bp + (gn ? cs0 + cq : 0));
    }

    // ---- 2. fill the weight ring ----------------------------------------------------------------
    const int U = (SL >> 4) * WTAPS;                   // (16-channel granule, tap) units of the slice
    const int nU = U > wave ? (U - wave + 7) >> 3 : 0; // units of this wave: wave, wave + 8, ...
    const int klast = max(nU - 1, 0);
    const int l32 = lane & 31, h = lane >> 5, l16 = lane & 15, g4 = lane >> 4;
    const int lane_w = NR == 32 ? (m0 + l32) * 16 + 4 * h : (m0 + l16) * 16 + 4 * g4;
    const float* const wlane = p.w + lane_w;
    const long g0 = c0 >> 4;
    auto wload = [&](int k, float4& x0, float4& x1) {
        const int u = min(wave + 8 * min(k, klast), U - 1);
        const int gr = u / WTAPS, tap = u - gr * WTAPS;
        const float* b = wlane + ((g0 + gr) * p.wtaps + tap) * (long)M * 16;
        x0 = ldg4(b);
        x1 = ldg4(b + (NR == 32 ? 8 : 16 * 16));       // second 8-channel group / second 16-row channel half
    };
    // (unconditional: slots past the wave's last unit re-read it.  Loads under `if (i < nU)` make
    // every later wait assume that none of them was issued — the first use of the input's loads
    // then drains the whole ring)
    float4 wq0[CCW_DEPTH], wq1[CCW_DEPTH];
#pragma unroll
    for (int i = 0; i < CCW_DEPTH; ++i) wload(i, wq0[i], wq1[i]);
    CC_STAMP(1);

    // ---- 3. finish the input slice into LDS ------------------------------------------------------
    // zero: halo rows, absent samples, columns past the staged channels
    if (nch < SL || rows_valid < rows_tile) {
        for (int i = tid; i < XROWS * XS4; i += CC_THREADS) smem4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    } else if (PAD > 0) {
        const int sq4 = SL >> 2;
        for (int i = tid; i < SPT * 2 * PAD * sq4; i += CC_THREADS) {
            const int hr = i / sq4, q = i - hr * sq4;
            const int smp = hr / (2 * PAD), j = hr - smp * (2 * PAD);
            const int row = smp * SEG + (j < PAD ? j : Lin + j);
            *reinterpret_cast<float4*>(Xb + row * XS + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (gn && tid < q4) {
        *reinterpret_cast<float4*>(GBb + 4 * tid) = gam0;
        *reinterpret_cast<float4*>(GBb + SL + 4 * tid) = bet0;
    }
    const int lshL = p.lshift_in;
    auto finish_first_pass = [&](int i, const CcwElem<RIDE>& e) {
        if (i >= n4) return;
        float4 v, ex;
        ccw_reduce<RIDE>(src, e, v, ex);
        const int r = by_q4(i), q = i - r * q4;
        const int smp = r >> lshL, l = r & (Lin - 1);
        float* xp = Xb + (smp * SEG + PAD + l) * XS + 4 * q;
        if (gn) {
            *reinterpret_cast<float4*>(xp) = v;
            *reinterpret_cast<float4*>(EXb + r * XS + 4 * q) = ex;
        } else {
            v.x += ex.x; v.y += ex.y; v.z += ex.z; v.w += ex.w;       // (no norm: bias-only tensors; ex is 0)
            *reinterpret_cast<float4*>(xp) = v;
            if (publish) store_f4_sc1(src.mat + (long)(r0 + r) * src.C + cs0 + 4 * q, v);
        }
    };
    finish_first_pass(tid, e0);
    for (int i = tid + CC_THREADS; i < n4; i += CC_THREADS) {          // further elements: second round trip
        CcwElem<RIDE> e;
        ccw_issue<RIDE>(src, e, i, n4, q4, by_q4, r0, rows_valid, cs0);
        finish_first_pass(i, e);
    }
    if (gn && q4 > CC_THREADS)
        for (int t = tid + CC_THREADS; t < q4; t += CC_THREADS) {
            *reinterpret_cast<float4*>(GBb + 4 * t) = ldg4(src.gamma + cs0 + 4 * t);
            *reinterpret_cast<float4*>(GBb + SL + 4 * t) = ldg4(src.beta + cs0 + 4 * t);
        }
    __syncthreads();
    CC_STAMP(2);
    if (gn) {
        // statistics of every (sample, group) pair of the slice: one wave per pair, from LDS
        const int cpg = src.cpg, groups = nch / cpg, cqp = cpg >> 2;
        const CcwDiv by_cqp(cqp);
        const int cnt4 = Lin * cqp;
        const float inv_cnt = 1.0f / (float)(Lin * cpg);
        for (int pr = wave; pr < nvalid * groups; pr += CC_THREADS / 64) {
            const int smp = pr / groups, g = pr - smp * groups;
            if (cnt4 > 8 * 64) {
                // pairs beyond 2048 elements (e.g. 2048 channels at 16 positions): same two passes,
                // re-reading LDS instead of keeping the pair in registers
                const float* base = Xb + (smp * SEG + PAD) * XS + g * cpg;
                float sum = 0.0f;
                for (int j = lane; j < cnt4; j += 64) {
                    const int l = by_cqp(j), cl = (j - l * cqp) * 4;
                    const float4 t = *reinterpret_cast<const float4*>(base + l * XS + cl);
                    sum += (t.x + t.y) + (t.z + t.w);
                }
                const float mean = wave_sum(sum) * inv_cnt;
                float sq = 0.0f;
                for (int j = lane; j < cnt4; j += 64) {
                    const int l = by_cqp(j), cl = (j - l * cqp) * 4;
                    const float4 t = *reinterpret_cast<const float4*>(base + l * XS + cl);
                    const float dx = t.x - mean, dy = t.y - mean, dz = t.z - mean, dw = t.w - mean;
                    sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
                }
                const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_cnt + 1e-5f);
                if (lane == 0) { STb[2 * pr] = mean; STb[2 * pr + 1] = rstd; }
                continue;
            }
            float4 vv[8];
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = lane + 64 * k;
                const bool on = j < cnt4;
                const int jj = on ? j : 0;
                const int l = by_cqp(jj), cl = (jj - l * cqp) * 4;
                const float4 t = *reinterpret_cast<const float4*>(Xb + (smp * SEG + PAD + l) * XS + g * cpg + cl);
                vv[k] = on ? t : make_float4(0.f, 0.f, 0.f, 0.f);
                sum += (vv[k].x + vv[k].y) + (vv[k].z + vv[k].w);
            }
            const float mean = wave_sum(sum) * inv_cnt;
            float sq = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float mk = lane + 64 * k < cnt4 ? 1.0f : 0.0f;
                const float dx = vv[k].x - mean, dy = vv[k].y - mean, dz = vv[k].z - mean, dw = vv[k].w - mean;
                sq += mk * ((dx * dx + dy * dy) + (dz * dz + dw * dw));
            }
            const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_cnt + 1e-5f);
            if (lane == 0) { STb[2 * pr] = mean; STb[2 * pr + 1] = rstd; }
        }
        __syncthreads();
        CC_STAMP(7);
        for (int i = tid; i < n4; i += CC_THREADS) {
            const int r = by_q4(i), q = i - r * q4;
            const int smp = r >> lshL, l = r & (Lin - 1);
            const int pr = smp * groups + by_cqp(q);
            const float mean = STb[2 * pr], rstd = STb[2 * pr + 1];
            float* xp = Xb + (smp * SEG + PAD + l) * XS + 4 * q;
            const float4 v = *reinterpret_cast<const float4*>(xp);
            const float4 ex = *reinterpret_cast<const float4*>(EXb + r * XS + 4 * q);
            const float4 gam = *reinterpret_cast<const float4*>(GBb + 4 * q);
            const float4 bet = *reinterpret_cast<const float4*>(GBb + SL + 4 * q);
            float4 y;
            y.x = mish_fast_f32((v.x - mean) * rstd * gam.x + bet.x) + ex.x;
            y.y = mish_fast_f32((v.y - mean) * rstd * gam.y + bet.y) + ex.y;
            y.z = mish_fast_f32((v.z - mean) * rstd * gam.z + bet.z) + ex.z;
            y.w = mish_fast_f32((v.w - mean) * rstd * gam.w + bet.w) + ex.w;
            *reinterpret_cast<float4*>(xp) = y;
            if (publish) store_f4_sc1(src.mat + (long)(r0 + r) * src.C + cs0 + 4 * q, y);
        }
        __syncthreads();
    }

    CC_STAMP(3);
    // ---- 4. K loop: A fragments from LDS, B fragments from the rolling register ring ------------
    const int phase_shift = (TAPS == 2 && p.interleave && m0 >= (M >> 1)) ? 1 : 0;
    constexpr int ES = 36;
    float* const E = smem;                             // [8][NR][ES]
    float* const ER = smem + 8 * NR * ES;
    const int rowl = NR == 32 ? l32 : l16;
    const int arow4 = (((rowl >> p.lshift) * SEG + (rowl & (Lout - 1)) * STRIDE + phase_shift) * XS + 4 * (NR == 32 ? h : g4)) >> 2;
    f32x16 acc, acc2, accr;                            // NR == 32
    f32x4 c0a = {0.f, 0.f, 0.f, 0.f}, c0b = c0a, c1a = c0a, c1b = c0a, rr0 = c0a, rr1 = c0a;   // NR == 16
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; accr[r] = 0.f; }
    auto unit = [&](int k, const float4& b0, const float4& b1) {
        const int u = wave + 8 * k;
        const int gr = u / WTAPS, wtap = u - gr * WTAPS;
        const bool ride_unit = RES && wtap == TAPS;
        const int tap = ride_unit ? PAD : wtap;
        if constexpr (NR == 32) {
            const float4 a0 = smem4[arow4 + tap * XS4 + gr * 4];
            const float4 a1 = smem4[arow4 + tap * XS4 + gr * 4 + 2];
            if (ride_unit) {
                accr = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, accr, 0, 0, 0);
                accr = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, accr, 0, 0, 0);
                accr = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, accr, 0, 0, 0);
                accr = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, accr, 0, 0, 0);
                accr = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, accr, 0, 0, 0);
                accr = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, accr, 0, 0, 0);
                accr = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, accr, 0, 0, 0);
                accr = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, accr, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc2, 0, 0, 0);
            }
        } else {
            // lane = (row or channel l16, k group g4): the float4 at channels 4*g4.. of the granule
            // is component j of MFMA j's k index g4 — the same bijection of K for both operands
            const float4 a = smem4[arow4 + tap * XS4 + gr * 4];
            if (ride_unit) {
                rr0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, rr0, 0, 0, 0);
                rr1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, rr1, 0, 0, 0);
                rr0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, rr0, 0, 0, 0);
                rr1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, rr1, 0, 0, 0);
                rr0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, rr0, 0, 0, 0);
                rr1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, rr1, 0, 0, 0);
                rr0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, rr0, 0, 0, 0);
                rr1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, rr1, 0, 0, 0);
            } else {
                c0a = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, c0a, 0, 0, 0);
                c1a = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, c1a, 0, 0, 0);
                c0b = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, c0b, 0, 0, 0);
                c1b = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, c1b, 0, 0, 0);
                c0a = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, c0a, 0, 0, 0);
                c1a = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, c1a, 0, 0, 0);
                c0b = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, c0b, 0, 0, 0);
                c1b = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, c1b, 0, 0, 0);
            }
        }
    };
    // Full rounds of the ring form a single basic block (a conditional inside the loop made hipcc
    // start every round with s_waitcnt vmcnt(0), emptying the ring); the remaining units follow
    // without refills.
    int base = 0;
    for (; base + CCW_DEPTH <= nU; base += CCW_DEPTH) {
#pragma unroll
        for (int i = 0; i < CCW_DEPTH; ++i) {
            // the unit's MFMAs first, then the ring slot is refilled in place
            unit(base + i, wq0[i], wq1[i]);
            wload(base + i + CCW_DEPTH, wq0[i], wq1[i]);
            __builtin_amdgcn_sched_barrier(0);          // (else the scheduler sinks all refills to the end of the round)
        }
    }
#pragma unroll
    for (int i = 0; i < CCW_DEPTH; ++i)
        if (base + i < nU) unit(base + i, wq0[i], wq1[i]);
    CC_STAMP(4);
    __syncthreads();                                   // all fragment reads done: LDS becomes the exchange tile
    if constexpr (NR == 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            E[(wave * 32 + row) * ES + l32] = acc[r] + acc2[r];
            if (RES) ER[(wave * 32 + row) * ES + l32] = accr[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * g4 + r;
            E[(wave * 16 + row) * ES + l16] = c0a[r] + c0b[r];
            E[(wave * 16 + row) * ES + 16 + l16] = c1a[r] + c1b[r];
            if (RES) {
                ER[(wave * 16 + row) * ES + l16] = rr0[r];
                ER[(wave * 16 + row) * ES + 16 + l16] = rr1[r];
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
    const int smp = row >> p.lshift, l = row & (Lout - 1);
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
    // write-through, like the batch-256 epilogue: nothing dirty is left for the end-of-kernel release
    // (-1.3 % per step on HalfCheetah at batch 1; the narrow conv_cc kernels measured +1.2 % with it and
    // keep ordinary stores)
    store_f4_sc1(out, v);
    CC_STAMP(6);
}

}  // namespace dad
