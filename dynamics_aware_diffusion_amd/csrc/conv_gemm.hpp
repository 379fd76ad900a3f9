// conv_gemm.hpp — Conv1d / ConvTranspose1d of the TemporalUnet as an implicit GEMM on the
// gfx950 matrix cores — exact fp32 products (v_mfma_f32_32x32x2_f32) or, template parameter X3,
// split-f16 operands (three v_mfma_f32_32x32x16_f16 per product block, fp32 accumulation) —
// with the whole Conv1dBlock tail (bias -> GroupNorm(8) -> Mish -> + time embedding ->
// + residual) fused in the epilogue.
//
// Replaces, per launch, the reference's  F.conv1d / F.conv_transpose1d -> F.group_norm ->
// F.mish -> add chains of m_diffuser/models/temporal_unet.py:57-76 (Conv1dBlock),
// :106-122 (ResidualTemporalBlock), :35-54 (Down/Upsample1d).
//
// GEMM view:  Y[N = B*L_out][M = C_out] = Xcol[N][K = taps*C_in] * W^T[K][M]
//   * activations are channels-last in HBM:  act[b*L + l][c]  (the reference's external
//     (batch, horizon, transition_dim) layout is the same thing with c = transition_dim), so a
//     tile of BN rows is BN/L_out whole samples and a GroupNorm group (C_out/8 channels x
//     L_out) never straddles a tile when BM is a multiple of C_out/8.
//   * im2col is never materialised: a K-chunk of KC input channels is staged in LDS as one
//     row per (sample, position) with PAD zero rows on both sides of every sample; tap j of
//     the filter is then "+ j rows" on the LDS address.
//   * weights are pre-packed on the host as [C_in/16][tap][M][16] so that both MFMA operands
//     are fetched with one ds_read_b128 per four MFMAs: lane half h of the wave owns input
//     channels 4h..4h+3 of every 8-channel group (any bijection of K onto (step, half) is a
//     valid summation order as long as both operands use it).
//   * SK waves share one 32x32 output tile and split the K units between them (intra-block
//     split-K): at batch 256 a layer has only ~1 output tile per SIMD, the second wave per
//     SIMD is what hides LDS/barrier latency.  Partial sums meet in the LDS epilogue tile.
//   * a transposed conv (k=4, s=2, p=1) runs as a 2-tap conv with M = 2*C_out columns: even
//     outputs y[2j] = W3 x[j-1] + W1 x[j] (columns [0, C_out)), odd outputs
//     y[2j+1] = W2 x[j] + W0 x[j+1] (columns [C_out, 2 C_out)); the tile's phase shifts the LDS
//     row base by one, the store interleaves the two phases.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_shapes.hpp"

namespace dad {

#define DAD_LBID (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z))
#ifdef DAD_STAMPS
#define DAD_STAMP(i) do { if (p.stamps != nullptr && threadIdx.x == 0 && DAD_LBID < 4096) p.stamps[DAD_LBID * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DAD_STAMP(i) do {} while (0)
#endif
#ifdef DAD_STAMPS
#define DAD_CLOCK(i) do { if (p.stamps != nullptr && threadIdx.x == 0 && DAD_LBID < 4096) p.stamps[DAD_LBID * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DAD_CLOCK(i) do {} while (0)
#endif
#ifdef DAD_STAMPS_PROLOGUE   /* variant: the six stamps resolve the prologue instead */
#define DAD_PSTAMP(i) do { if (p.stamps != nullptr && threadIdx.x == 0 && DAD_LBID < 4096) p.stamps[DAD_LBID * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#undef DAD_STAMP
#define DAD_STAMP(i) do { if ((i) == 0) DAD_PSTAMP(0); } while (0)
#undef DAD_CLOCK
#define DAD_CLOCK(i) do {} while (0)
#else
#define DAD_PSTAMP(i) do {} while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---- split-f16 operands (template parameter X3) ---------------------------------------------
// An fp32 value v is carried as two halves  v ~= hi + lo * 2^-11,  hi = f16(v),
// lo = f16((v - hi) * 2^11)  (22 significant bits; the residual is scaled so it stays a normal
// f16).  A product is then three v_mfma_f32_32x32x16_f16 with fp32 accumulation:
//     x*w ~= xh*wh  +  2^-11 (xh*wl + xl*wh)          (the dropped xl*wl term is 2^-22 relative)
// — 3/16 of the matrix-pipe cycles of the exact-fp32 MFMA at the same LDS/HBM bytes (2+2 B per
// element).  Weights are split on the host at pack time (pre-scaled by a power of two per layer
// so that small weights stay normal halves), activations while they are staged into LDS.
// Measured against the fp64 run of the reference this path is as close as the reference's own
// fp32 arithmetic (tests/test_hip_split.py); value range: |activation| <= 65504 (saturates).
__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
    const float c = __builtin_fminf(__builtin_fmaxf(v, -65504.0f), 65504.0f);
    hi = (_Float16)c;
    lo = (_Float16)((c - (float)hi) * 2048.0f);
}
__device__ __forceinline__ void split_f16x4(const float4 v, float2& hi, float2& lo) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    f16x4 h, l;
    _Float16 a, b;
    split_f16(v.x, a, b); h[0] = a; l[0] = b;
    split_f16(v.y, a, b); h[1] = a; l[1] = b;
    split_f16(v.z, a, b); h[2] = a; l[2] = b;
    split_f16(v.w, a, b); h[3] = a; l[3] = b;
    hi = __builtin_bit_cast(float2, h);
    lo = __builtin_bit_cast(float2, l);
}

struct ConvParams {
    const float* src0;   // [B*Lin][cin0]
    const float* src1;   // [B*Lin][cin1]  second half of a virtual channel concat, or nullptr
    const float* w;      // packed [cin_pad/KG][TAPS][M][KG], KG = min(KC, 16)
    const float* bias;   // [M]
    const float* gamma;  // [M] GroupNorm weight, or nullptr (no norm, no Mish)
    const float* beta;   // [M]
    const float* temb;   // [M] time-embedding projection for this t, or nullptr
    const int32_t* trow; // per-row timesteps (device, [B]) or nullptr: row b reads temb + trow[b] * temb_stride
    int32_t temb_stride; //   (training-side forward: diffusion.py:253-290 draws one t per trajectory)
    const float* res;    // [B*Lout][M] residual to add after Mish, or nullptr
    float* dst;          // [B*Lout][M]; interleave mode: [B*2*Lout][M/2]
    int32_t cin0, cin1;  // channels taken from src0 / src1
    int32_t cin_pad;     // (cin0+cin1) rounded up to KC
    int32_t M;           // GEMM columns (2*C_out for the transposed conv)
    int32_t cpg;         // channels per GroupNorm group (M/8) when gamma != nullptr
    int32_t lreal;       // > 0: only the first lreal GEMM rows of every sample exist (dad_model_set_horizon): the rest are
                         // padding — left out of the statistics and stored as zeros (the next conv's halo); 0: all
    int32_t src_len;     // > 0: src0 is the external trajectory with src_len rows per sample (its real horizon)
    int32_t cpg_real;    // > 0: only the first cpg_real channels of every group exist (dad_model_set_group_channels:
                         // the rest are zero padding and do not count in the statistics); 0: all of them
    int32_t B;           // batch rows in this call
    int32_t Lin, Lout;   // per-sample input / output length of the GEMM
    int32_t lshift;      // log2(Lout)
    int32_t lshift_in;   // log2(Lin)
    int32_t interleave;  // transposed-conv store: col m<M/2 -> row 2l, m>=M/2 -> row 2l+1
    int32_t ntiles_n;    // number of N tiles
    // XCD-aware tile order (xcd_gn > 0): the launch is (K slices, MT*NT tiles) and the hardware deals
    // consecutive workgroups to the 8 XCDs round-robin.  The XCDs form a gm x gn grid (gm*gn = 8);
    // XCD (i, j) owns M tiles [i*MT/gm, ..) x N tiles [j*NT/gn, ..), so a weight slab is fetched by gn
    // XCDs and an activation slab by gm of them instead of 1 and 8 (or 8 and 1).
    int32_t xcd_gn;      // 0: plain 3-D grid (y = M tile, z = N tile)
    int32_t xcd_mts;     // log2(MT / gm): M tiles per XCD
    int32_t xcd_ntn;     // NT / gn: N tiles per XCD
    // grid-level split-K (few tiles: small batches, the deepest levels of wide nets)
    int32_t kslices;     // blocks per output tile (1 = off)
    int32_t chunks_per_slice;
    float c1, c2;        // split-f16 kernels: out = acc_hh * c1 + acc_cross * c2  (2^-s, 2^-s-11)
    uint64_t xswz;       // 16 x 4 bits: sample s of the tile sits xswz[s] 16-byte slots further right
                         // in the X stage (bank-conflict-free ds_read_b128 at L <= 16; 0 = plain)
    // RES kernels: the block's 1x1 residual conv (temporal_unet.py:117-121) rides along as tap
    // index TAPS of the weight image — same staged rows (centre tap), own accumulators, bias only.
    int32_t wtaps;       // tap slots per (chunk, granule) of the weight image (TAPS, or TAPS + 1)
    const float* rbias;  // [M] residual-conv bias
    float* rdst;         // [B*Lout][M] residual-conv output
    // training forward (dad_unet_forward_train): what the backward pass needs of a GroupNorm'd conv
    float* pre;          // [B*Lout][M] conv + bias BEFORE the normalisation, or nullptr
    float* stats;        // [B][M / cpg][2] (mean, rstd) of every (sample, group) pair, or nullptr
    float* slab;         // [tiles][kslices][BN*BM] fp32 partial tiles (workspace)
    unsigned* counters;  // [tiles] arrival tickets, zero between launches
#ifdef DAD_STAMPS
    unsigned long long* stamps;   // diagnostic build only: [4096][8] realtime stamps per block
#endif
};

__device__ __forceinline__ float mish_f32(float y) {
    // x * tanh(softplus(x)), softplus threshold 20 (torch.nn.functional.mish semantics).
    // tanh(log(1+e^x)) = ((1+e^x)^2 - 1) / ((1+e^x)^2 + 1) = w / (w + 2), w = e^x (e^x + 2)
    if (y > 20.0f) return y;
    const float n = expf(y);
    const float w = n * (n + 2.0f);
    return y * (w / (w + 2.0f));
}

// Same function on the hardware transcendental units (v_exp_f32, v_rcp_f32: ~1 ulp each).
// e^y = 2^(y*log2e); the product is split hi/lo so the argument reduction keeps full fp32
// accuracy for |y| up to the softplus threshold.  Used in the conv epilogue (hot); the table
// builders keep the libm version above.
__device__ __forceinline__ float mish_fast_f32(float y) {
    if (y > 20.0f) return y;
    const float L2E_HI = 1.44269502162933349609375f;        // fp32(log2 e)
    const float L2E_LO = 1.92596299112661746e-08f;          // log2 e - L2E_HI
    const float t = y * L2E_HI;
    const float tl = fmaf(y, L2E_HI, -t) + y * L2E_LO;      // rounding error of t + low part
    const float n = __builtin_amdgcn_exp2f(t) * (1.0f + 0.693147180559945f * tl);
    const float w = n * (n + 2.0f);
    return y * (w * __builtin_amdgcn_rcpf(w + 2.0f));
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL,
                                                                 0xF, 0xF, true));
}
// After the in-row reduction every 16-lane row holds one value; rows are combined through
// v_readlane (scalar broadcast) — four reads, no LDS, fixed order.
__device__ __forceinline__ float rows_sum(float v, int width, int lane) {
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    const float lo = r0 + r1, hi = r2 + r3;
    if (width >= 64) return lo + hi;
    return lane < 32 ? lo : hi;
}

// 16-byte load through an explicitly GLOBAL pointer.  A `cond ? *p : zero` on a generic pointer
// makes hipcc select between addresses and emit flat_load: flat loads count on lgkmcnt as well
// as vmcnt, so the next LDS/scalar wait drains them (measured: +2 us in every kernel prologue).
__device__ __forceinline__ float4 ldg4(const float* p) {
    typedef float __attribute__((ext_vector_type(4))) f4;
    const f4 v = *reinterpret_cast<const __attribute__((address_space(1))) f4*>(
        (const __attribute__((address_space(1))) float*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}

// 16-byte WRITE-THROUGH store (sc1): the bytes leave the XCD's L2 with the store itself, so a
// hand-off to a workgroup on another XCD needs no release fence (buffer_wbl2) afterwards — only
// every storing wave's s_waitcnt vmcnt(0) in front of the signal (MI355X_MICROARCH.md, visibility:
// "publish-large": 8.2 -> 3.0 us for 64 KB per workgroup).  An asm store is invisible to hipcc's
// vmcnt bookkeeping: the explicit s_waitcnt after the stores is part of the recipe.
__device__ __forceinline__ void store_f4_sc1(float* p, const float4 v) {
    typedef float __attribute__((ext_vector_type(4))) f4;
    const f4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
}

// Activation rows in LDS and bank conflicts.  A wave's ds_read_b128 is served in four groups of
// 16 lanes; a group is conflict-free when its lanes hit 16 different 16-byte slots of the 256-byte
// bank row.  Lane n reads row  s*SEG + l  (sample s, position l), rows are an odd number of slots
// apart, so 16 consecutive rows are fine — but samples of L <= 16 positions are separated by their
// halo rows, and rows of different samples alias (measured: 35 % of all LDS cycles were 2-way
// conflicts).  Cure: sample s is stored shifted right by d(s) slots, d in [0, 16), chosen on the
// host so that every group sees 16 distinct slots (ConvParams::xswz; dad_lib.hip find_xswz).  A
// shift moves a sample's trailing halo over the next sample's leading halo — zeros over zeros;
// the search keeps it off the neighbour's real rows.  The stage grows by 15 slots.
// BDIR: the B operand (weights) goes global -> registers directly, no LDS.  For
// tiles one wave-tile wide (BN = 32) every weight fragment is used by exactly one wave, so staging
// it through LDS buys no sharing and only costs capacity: the 256-row tile of the 2048-channel
// layers (GroupNorm groups of 256 channels) cannot double-buffer a 16-channel split-f16 stage in
// 160 KiB, and its fp32 stage is limited to 8 channels (a barrier every 20 MFMAs).  The fragments
// roll like the staged items: those of unit u are consumed by the unit's MFMAs and the registers
// immediately receive the same unit of the next chunk.  Weights are packed in 16-channel granules
// for both arithmetics: [granule][tap][M][16 floats] (split-f16: 16 hi halves | 16 lo halves).
// PADDED: the zero-padded forms (dad_model_set_horizon / dad_model_set_group_channels) as their own instantiations —
// as runtime branches of the one kernel they cost the BASELINE configurations 0.4-0.8 % (A/B of two builds on one
// box): the unpadded kernels are instruction for instruction what they were.
template <int BM, int BN, int SK, int KC, int TAPS, int STRIDE, bool RAGGED, bool X3 = false, bool BDIR = false,
          bool RES = false, bool PADDED = false>
__global__ __launch_bounds__(64 * (BM / 32) * (BN / 32) * SK) void conv_gemm_f32(const ConvParams p) {
    constexpr int TMW = BM / 32;                 // wave tiles along M
    constexpr int TNW = BN / 32;                 // wave tiles along N
    constexpr int WT = TMW * TNW;
    constexpr int NT = 64 * WT * SK;
    constexpr int PAD = TAPS / 2;
    constexpr int WTAPS = TAPS + (RES ? 1 : 0);  // weight taps staged per chunk (RES: + the 1x1 ride)
    constexpr int KP = KC + 4;                   // LDS row stride (floats): 16-B aligned, odd in
                                                 // 16-B units -> conflict-free ds_read_b128
    constexpr int KU = X3 ? 16 : 8;              // channels per unit: 4 fp32 MFMAs (k=2 each) or
                                                 // 3 split-f16 MFMAs (k=16); a unit's operands
                                                 // take KU floats of an LDS row in either format
    constexpr int G = KC / KU;                   // units per tap per chunk
    constexpr int GW = G / SK;                   // units each split-K wave owns, per tap
    constexpr int KG = KC < 16 ? KC : 16;        // packing granule of the weights
    constexpr int NSUB = KC / KG;                // packed granules per chunk
    static_assert(KC % KU == 0 && G % SK == 0 && GW >= 1, "K chunk must split evenly over the SK waves");
    static_assert(!BDIR || (BN == 32 && SK == 1 && !RAGGED && KC >= 16), "direct-B tiles");
    static_assert(!RES || (!X3 && !BDIR && (TAPS & 1) == 1 && TAPS >= 3 && STRIDE == 1), "the residual ride exists for fp32 stride-1 convs of the net's kernel size");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    // Vector accesses go through these views with an index in vector units: every offset below
    // is a multiple of 4 floats by construction, but only the index form lets hipcc see it (one
    // runtime term of unknown low bits and it splits each ds_read_b128 into dword pairs — which
    // doubled the split-f16 kernels' time when the per-sample slot shift was added).
    float4* const smem4 = reinterpret_cast<float4*>(smem);
    float2* const smem2 = reinterpret_cast<float2*>(smem);
#ifdef DAD_ABLATE_NULL
    if (p.B > 0) return;
#endif

    DAD_STAMP(0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int wt = wave % WT;
    const int ks = wave / WT;                    // split-K slice of this wave
    const int tn = wt / TMW;
    const int tm = wt % TMW;
    const int l32 = lane & 31;
    const int h = lane >> 5;

    // Grid = (K slices, M tiles, N tiles), or (K slices, tiles) with the XCD-aware order below.
    // Hardware deals linear block ids (x fastest) round-robin over the 8 XCDs; which tiles meet
    // in one XCD's L2 decides how often the weight and activation slabs are fetched from memory
    // (speed / traffic only; any placement is correct).
    const int kb = blockIdx.x;                   // this block's K slice of its output tile
    int mt = blockIdx.y;
    int nt = blockIdx.z;
    if (p.xcd_gn > 0) {                          // scalar arithmetic only (blockIdx is uniform)
        const int wg = blockIdx.y, c = wg & 7, j = wg >> 3;
        const int im = c / p.xcd_gn, in = c - im * p.xcd_gn;
        mt = (im << p.xcd_mts) + (j & ((1 << p.xcd_mts) - 1));
        nt = in * p.xcd_ntn + (j >> p.xcd_mts);
    }
    const int tile = mt * p.ntiles_n + nt;       // id for the split-K slab / ticket

    const int Lin = p.Lin, Lout = p.Lout;
    const int SPT = BN >> p.lshift;              // whole samples per tile
    const int SEG = Lin + 2 * PAD;
    const int XROWS = SPT * SEG;
    const int s0 = nt * SPT;                     // first sample of this tile
    const int m0 = mt * BM;                      // first output channel of this tile
    const int M = p.M;
    const int nvalid = min(SPT, p.B - s0);

    const int XF = XROWS * KP + kXSwzPad;
    const int STAGE = XF + (BDIR ? 0 : WTAPS * BM * KP);  // floats per stage: [X rows][W rows]
    const int STAGE4 = STAGE >> 2;               // the same in float4 units (every term is a multiple of 4)


    // A operand (activations): lane's GEMM row n -> LDS row of tap 0
    const int n_loc = tn * 32 + l32;
    // Transposed conv as two 2-tap phases: even outputs read positions (l-1, l), odd outputs
    // (l, l+1); the M tile's phase shifts the first row by one.
    const int phase_shift = (TAPS == 2 && p.interleave && m0 >= (p.M >> 1)) ? 1 : 0;
    const int arow = ((n_loc >> p.lshift) * SEG + (n_loc & (Lout - 1)) * STRIDE + phase_shift) * KP + 4 * h +
                     (int)((p.xswz >> (4 * (n_loc >> p.lshift))) & 15) * 4;
    // B operand (weights): lane's output channel
    const int brow = XF + (tm * 32 + l32) * KP + 4 * h;

    // four independent accumulation chains per wave (one per k-step of a unit), summed pairwise in
    // the epilogue: 4x shorter fp32 chains than a single accumulator (K reaches 20480 on the wide
    // nets), at no cost in matrix-pipe time
    f32x16 acc, acc2, acc3, acc4;
    f32x16 accr, accr2;                          // RES: the riding 1x1 conv (two chains)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; acc2[r] = 0.0f; acc3[r] = 0.0f; acc4[r] = 0.0f; accr[r] = 0.0f; accr2[r] = 0.0f; }

    const int cin = p.cin0 + p.cin1;
    const int c_begin = kb * p.chunks_per_slice;
#ifdef DAD_ABLATE_NOLOOP
    const int nchunks = c_begin + 1;
#else
    // chunks that hold real channels (cin_pad may add all-zero chunks; they are never visited)
    const int nchunks = min((p.cin0 + p.cin1 + KC - 1) / KC, c_begin + p.chunks_per_slice);   // one past the last
#endif

    // ---- staging: global -> registers (prefetch) -> LDS ---------------------------------
    constexpr int W_F4 = WTAPS * BM * KC / 4;
    constexpr int W_PER_T = BDIR ? 0 : (W_F4 + NT - 1) / NT;   // staged W items (none in direct-B mode)
    constexpr int W_ARR = W_PER_T ? W_PER_T : 1;
    constexpr int KQ = KC / 4;                              // float4 per row
    constexpr int X_F4_MAX = BN * STRIDE * KQ;              // SPT*Lin == BN*STRIDE rows
    constexpr int X_PER_T = (X_F4_MAX + NT - 1) / NT;
    const int xrows_real = SPT * Lin;
    float4 wreg[W_ARR];
    float4 xreg[X_PER_T];
    // Per-thread staging addresses are chunk-invariant up to a uniform stride: computed once.
    int w_goff[W_ARR], w_loff[W_ARR];                      // global / LDS float offsets (W)
    int x_grow[X_PER_T], x_q4[X_PER_T], x_loff[X_PER_T];   // global row, channel quad, LDS (X)
    constexpr int GQ = KG / 4;                              // float4 per packed row
#pragma unroll
    for (int i = 0; i < W_PER_T; ++i) {
        const int e = tid + i * NT;
        const int row = e / KQ;                             // tap*BM + m
        const int q = e - row * KQ;
        const int tap = row / BM;
        const int mm = row - tap * BM;
        const int sub = q / GQ;
        w_goff[i] = ((sub * p.wtaps + tap) * M + m0 + mm) * KG + (q - sub * GQ) * 4;
        w_loff[i] = (XF + row * KP + q * 4) >> 2;         // float4 units
    }
#pragma unroll
    for (int i = 0; i < X_PER_T; ++i) {
        const int e = tid + i * NT;
        const int row = e / KQ;                             // s*Lin + l
        const int q = e - row * KQ;
        const int s = row >> p.lshift_in;                   // Lin is a power of two
        const int l = row - (s << p.lshift_in);
        const bool ok = e < xrows_real * KQ && s < nvalid;
        x_grow[i] = ok ? s0 * Lin + row : -1;
        if constexpr (PADDED)
            if (p.src_len > 0)                              // external trajectory of a zero-padded horizon
                x_grow[i] = (ok && l < p.src_len) ? (s0 + s) * p.src_len + l : -1;
        x_q4[i] = q * 4;
        // split-f16 rows: per 16-channel unit [8 floats of hi halves | 8 floats of lo halves]
        const int qoff = X3 ? (q >> 2) * 16 + (q & 3) * 2 : q * 4;
        // float4 units (fp32 rows) or float2 units (split-f16 rows: a float4 becomes 8 bytes of hi
        // halves and, 8 floats further, 8 bytes of lo halves)
        const int xo = (s * SEG + PAD + l) * KP + qoff + (int)((p.xswz >> (4 * s)) & 15) * 4;
        x_loff[i] = e < xrows_real * KQ ? (X3 ? xo >> 1 : xo >> 2) : -1;
    }
    DAD_PSTAMP(6);
    const long w_chunk_stride = (long)NSUB * p.wtaps * M * KG;
    // RAGGED = false promises (host-checked) that every K chunk lies inside one concat source and
    // below cin, with 16-byte aligned channel quads: the X loads use one base pointer per chunk.
    // RAGGED = true is the general path: first layer (cin = transition_dim), narrow nets.
    //
    // Rolling staging.  A staged "item" is one float4 of W or X per thread; a chunk has NLD of
    // them.  In unit (k * UW) / NLD of chunk ch, item k — loaded during chunk ch-1 with the data of
    // chunk ch+1 — is written to the other LDS stage and its registers immediately receive the
    // load for chunk ch+2.  Loads and LDS writes are spread evenly over the chunk instead of
    // bunching around the barrier, each load has a whole chunk period to land, and NLD loads per
    // thread are always in flight.  (The dealing keeps the issue order inside a chunk equal to the
    // natural order 0..NLD-1 of the prologue, so "NLD-1 younger loads" holds on every path into
    // the loop and hipcc's s_waitcnt comes out exact; the steady-state body is branch-free for
    // the same reason — a load under a condition makes hipcc fall back to vmcnt(0), a full L2
    // round trip exposed per unit.  The last two chunks run as peeled variants.)
    constexpr int NLD = W_PER_T + X_PER_T;
    auto item_store = [&](int k, int stage) {
        const int base4 = stage * STAGE4;
        if (k < W_PER_T) {
            if (W_F4 % NT == 0 || tid + k * NT < W_F4)
                smem4[base4 + w_loff[k]] = wreg[k];
        } else {
            const int i = k - W_PER_T;
            if (x_loff[i] >= 0) {
                float4 v = xreg[i];
                if (!RAGGED && x_grow[i] < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);   // no such sample
                if constexpr (X3) {
                    float2 hi, lo;
                    split_f16x4(v, hi, lo);
                    smem2[base4 * 2 + x_loff[i]] = hi;
                    smem2[base4 * 2 + x_loff[i] + 4] = lo;
                } else {
                    smem4[base4 + x_loff[i]] = v;
                }
            }
        }
    };
    auto item_load = [&](int k, int chunk) {
        if (k < W_PER_T) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (W_F4 % NT == 0 || tid + k * NT < W_F4)
                v = *reinterpret_cast<const float4*>(p.w + chunk * w_chunk_stride + w_goff[k]);
            wreg[k] = v;
        } else {
            const int i = k - W_PER_T;
            const int c0 = chunk * KC;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (!RAGGED) {
                // whole chunk inside one source and inside cin: one pointer per chunk.  The load is
                // unconditional (rows that do not exist read row 0 and are zeroed at the store)
                const bool second = c0 >= p.cin0;
                const float* xsrc = second ? p.src1 : p.src0;
                const int cs = second ? p.cin1 : p.cin0;      // row stride of the source
                const int cb = second ? c0 - p.cin0 : c0;
                v = *reinterpret_cast<const float4*>(xsrc + (long)max(x_grow[i], 0) * cs + cb + x_q4[i]);
            } else {
                const int cc = c0 + x_q4[i];                  // first channel of this float4
                if (x_grow[i] >= 0 && cc < cin) {             // beyond cin: zero padding of the chunk
                    // virtual concat: channels [0, cin0) come from src0, the rest from src1
                    const bool second = cc >= p.cin0;
                    const int cs = second ? p.cin1 : p.cin0;  // row stride of that source
                    const int cl = second ? cc - p.cin0 : cc;
                    const float* g = (second ? p.src1 : p.src0) + (long)x_grow[i] * cs + cl;
                    const int left = cs - cl;                 // channels remaining in this source
                    if ((cs & 3) == 0) {
                        v = *reinterpret_cast<const float4*>(g);
                    } else {                                  // first layer: cin = transition_dim
                        if (left > 0) v.x = g[0];
                        if (left > 1) v.y = g[1];
                        if (left > 2) v.z = g[2];
                        if (left > 3) v.w = g[3];
                    }
                }
            }
            xreg[i] = v;
        }
    };

    // ---- MFMA operand fragments ---------------------------------------------------------------
    // Fragments are read one unit ahead of the MFMAs that consume them; the chunk's single barrier
    // sits in FRONT of its last unit's MFMAs: the first fragments of the next chunk are fetched
    // behind the barrier while those MFMAs run, so the matrix pipe does not drain at the barrier.
    constexpr int UW = WTAPS * GW;                      // units per wave per chunk
    const int koff = ks * (GW * KU);                    // this wave's units in a chunk
    const int afrag4 = (arow + koff) >> 2;              // float4 units: the constant parts of the
    const int bfrag4 = (brow + koff) >> 2;              // fragment addresses fold into ds_read offsets
    auto frag_a = [&](int stage, int u, int lo) -> float4 {   // lo = 8: the residual halves (X3)
        const int wtap = u / GW, gw = u - wtap * GW;
        const int tap = (RES && wtap == TAPS) ? PAD : wtap;   // the 1x1 ride reads the centre rows
        return smem4[stage * STAGE4 + afrag4 + tap * (KP / 4) + gw * (KU / 4) + lo / 4];
    };
    auto frag_b = [&](int stage, int u, int lo) -> float4 {
        const int tap = u / GW, gw = u - tap * GW;
        return smem4[stage * STAGE4 + bfrag4 + tap * (BM * KP / 4) + gw * (KU / 4) + lo / 4];
    };
    // direct-B mode: this lane's weight fragments of every unit of one chunk (hi, lo), straight from
    // the packed image  [granule][tap][M][16 floats = 16 hi halves | 16 lo halves]
    constexpr int BPU = X3 ? 2 : 1;                     // fragment registers (float4) per unit
    float4 breg[BDIR ? BPU * UW : 1];
    const float* bsrc = p.w + ((long)(m0 + tm * 32 + l32) * 16 + 4 * h);
    auto bload = [&](int u, int chunk) {
        const int tap = u / GW, gw = u - tap * GW;
        if constexpr (X3) {                             // unit = one 16-channel granule: hi, lo
            const float* q = bsrc + (long)chunk * w_chunk_stride + (long)((gw * p.wtaps + tap) * M) * 16;
            breg[2 * u] = ldg4(q);
            breg[2 * u + 1] = ldg4(q + 8);
        } else {                                        // unit = 8 channels = half a granule
            const float* q = bsrc + (long)chunk * w_chunk_stride +
                             (long)(((gw >> 1) * p.wtaps + tap) * M) * 16 + (gw & 1) * 8;
            breg[u] = ldg4(q);
        }
    };

    // The first stage loads go out before anything else is computed: everything up to the first
    // LDS store (epilogue ownership arithmetic, parameter fetches, halo zeroing) runs under their
    // latency instead of in front of it.
#pragma unroll
    for (int k = 0; k < NLD; ++k) item_load(k, c_begin);
    if constexpr (BDIR) {
#pragma unroll
        for (int u = 0; u < UW; ++u) bload(u, c_begin);
    }

    // ---- epilogue ownership (decided up front so its global loads can fly under the K loop) --
    // After the LDS exchange each thread owns F4PL float4 (4 channels x 1 position) of one
    // (GroupNorm group, sample) pair; lanes of a pair are contiguous.  Without GroupNorm the
    // mapping is plain row-major.
    constexpr int ES = BM + 4;
    constexpr int ECOPY = BN * ES;
    constexpr int F4PL = (BN * BM / 4) / NT;           // float4 per thread
    static_assert(F4PL >= 1 && (BN * BM / 4) % NT == 0, "epilogue mapping");
#ifdef DAD_ABLATE_GN
    const bool has_gn = false;
#else
    const bool has_gn = p.gamma != nullptr;
#endif
    const int cpg = has_gn ? p.cpg : BM;
    const int cq = cpg >> 2;                           // float4 per row of a pair
    const int cnt4 = has_gn ? (Lout * cq) : (BN * cq); // float4 per pair
    static_assert((F4PL & (F4PL - 1)) == 0, "F4PL is a power of two");
    const int lpp = cnt4 >> (F4PL == 1 ? 0 : F4PL == 2 ? 1 : F4PL == 4 ? 2 : 3);                       // lanes per pair (power of two >= 1)
    // every quantity here is a power of two: shifts, not divisions (this runs before the first
    // global load can be issued)
    const int lpp_sh = 31 - __clz(lpp);
    const int cq_sh = 31 - __clz(cq);
    const int gpt_sh = 31 - __clz(BM) - (31 - __clz(cpg));
    const int pr = tid >> lpp_sh;
    const int lp = tid & (lpp - 1);
    const int ps = has_gn ? pr >> gpt_sh : 0;          // sample of the pair
    const int pg = has_gn ? pr & ((1 << gpt_sh) - 1) : 0;   // group of the pair
    int erow[F4PL], ecol[F4PL];
    int eoff[F4PL];                                    // output offsets fit 31 bits (host-checked)
    float4 bias4[F4PL], gam4[F4PL], bet4[F4PL], temb4[F4PL], res4[F4PL];
#pragma unroll
    for (int k = 0; k < F4PL; ++k) {
        const int j = lp + k * lpp;                    // float4 index inside the pair
        const int r = j >> cq_sh;
        erow[k] = ps * Lout + r;                       // tile row (position)
        ecol[k] = pg * cpg + (j & (cq - 1)) * 4;       // tile column (channel)
        const int s = erow[k] >> p.lshift;
        const int l = erow[k] & (Lout - 1);
        const int em = m0 + ecol[k];
        if (!p.interleave) {
            eoff[k] = ((s0 + s) * Lout + l) * M + em;
        } else {
            const int half = M >> 1;
            const int phase = em >= half;
            eoff[k] = ((s0 + s) * (2 * Lout) + 2 * l + phase) * half + (em - phase * half);
        }
        if (s >= nvalid) eoff[k] = -1;
    }
    // Per-channel parameters and the residual tile.  With few registers at stake (F4PL <= 2)
    // they are fetched before the K loop (their latency merges with the first weight loads);
    // wide-tile variants fetch them after the LDS exchange instead.
    constexpr bool EARLY_PARAMS = F4PL <= 2;
    // (no lambda here: capturing the ownership arrays by reference would park them in scratch)
#define DAD_FETCH_PARAMS()                                                                       \
    _Pragma("unroll") for (int k = 0; k < F4PL; ++k) {                                           \
        const int em = m0 + ecol[k];                                                             \
        long toff = 0;                                                                           \
        if (p.trow != nullptr)        /* per-row timesteps: the training-side forward only */    \
            toff = (long)p.trow[min(s0 + (erow[k] >> p.lshift), p.B - 1)] * p.temb_stride;       \
        bias4[k] = ldg4(p.bias + em);                                                            \
        gam4[k] = ldg4(gptr + em);                                                               \
        bet4[k] = ldg4(bptr + em);                                                               \
        temb4[k] = ldg4(tptr + toff + em);                                                       \
        res4[k] = ldg4(has_res ? p.res + max(eoff[k], 0) : p.bias + em);                         \
    }
    // Every parameter load is UNCONDITIONAL: an absent operand reads the bias row instead and is
    // cancelled by a select where it is used (has_temb / has_res below).  A load under `if (p.gamma)` made hipcc copy the
    // loaded register at the end of the conditional block (a phi), with an s_waitcnt in front of the
    // copy that drained the stage loads issued before it: one exposed memory round trip (~0.6 us)
    // in the prologue of every GroupNorm'd launch (found with -DDAD_STAMPS_PROLOGUE + the ISA).
    const float* const gptr = has_gn ? p.gamma : p.bias;
    const float* const bptr = has_gn ? p.beta : p.bias;
    const float* const tptr = p.temb != nullptr ? p.temb : p.bias;
    const bool has_res = p.res != nullptr && !p.interleave;
    const bool has_temb = p.temb != nullptr;
    DAD_PSTAMP(7);
    if (EARLY_PARAMS) { DAD_FETCH_PARAMS() }   // younger than the stage loads: not waited with them
    DAD_PSTAMP(1);
    // Zero the halo rows of both X stages once (PAD rows on either side of every sample).  Real
    // rows — including those of samples beyond the batch, which get zeros — are rewritten in full
    // by every chunk's staging, so nothing else needs clearing and, the two sets of rows being
    // disjoint, no barrier separates this from the first stores.
    if constexpr (PAD > 0) {
        constexpr int KP4 = KP / 4;
        const int nz = SPT * (2 * PAD) * KP4;
        for (int i = tid; i < nz; i += NT) {
            const int hr = i / KP4, c4 = i - hr * KP4;
            const int s = hr / (2 * PAD), j = hr - s * (2 * PAD);
            const int row = s * SEG + (j < PAD ? j : Lin + j);
            const int at = row * KP + c4 * 4 + (int)((p.xswz >> (4 * s)) & 15) * 4;
            smem4[at >> 2] = make_float4(0.f, 0.f, 0.f, 0.f);
            smem4[STAGE4 + (at >> 2)] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    DAD_PSTAMP(2);
#pragma unroll
    for (int k = 0; k < NLD; ++k) item_store(k, 0);
    DAD_PSTAMP(3);
    if (c_begin + 1 < nchunks) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) item_load(k, c_begin + 1);
    }
    DAD_PSTAMP(4);
    __syncthreads();
    DAD_PSTAMP(5);

    // ---- main loop -------------------------------------------------------------------------
    // fp32: a unit is 8 channels = 4 v_mfma_f32_32x32x2_f32 into four accumulation chains.
    // split-f16: a unit is 16 channels = 3 v_mfma_f32_32x32x16_f16 (hi*hi into acc, the two cross
    // terms into acc2 / acc3, which carry a factor 2^11).
    // (DAD_ABLATE_* exist only in timing-only diagnostic builds: wrong results by design.)
    float4 ah = frag_a(0, 0, 0), bh = ah;
    if constexpr (!BDIR) bh = frag_b(0, 0, 0);
    float4 al = ah, bl = bh;
    if constexpr (X3) {
        al = frag_a(0, 0, 8);
        if constexpr (!BDIR) bl = frag_b(0, 0, 8);
    }
    DAD_STAMP(1);
    DAD_CLOCK(6);
#ifndef DAD_ABLATE_STAGE
#define DAD_ROLL(STORE, LOAD)                                                                    \
    _Pragma("unroll") for (int k = 0; k < NLD; ++k) {                                            \
        if ((BDIR ? UW - 1 : (k * UW) / NLD) != u) continue;   /* direct-B: after the B loads */    \
        if (STORE) item_store(k, cur ^ 1);                                                       \
        if (LOAD) item_load(k, ch + 2);                                                          \
    }
#else
#define DAD_ROLL(STORE, LOAD)
#endif
#ifndef DAD_ABLATE_BARRIER
#define DAD_SYNC() __syncthreads();
#else
#define DAD_SYNC()
#endif
#ifndef DAD_ABLATE_LDSREAD
#define DAD_READ(ST, U)                                                                          \
    nah = frag_a(ST, U, 0);                                                                      \
    if constexpr (X3) nal = frag_a(ST, U, 8);                                                    \
    if constexpr (!BDIR) {                                                                       \
        nbh = frag_b(ST, U, 0);                                                                  \
        if constexpr (X3) nbl = frag_b(ST, U, 8);                                                \
    }
#else
#define DAD_READ(ST, U)
#endif
#ifndef DAD_ABLATE_MFMA
#define DAD_MFMA()                                                                               \
    if constexpr (BDIR && X3) { bh = breg[2 * u]; bl = breg[2 * u + 1]; }                        \
    if constexpr (BDIR && !X3) bh = breg[u];                                                     \
    if constexpr (X3) {                                                                          \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah),              \
                                                     __builtin_bit_cast(f16x8, bh), acc, 0, 0, 0);  \
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah),             \
                                                      __builtin_bit_cast(f16x8, bl), acc2, 0, 0, 0); \
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al),             \
                                                      __builtin_bit_cast(f16x8, bh), acc3, 0, 0, 0); \
    } else if (RES && u / GW == TAPS) {                                                          \
        accr = __builtin_amdgcn_mfma_f32_32x32x2f32(ah.x, bh.x, accr, 0, 0, 0);                  \
        accr2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ah.y, bh.y, accr2, 0, 0, 0);                \
        accr = __builtin_amdgcn_mfma_f32_32x32x2f32(ah.z, bh.z, accr, 0, 0, 0);                  \
        accr2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ah.w, bh.w, accr2, 0, 0, 0);                \
    } else {                                                                                     \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ah.x, bh.x, acc, 0, 0, 0);                    \
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ah.y, bh.y, acc2, 0, 0, 0);                  \
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(ah.z, bh.z, acc3, 0, 0, 0);                  \
        acc4 = __builtin_amdgcn_mfma_f32_32x32x2f32(ah.w, bh.w, acc4, 0, 0, 0);                  \
    }
#else
#define DAD_MFMA() acc[0] += ah.x * bh.x + al.y * bl.y;
#endif
    // STORE: chunk ch+1 exists (its operands go to the other stage); LOAD: chunk ch+2 exists
#define DAD_CHUNK(STORE, LOAD)                                                                   \
    {                                                                                            \
        const int cur = (ch - c_begin) & 1;                                                      \
        _Pragma("unroll") for (int u = 0; u < UW; ++u) {                                         \
            float4 nah = ah, nal = al, nbh = bh, nbl = bl;                                       \
            DAD_ROLL(STORE, LOAD)                                                                \
            if (u + 1 < UW) {                                                                    \
                DAD_READ(cur, u + 1)                                                             \
            } else {                                                                             \
                DAD_SYNC()                                                                       \
                if (STORE) { DAD_READ(cur ^ 1, 0) }                                              \
            }                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            DAD_MFMA()                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            if constexpr (BDIR) { if (STORE) bload(u, ch + 1); }                                 \
            ah = nah; al = nal; bh = nbh; bl = nbl;                                              \
        }                                                                                        \
    }
    int ch = c_begin;
    for (; ch + 2 < nchunks; ++ch) DAD_CHUNK(true, true)
    if (ch + 1 < nchunks) { DAD_CHUNK(true, false) ++ch; }
    if (ch < nchunks) DAD_CHUNK(false, false)
#undef DAD_CHUNK
#undef DAD_ROLL
#undef DAD_SYNC
#undef DAD_READ
#undef DAD_MFMA
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if constexpr (X3) acc[r] = acc[r] * p.c1 + (acc2[r] + acc3[r]) * p.c2;
        else acc[r] = (acc[r] + acc2[r]) + (acc3[r] + acc4[r]);
    }
    __syncthreads();                       // all MFMAs retired before the stage memory is reused
    DAD_STAMP(2);
    DAD_CLOCK(7);

    // ---- epilogue ---------------------------------------------------------------------------
    // 1. every wave drops its accumulators into its split-K copy of the tile E[ks][n][m] (LDS);
    // 2. after ONE barrier each thread owns F4PL float4 (4 channels x 1 position) of one
    //    (GroupNorm group, sample) pair: it folds the copies and the bias into registers,
    //    takes part in the pair's mean / variance reductions (two-pass, fp32, fixed order:
    //    xor-shuffles inside the wave, an LDS hop when a pair spans several waves), then
    //    normalises, applies Mish, adds time embedding / residual and stores — without touching
    //    LDS again.
    float* E = smem;                                   // [SK][BN][ES]
    float* red = smem + SK * ECOPY;                    // [2][waves] cross-wave partials
#ifdef DAD_ABLATE_NOEPI
    if (p.B > 0) { if (acc[0] == 123.456f) p.dst[0] = acc[1]; return; }
#endif
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int col = tm * 32 + l32;
        E[ks * ECOPY + row * ES + col] = acc[r];
    }
    __syncthreads();
    DAD_STAMP(3);
    if (!EARLY_PARAMS) { DAD_FETCH_PARAMS() }
#undef DAD_FETCH_PARAMS

    float y[F4PL][4];
#pragma unroll
    for (int k = 0; k < F4PL; ++k) {
        const float* q = E + erow[k] * ES + ecol[k];
        float4 v = *reinterpret_cast<const float4*>(q);
#pragma unroll
        for (int c = 1; c < SK; ++c) {
            const float4 u = *reinterpret_cast<const float4*>(q + c * ECOPY);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        y[k][0] = v.x; y[k][1] = v.y; y[k][2] = v.z; y[k][3] = v.w;
    }

    if constexpr (RES) {
        // The riding 1x1 conv: same exchange through E, same ownership (any bijection of tile
        // elements onto threads will do without GroupNorm), bias, store.  Its stores fly while
        // the GroupNorm reductions below run.
        __syncthreads();                               // every thread has read its main-tile values
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int col = tm * 32 + l32;
            E[ks * ECOPY + row * ES + col] = accr[r] + accr2[r];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < F4PL; ++k) {
            const float* q = E + erow[k] * ES + ecol[k];
            float4 v = *reinterpret_cast<const float4*>(q);
#pragma unroll
            for (int c = 1; c < SK; ++c) {
                const float4 u = *reinterpret_cast<const float4*>(q + c * ECOPY);
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            const float4 rb = ldg4(p.rbias + m0 + ecol[k]);
            if (eoff[k] >= 0)
                store_f4_sc1(p.rdst + eoff[k], make_float4(v.x + rb.x, v.y + rb.y, v.z + rb.z, v.w + rb.w));
        }
    }

    if (p.kslices > 1) {
        // Grid-level split-K: every block of the tile parks its partial tile in HBM; the block
        // that draws the last ticket adds all of them in slice order (bit-reproducible whichever
        // block is last) and goes on to the epilogue alone.  Placement-independent hand-off
        // (cdna guide G16, recipe R1): write-through (sc1) slab stores -> every wave drains ->
        // barrier -> ticket (agent-scope atomic; no release fence: nothing is left dirty in L2);
        // the reducer takes one agent-scope acquire before any slab load.
        const int KS = p.kslices;
        float* mine = p.slab + ((long)tile * KS + kb) * (BN * BM);
#pragma unroll
        for (int k = 0; k < F4PL; ++k)
            if (eoff[k] >= 0)                         // rows of samples that exist (small batches)
                store_f4_sc1(mine + erow[k] * BM + ecol[k], make_float4(y[k][0], y[k][1], y[k][2], y[k][3]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(smem + SK * ECOPY + 32);
        if (tid == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(p.counters + tile, 1u, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = ticket == (unsigned)(KS - 1);
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(p.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0u) return;
        const float* base = p.slab + (long)tile * KS * (BN * BM);
#pragma unroll
        for (int k = 0; k < F4PL; ++k) {
            if (eoff[k] < 0) continue;                // padding rows keep their (unused) partials
            const float* q = base + erow[k] * BM + ecol[k];
            float4 v = ldg4(q);
            // eight slabs per round trip (a one-at-a-time loop paid a full memory latency per slab:
            // 32 slices = 32 round trips in the reducer of a wide layer at batch 1); slabs past
            // the end re-read the last one and are weighted 0 — same sums, slice order kept
            for (int c0 = 1; c0 < KS; c0 += 8) {
                float4 u[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) u[j] = ldg4(q + (long)min(c0 + j, KS - 1) * (BN * BM));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float mj = c0 + j < KS ? 1.0f : 0.0f;
                    v.x = fmaf(u[j].x, mj, v.x); v.y = fmaf(u[j].y, mj, v.y);
                    v.z = fmaf(u[j].z, mj, v.z); v.w = fmaf(u[j].w, mj, v.w);
                }
            }
            y[k][0] = v.x; y[k][1] = v.y; y[k][2] = v.z; y[k][3] = v.w;
        }
    }
#pragma unroll
    for (int k = 0; k < F4PL; ++k) {
        y[k][0] += bias4[k].x; y[k][1] += bias4[k].y; y[k][2] += bias4[k].z; y[k][3] += bias4[k].w;
    }

    if (p.pre != nullptr) {                            // (block-uniform: stores only, no load sits under it)
#pragma unroll
        for (int k = 0; k < F4PL; ++k)
            if (eoff[k] >= 0)
                *reinterpret_cast<float4*>(p.pre + eoff[k]) = make_float4(y[k][0], y[k][1], y[k][2], y[k][3]);
    }

    if (has_gn) {
        const int creal = (PADDED && p.cpg_real > 0) ? p.cpg_real : cpg;      // channels of a group that exist
        const int lr = (PADDED && p.lreal > 0) ? p.lreal : Lout;               // positions of a sample that exist
        const float inv_cnt = PADDED ? 1.0f / (float)(lr * creal) : 1.0f / (float)(cnt4 * 4);
        const int width = lpp < 64 ? lpp : 64;
        const int wpp = lpp >> 6;                      // waves per pair when a pair spans waves
        auto pair_sum = [&](float v, int slot) -> float {
            // all-reduce over the `width` contiguous lanes of the pair with cross-lane VALU ops
            // (DPP quad / mirror permutes inside a row, v_readlane across rows): no LDS round trips
            if (width >= 2) v += dpp_f32<0xB1>(v);           // quad_perm [1,0,3,2]
            if (width >= 4) v += dpp_f32<0x4E>(v);           // quad_perm [2,3,0,1]
            if (width >= 8) v += dpp_f32<0x141>(v);          // row_half_mirror
            if (width >= 16) v += dpp_f32<0x140>(v);         // row_mirror
            if (width >= 32) v = rows_sum(v, width, lane);   // rows 0+1 | 2+3, then halves
            if (wpp > 1) {                             // block-uniform branch
                if (lane == 0) red[slot * 16 + wave] = v;
                __syncthreads();
                const int w0 = (wave / wpp) * wpp;
                v = 0.0f;
                for (int w = 0; w < wpp; ++w) v += red[slot * 16 + w0 + w];
            }
            return v;
        };
        float sum = 0.0f;
        if constexpr (PADDED) {
            // padded positions hold conv + bias of the edge rows (not zero): left out of both passes; padded channels
            // hold exactly zero: nothing for the sum, masked in the variance
#pragma unroll
            for (int k = 0; k < F4PL; ++k)
                if ((erow[k] & (Lout - 1)) < lr) sum += (y[k][0] + y[k][1]) + (y[k][2] + y[k][3]);
        } else {
#pragma unroll
            for (int k = 0; k < F4PL; ++k) sum += (y[k][0] + y[k][1]) + (y[k][2] + y[k][3]);
        }
        const float mean = pair_sum(sum, 0) * inv_cnt;
        float sq = 0.0f;
        if constexpr (PADDED) {
#pragma unroll
            for (int k = 0; k < F4PL; ++k) {
                const int cl = ecol[k] & (cpg - 1);    // channel of the float4 inside its group
                const bool row_on = (erow[k] & (Lout - 1)) < lr;
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float d = (row_on && cl + c < creal) ? y[k][c] - mean : 0.0f; sq += d * d; }
            }
        } else {
#pragma unroll
            for (int k = 0; k < F4PL; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float d = y[k][c] - mean; sq += d * d; }
        }
        const float rstd = 1.0f / sqrtf(pair_sum(sq, 1) * inv_cnt + 1e-5f);
        if (p.stats != nullptr && lp == 0 && eoff[0] >= 0) {      // one lane per (sample, group) pair
            float* st = p.stats + ((long)(s0 + ps) * (M >> (31 - __clz(cpg))) + ((m0 >> (31 - __clz(cpg))) + pg)) * 2;
            st[0] = mean; st[1] = rstd;
        }
#pragma unroll
        for (int k = 0; k < F4PL; ++k) {
            y[k][0] = mish_fast_f32((y[k][0] - mean) * rstd * gam4[k].x + bet4[k].x);
            y[k][1] = mish_fast_f32((y[k][1] - mean) * rstd * gam4[k].y + bet4[k].y);
            y[k][2] = mish_fast_f32((y[k][2] - mean) * rstd * gam4[k].z + bet4[k].z);
            y[k][3] = mish_fast_f32((y[k][3] - mean) * rstd * gam4[k].w + bet4[k].w);
        }
    }
    DAD_STAMP(4);
#pragma unroll
    for (int k = 0; k < F4PL; ++k) {
        if (eoff[k] < 0) continue;
        if constexpr (PADDED) {                        // zero-padded horizon: the padding stays zero
            if (p.lreal > 0 && (erow[k] & (Lout - 1)) >= p.lreal) {
                store_f4_sc1(p.dst + eoff[k], make_float4(0.f, 0.f, 0.f, 0.f));
                continue;
            }
        }
        // absent operands were loaded from the bias row: cancelled with a select (v_cndmask), not a
        // multiply by 0 — 0 * Inf would turn a non-finite word of an unrelated tensor into NaN here
        const float4 t4 = has_temb ? temb4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 r4 = has_res ? res4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 o4 = make_float4(y[k][0] + (t4.x + r4.x), y[k][1] + (t4.y + r4.y),
                                      y[k][2] + (t4.z + r4.z), y[k][3] + (t4.w + r4.w));
        // write-through: nothing of the tile is left dirty in this XCD's L2 for the end-of-kernel
        // release to write back (+0.5 % at batch 256, +0.7 % on Door; same bytes)
        store_f4_sc1(p.dst + eoff[k], o4);
    }
    DAD_STAMP(5);
}

}  // namespace dad
