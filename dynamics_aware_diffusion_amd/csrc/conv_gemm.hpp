// conv_gemm.hpp — Conv1d / ConvTranspose1d of the TemporalUnet as an implicit GEMM on the
// gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32), with the whole Conv1dBlock tail
// (bias -> GroupNorm(8) -> Mish -> + time embedding -> + residual) fused in the epilogue.
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
//   * weights are pre-packed on the host as [C_in/KC][tap][M][KC] so that both MFMA operands
//     are fetched with one ds_read_b128 per four MFMAs: lane half h of the wave owns input
//     channels 4h..4h+3 of every 8-channel group (any bijection of K onto (step, half) is a
//     valid summation order as long as both operands use it).
//   * SK waves share one 32x32 output tile and split the K units between them (intra-block
//     split-K): at batch 256 a layer has only ~1 output tile per SIMD, the second wave per
//     SIMD is what hides LDS/barrier latency.  Partial sums meet in the LDS epilogue tile.
//   * a transposed conv (k=4, s=2, p=1) runs as a 3-tap conv with M = 2*C_out columns
//     (even-phase columns, then odd-phase; unused taps are zero) and an interleaving store.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dad {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float* src0;   // [B*Lin][cin0]
    const float* src1;   // [B*Lin][cin1]  second half of a virtual channel concat, or nullptr
    const float* w;      // packed [cin_pad/KG][TAPS][M][KG], KG = min(KC, 16)
    const float* bias;   // [M]
    const float* gamma;  // [M] GroupNorm weight, or nullptr (no norm, no Mish)
    const float* beta;   // [M]
    const float* temb;   // [M] time-embedding projection for this t, or nullptr
    const float* res;    // [B*Lout][M] residual to add after Mish, or nullptr
    float* dst;          // [B*Lout][M]; interleave mode: [B*2*Lout][M/2]
    int32_t cin0, cin1;  // channels taken from src0 / src1
    int32_t cin_pad;     // (cin0+cin1) rounded up to KC
    int32_t M;           // GEMM columns (2*C_out for the transposed conv)
    int32_t cpg;         // channels per GroupNorm group (M/8) when gamma != nullptr
    int32_t B;           // batch rows in this call
    int32_t Lin, Lout;   // per-sample input / output length of the GEMM
    int32_t lshift;      // log2(Lout)
    int32_t interleave;  // transposed-conv store: col m<M/2 -> row 2l, m>=M/2 -> row 2l+1
    int32_t ntiles_n;    // number of N tiles (for the XCD-aware tile order)
};

__device__ __forceinline__ float mish_f32(float y) {
    // x * tanh(softplus(x)), softplus threshold 20 (torch.nn.functional.mish semantics).
    // tanh(log(1+e^x)) = ((1+e^x)^2 - 1) / ((1+e^x)^2 + 1) = w / (w + 2), w = e^x (e^x + 2)
    if (y > 20.0f) return y;
    const float n = expf(y);
    const float w = n * (n + 2.0f);
    return y * (w / (w + 2.0f));
}

// Rows of the X stage: every sample of the tile with its zero halo.
__host__ __device__ inline int conv_xrows(int BN, int Lin, int Lout, int taps) {
    return (BN / Lout) * (Lin + 2 * (taps / 2));
}
// LDS floats of one block (the host sizes the dynamic allocation with the same formula).
__host__ __device__ inline size_t conv_lds_floats(int BM, int BN, int KC, int taps, int Lin,
                                                  int Lout) {
    const size_t kp = KC + 4;
    const size_t stage = ((size_t)conv_xrows(BN, Lin, Lout, taps) + (size_t)taps * BM) * kp;
    const size_t epi = (size_t)BN * (BM + 4) + 2 * 512;
    const size_t k = 2 * stage;
    return k > epi ? k : epi;
}

template <int BM, int BN, int SK, int KC, int TAPS, int STRIDE>
__global__ __launch_bounds__(64 * (BM / 32) * (BN / 32) * SK) void conv_gemm_f32(const ConvParams p) {
    constexpr int TMW = BM / 32;                 // wave tiles along M
    constexpr int TNW = BN / 32;                 // wave tiles along N
    constexpr int WT = TMW * TNW;
    constexpr int NT = 64 * WT * SK;
    constexpr int PAD = TAPS / 2;
    constexpr int KP = KC + 4;                   // LDS row stride (floats): 16-B aligned, odd in
                                                 // 16-B units -> conflict-free ds_read_b128
    constexpr int G = KC / 8;                    // 8-channel groups per chunk
    constexpr int GW = G / SK;                   // groups each split-K wave owns, per tap
    constexpr int KG = KC < 16 ? KC : 16;        // packing granule of the weights
    constexpr int NSUB = KC / KG;                // packed granules per chunk
    static_assert(KC % 8 == 0 && G % SK == 0, "K chunk must split evenly over the SK waves");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int wt = wave % WT;
    const int ks = wave / WT;                    // split-K slice of this wave
    const int tn = wt / TMW;
    const int tm = wt % TMW;
    const int l32 = lane & 31;
    const int h = lane >> 5;

    // XCD-aware tile order: hardware deals consecutive block ids round-robin over the 8 XCDs;
    // give every XCD a contiguous run of tiles in M-major order so the blocks sharing one
    // weight slab share an L2 (speed only; any placement is correct).
    int tile;
    {
        const int nblk = gridDim.x;
        const int bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = tile / p.ntiles_n;
    const int nt = tile - mt * p.ntiles_n;

    const int Lin = p.Lin, Lout = p.Lout;
    const int SPT = BN >> p.lshift;              // whole samples per tile
    const int SEG = Lin + 2 * PAD;
    const int XROWS = SPT * SEG;
    const int s0 = nt * SPT;                     // first sample of this tile
    const int m0 = mt * BM;                      // first output channel of this tile
    const int M = p.M;
    const int nvalid = min(SPT, p.B - s0);

    const int XF = XROWS * KP;
    const int STAGE = XF + TAPS * BM * KP;       // floats per stage: [X rows][W rows]

    // zero both X stages once: halo rows (and rows of samples that do not exist) stay zero,
    // staging only ever writes real positions.
    for (int i = tid; i < XF; i += NT) { smem[i] = 0.0f; smem[STAGE + i] = 0.0f; }

    // A operand (activations): lane's GEMM row n -> LDS row of tap 0
    const int n_loc = tn * 32 + l32;
    const int arow = ((n_loc >> p.lshift) * SEG + (n_loc & (Lout - 1)) * STRIDE) * KP + 4 * h;
    // B operand (weights): lane's output channel
    const int brow = XF + (tm * 32 + l32) * KP + 4 * h;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    const int cin = p.cin0 + p.cin1;
    const int nchunks = p.cin_pad / KC;

    // ---- staging: global -> registers (prefetch) -> LDS ---------------------------------
    constexpr int W_F4 = TAPS * BM * KC / 4;
    constexpr int W_PER_T = (W_F4 + NT - 1) / NT;
    constexpr int KQ = KC / 4;                              // float4 per row
    constexpr int X_F4_MAX = BN * STRIDE * KQ;              // SPT*Lin == BN*STRIDE rows
    constexpr int X_PER_T = (X_F4_MAX + NT - 1) / NT;
    const int xrows_real = SPT * Lin;
    float4 wreg[W_PER_T];
    float4 xreg[X_PER_T];

    auto load_stage = [&](int chunk) {
        constexpr int GQ = KG / 4;                          // float4 per packed row
#pragma unroll
        for (int i = 0; i < W_PER_T; ++i) {
            const int e = tid + i * NT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (W_F4 % NT == 0 || e < W_F4) {
                const int row = e / KQ;                         // tap*BM + m
                const int q = e - row * KQ;
                const int tap = row / BM;
                const int mm = row - tap * BM;
                const int sub = q / GQ;
                const long grow = ((long)(chunk * NSUB + sub) * TAPS + tap) * M + m0 + mm;
                v = *reinterpret_cast<const float4*>(p.w + grow * KG + (q - sub * GQ) * 4);
            }
            wreg[i] = v;
        }
        const int c0 = chunk * KC;
        const bool second = c0 >= p.cin0;
        const float* xsrc = second ? p.src1 : p.src0;
        const int cs = second ? p.cin1 : p.cin0;          // row stride of the source
        const int cb = second ? c0 - p.cin0 : c0;
#pragma unroll
        for (int i = 0; i < X_PER_T; ++i) {
            const int e = tid + i * NT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < xrows_real * KQ) {
                const int row = e / KQ;                   // s*Lin + l
                const int q = e - row * KQ;
                const int s = row / Lin;
                if (s < nvalid) {
                    const float* g = xsrc + ((long)s0 * Lin + row) * cs + cb + q * 4;
                    const int left = cs - (cb + q * 4);   // channels remaining in this source
                    if ((cs & 3) == 0 && left >= 4) {
                        v = *reinterpret_cast<const float4*>(g);
                    } else {                              // ragged tail (first layer: cin = td)
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
    auto store_stage = [&](int stage) {
        const int base = stage * STAGE;
#pragma unroll
        for (int i = 0; i < W_PER_T; ++i) {
            const int e = tid + i * NT;
            if (W_F4 % NT == 0 || e < W_F4) {
                const int row = e / KQ;                   // tap*BM + m
                const int q = e - row * KQ;
                *reinterpret_cast<float4*>(&smem[base + XF + row * KP + q * 4]) = wreg[i];
            }
        }
#pragma unroll
        for (int i = 0; i < X_PER_T; ++i) {
            const int e = tid + i * NT;
            if (e < xrows_real * KQ) {
                const int row = e / KQ;
                const int q = e - row * KQ;
                const int s = row / Lin;
                const int l = row - s * Lin;
                *reinterpret_cast<float4*>(&smem[base + (s * SEG + PAD + l) * KP + q * 4]) = xreg[i];
            }
        }
    };
    (void)cin;

    __syncthreads();                       // zero fill done before real rows land
    load_stage(0);
    store_stage(0);
    __syncthreads();

    const int koff = ks * (GW * 8);        // this wave's channel groups inside the chunk
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1;
        const bool more = (ch + 1) < nchunks;
        if (more) load_stage(ch + 1);                    // global loads in flight under MFMA
        const int abase = cur * STAGE + arow + koff;
        const int bbase = cur * STAGE + brow + koff;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
            for (int gw = 0; gw < GW; ++gw) {
                const float4 a = *reinterpret_cast<const float4*>(&smem[abase + tap * KP + gw * 8]);
                const float4 b = *reinterpret_cast<const float4*>(&smem[bbase + tap * BM * KP + gw * 8]);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
        }
        if (more) store_stage(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue through LDS: E[n][m] -----------------------------------------------------
    constexpr int ES = BM + 4;
    float* E = smem;                                   // [BN][ES]
    float* stat_mean = smem + BN * ES;                 // [<=512]
    float* stat_rstd = stat_mean + 512;
#pragma unroll
    for (int s = 0; s < SK; ++s) {
        if (ks == s) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int col = tm * 32 + l32;
                float* e = &E[row * ES + col];
                if (s == 0) *e = acc[r] + p.bias[m0 + col];
                else *e += acc[r];
            }
        }
        __syncthreads();
    }

    const bool has_gn = p.gamma != nullptr;
    if (has_gn) {
        // (group, sample) pairs of this tile; two-pass mean / biased variance in fp32.
        const int cpg = p.cpg;
        const int gpt = BM / cpg;
        const int pairs = gpt * SPT;
        const int cnt = cpg * Lout;
        int tpp = NT / pairs;
        tpp = tpp < 1 ? 1 : (tpp > 64 ? 64 : tpp);
        tpp = 1 << (31 - __clz(tpp));                  // power of two
        const int ppr = NT / tpp;
        const int sub = tid & (tpp - 1);
        const int cshift = 31 - __clz(cpg);
        for (int pr = tid / tpp; pr < pairs; pr += ppr) {
            const int g = pr / SPT;
            const int s = pr - g * SPT;
            const float* base = E + (s * Lout) * ES + g * cpg;
            float sum = 0.0f;
            for (int e = sub; e < cnt; e += tpp) sum += base[(e >> cshift) * ES + (e & (cpg - 1))];
            for (int o = tpp >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            const float mean = sum / (float)cnt;
            float sq = 0.0f;
            for (int e = sub; e < cnt; e += tpp) {
                const float d = base[(e >> cshift) * ES + (e & (cpg - 1))] - mean;
                sq += d * d;
            }
            for (int o = tpp >> 1; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
            if (sub == 0) {
                stat_mean[pr] = mean;
                stat_rstd[pr] = 1.0f / sqrtf(sq / (float)cnt + 1e-5f);
            }
        }
        __syncthreads();
    }

    // normalise + Mish + adds + coalesced store (float4 along channels)
    constexpr int BM4 = BM / 4;
    for (int e = tid; e < BN * BM4; e += NT) {
        const int row = e / BM4;                       // n within tile
        const int c4 = (e - row * BM4) * 4;            // channel within tile
        const int s = row >> p.lshift;
        if (s >= nvalid) continue;
        const int l = row & (Lout - 1);
        const int m = m0 + c4;
        const float4 v = *reinterpret_cast<const float4*>(E + row * ES + c4);
        float y[4] = {v.x, v.y, v.z, v.w};
        if (has_gn) {
            const int pr = (c4 / p.cpg) * SPT + s;
            const float mu = stat_mean[pr], rs = stat_rstd[pr];
            const float4 ga = *reinterpret_cast<const float4*>(p.gamma + m);
            const float4 be = *reinterpret_cast<const float4*>(p.beta + m);
            y[0] = mish_f32((y[0] - mu) * rs * ga.x + be.x);
            y[1] = mish_f32((y[1] - mu) * rs * ga.y + be.y);
            y[2] = mish_f32((y[2] - mu) * rs * ga.z + be.z);
            y[3] = mish_f32((y[3] - mu) * rs * ga.w + be.w);
        }
        if (p.temb != nullptr) {
            const float4 tv = *reinterpret_cast<const float4*>(p.temb + m);
            y[0] += tv.x; y[1] += tv.y; y[2] += tv.z; y[3] += tv.w;
        }
        long off;
        if (!p.interleave) {
            off = ((long)(s0 + s) * Lout + l) * M + m;
            if (p.res != nullptr) {
                const float4 r = *reinterpret_cast<const float4*>(p.res + off);
                y[0] += r.x; y[1] += r.y; y[2] += r.z; y[3] += r.w;
            }
        } else {
            const int half = M >> 1;
            const int phase = m >= half;
            off = ((long)(s0 + s) * (2 * Lout) + 2 * l + phase) * half + (m - phase * half);
        }
        *reinterpret_cast<float4*>(p.dst + off) = make_float4(y[0], y[1], y[2], y[3]);
    }
}

}  // namespace dad
