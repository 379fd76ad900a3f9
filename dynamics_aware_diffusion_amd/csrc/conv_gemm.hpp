// conv_gemm.hpp — Conv1d / ConvTranspose1d of the TemporalUnet as an implicit GEMM on the
// gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32), with the whole Conv1dBlock tail
// (bias -> GroupNorm(8) -> Mish -> + time embedding -> + residual) fused in the epilogue.
//
// Replaces, per launch, the reference's  F.conv1d / F.conv_transpose1d -> F.group_norm ->
// F.mish -> add chains of m_diffuser/models/temporal_unet.py:57-76 (Conv1dBlock),
// :106-122 (ResidualTemporalBlock), :35-54 (Down/Upsample1d).
//
// GEMM view:  Y[M = C_out][N = B*L_out] = W[M][K = taps*C_in] * Xcol[K][N]
//   * activations live channel-major in HBM:  act[c][b*L + l]  ("CNL"), so a tile of
//     BN columns is BN/L_out whole samples and a GroupNorm group (C_out/8 channels x L_out)
//     never straddles a tile when BM is a multiple of C_out/8.
//   * im2col is never materialised: a K-chunk of KC input channels is staged in LDS as
//     KC rows, each row = the tile's samples back to back with `PAD` zero columns on both
//     sides of every sample; tap `j` of the filter is then just "+ j" on the LDS address.
//   * weights are pre-packed on the host as [C_in/8][tap][8][M] so a (chunk, M-tile) slab is
//     TAPS*KC rows of BM contiguous floats.
//   * a transposed conv (k=4, s=2, p=1) is run as a 3-tap conv with M = 2*C_out rows
//     (even-phase rows, then odd-phase rows; unused taps are zero) and an interleaving store.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dad {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float* src0;   // CNL [cin0][B*Lin]  (or external (B, Lin, cin0) when src_nlc)
    const float* src1;   // CNL [cin1][B*Lin]  second half of a virtual channel concat
    const float* w;      // packed [cin_pad/8][TAPS][8][M]
    const float* bias;   // [M]
    const float* gamma;  // [M] GroupNorm weight, or nullptr (no norm, no Mish)
    const float* beta;   // [M]
    const float* temb;   // [M] time-embedding projection for this t, or nullptr
    const float* res;    // CNL [M][B*Lout] residual to add after Mish, or nullptr
    float* dst;          // CNL [M][B*Lout]; interleave mode: [M/2][B*2*Lout]
    int32_t cin0, cin1;  // channels taken from src0 / src1
    int32_t cin_pad;     // (cin0+cin1) rounded up to 8
    int32_t M;           // GEMM rows (2*C_out for the transposed conv)
    int32_t cpg;         // channels per GroupNorm group (M/8) when gamma != nullptr
    int32_t B;           // batch rows in this call
    int32_t Lin, Lout;   // per-sample input / output length of the GEMM
    int32_t lshift;      // log2(Lout)
    int32_t src_nlc;     // src0 is the external (B, Lin, cin0) trajectory tensor
    int32_t interleave;  // transposed-conv store: row m<M/2 -> col 2l, m>=M/2 -> col 2l+1
};

__device__ __forceinline__ float mish_f32(float y) {
    // x * tanh(softplus(x)), softplus threshold 20 (torch.nn.functional.mish semantics).
    // tanh(log(1+e^x)) = ((1+e^x)^2 - 1) / ((1+e^x)^2 + 1) = w / (w + 2), w = e^x (e^x + 2)
    if (y > 20.0f) return y;
    const float n = expf(y);
    const float w = n * (n + 2.0f);
    return y * (w / (w + 2.0f));
}

// LDS floats needed by one block (host mirrors this to size the dynamic allocation).
__host__ __device__ inline int conv_xrow_stride(int BN, int Lin, int Lout, int taps) {
    const int spt = BN / Lout;
    const int seg = Lin + 2 * (taps / 2);
    return spt * seg + 4;   // +4: keeps rows 16-B aligned and breaks the 2-row bank pattern
}
__host__ __device__ inline size_t conv_lds_floats(int BM, int BN, int KC, int taps, int Lin,
                                                  int Lout) {
    const size_t stage = (size_t)taps * KC * BM + (size_t)KC * conv_xrow_stride(BN, Lin, Lout, taps);
    const size_t epi = (size_t)BM * (BN + 4) + 2 * 512;
    const size_t k = 2 * stage;
    return k > epi ? k : epi;
}

template <int BM, int BN, int WM, int WN, int KC, int TAPS, int STRIDE>
__global__ __launch_bounds__(64 * WM * WN) void conv_gemm_f32(const ConvParams p) {
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int PAD = TAPS / 2;
    static_assert(TM >= 1 && TN >= 1, "wave tile must be at least 32x32");
    static_assert(KC % 8 == 0, "KC is a multiple of the packing granule");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int l32 = lane & 31;
    const int khalf = lane >> 5;

    const int Lin = p.Lin, Lout = p.Lout;
    const int SPT = BN >> p.lshift;                 // whole samples per tile
    const int SEG = Lin + 2 * PAD;
    const int XS = SPT * SEG + 4;
    const int s0 = blockIdx.x * SPT;                // first sample of this tile
    const int m0 = blockIdx.y * BM;                 // first output row of this tile
    const int M = p.M;
    const int nvalid = min(SPT, p.B - s0);          // samples that exist

    constexpr int WFLOATS = TAPS * KC * BM;
    const int XFLOATS = KC * XS;
    float* Wl[2] = {smem, smem + WFLOATS + XFLOATS};
    float* Xl[2] = {smem + WFLOATS, smem + 2 * WFLOATS + XFLOATS};

    // zero both X stages once: the per-sample PAD columns (and rows/samples that do not
    // exist) stay zero for the whole K loop, staging only ever writes real positions.
    for (int i = tid; i < XFLOATS; i += NT) { Xl[0][i] = 0.0f; Xl[1][i] = 0.0f; }

    // per-lane B-operand column bases
    int colbase[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = wn * (BN / WN) + tn * 32 + l32;
        const int s = n >> p.lshift;
        const int l = n & (Lout - 1);
        colbase[tn] = s * SEG + l * STRIDE;
    }
    const int arow = wm * (BM / WM) + l32;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int cin = p.cin0 + p.cin1;
    const int nchunks = (p.cin_pad + KC - 1) / KC;
    const long NinTot = (long)p.B * Lin;

    // ---- staging helpers --------------------------------------------------------------
    constexpr int W_F4 = WFLOATS / 4;                       // float4 per W stage
    constexpr int W_PER_T = (W_F4 + NT - 1) / NT;
    constexpr int BM4 = BM / 4;
    const int xq = Lin >> 2;                                // float4 per sample row
    const int X_F4 = KC * SPT * xq;
    float4 wreg[W_PER_T];

    auto load_w = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < W_PER_T; ++i) {
            const int e = tid + i * NT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < W_F4) {
                const int row = e / BM4;                    // tap*KC + q*8 + ci8
                const int c4 = e - row * BM4;
                const int tap = row / KC;
                const int cl = row - tap * KC;
                const int g = chunk * (KC / 8) + (cl >> 3);
                if (g * 8 < p.cin_pad) {
                    const long grow = ((long)g * TAPS + tap) * 8 + (cl & 7);
                    v = *reinterpret_cast<const float4*>(p.w + grow * M + m0 + c4 * 4);
                }
            }
            wreg[i] = v;
        }
    };
    auto store_w = [&](float* dstW) {
#pragma unroll
        for (int i = 0; i < W_PER_T; ++i) {
            const int e = tid + i * NT;
            if (e < W_F4) *reinterpret_cast<float4*>(dstW + e * 4) = wreg[i];
        }
    };
    // X: straight global -> LDS (small: KC x BN*STRIDE floats); rows beyond cin stay zero.
    auto stage_x = [&](int chunk, float* dstX) {
        const int c0 = chunk * KC;
        if (p.src_nlc) {
            // external (B, Lin, cin0) layout: gather element-wise (first layer only)
            const int total = KC * SPT * Lin;
            for (int e = tid; e < total; e += NT) {
                const int row = e / (SPT * Lin);
                const int rem = e - row * (SPT * Lin);
                const int s = rem / Lin;
                const int l = rem - s * Lin;
                const int c = c0 + row;
                float v = 0.0f;
                if (c < cin && s < nvalid)
                    v = p.src0[((long)(s0 + s) * Lin + l) * p.cin0 + c];
                dstX[row * XS + s * SEG + PAD + l] = v;
            }
        } else {
            for (int e = tid; e < X_F4; e += NT) {
                const int row = e / (SPT * xq);
                const int rem = e - row * (SPT * xq);
                const int s = rem / xq;
                const int q = rem - s * xq;
                const int c = c0 + row;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < cin && s < nvalid) {
                    const float* base = (c < p.cin0) ? p.src0 + (long)c * NinTot
                                                     : p.src1 + (long)(c - p.cin0) * NinTot;
                    v = *reinterpret_cast<const float4*>(base + (long)(s0 + s) * Lin + q * 4);
                }
                float* d = dstX + row * XS + s * SEG + PAD + q * 4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        }
    };

    // ---- main loop: double-buffered over K chunks -------------------------------------
    __syncthreads();                       // zero fill visible before the first real rows
    load_w(0);
    store_w(Wl[0]);
    stage_x(0, Xl[0]);
    __syncthreads();

    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1;
        const bool more = (ch + 1) < nchunks;
        if (more) load_w(ch + 1);                        // global loads in flight under MFMA

        const float* Wc = Wl[cur];
        const float* Xc = Xl[cur];
#pragma unroll 2
        for (int kk = 0; kk < KC / 2; ++kk) {
            const int krow = kk * 2 + khalf;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = Wc[(tap * KC + krow) * BM + arow + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Xc[krow * XS + colbase[j] + tap];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        if (more) {
            store_w(Wl[cur ^ 1]);
            stage_x(ch + 1, Xl[cur ^ 1]);
        }
        __syncthreads();
    }

    // ---- epilogue through LDS ---------------------------------------------------------
    constexpr int ES = BN + 4;
    float* E = smem;                                   // [BM][ES]
    float* stat_mean = smem + BM * ES;                 // [<=512]
    float* stat_rstd = stat_mean + 512;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                const int col = wn * (BN / WN) + j * 32 + l32;
                E[row * ES + col] = acc[i][j][r] + p.bias[m0 + row];
            }
    __syncthreads();

    const bool has_gn = p.gamma != nullptr;
    if (has_gn) {
        // (group, sample) pairs of this tile; two-pass mean / biased variance.
        const int cpg = p.cpg;
        const int gpt = BM / cpg;                      // groups per tile
        const int pairs = gpt * SPT;
        const int cnt = cpg * Lout;
        int tpp = NT / pairs;                          // threads per pair (power of two)
        tpp = tpp < 1 ? 1 : (tpp > 64 ? 64 : tpp);
        const int ppr = NT / tpp;                      // pairs per round
        const int sub = tid & (tpp - 1);
        for (int pr = tid / tpp; pr < pairs; pr += ppr) {
            const int g = pr / SPT;
            const int s = pr - g * SPT;
            const float* base = E + (g * cpg) * ES + s * Lout;
            float sum = 0.0f;
            for (int e = sub; e < cnt; e += tpp) sum += base[(e >> p.lshift) * ES + (e & (Lout - 1))];
            for (int o = tpp >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            const float mean = sum / (float)cnt;
            float sq = 0.0f;
            for (int e = sub; e < cnt; e += tpp) {
                const float d = base[(e >> p.lshift) * ES + (e & (Lout - 1))] - mean;
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

    // normalise + Mish + adds + coalesced store (float4 along N)
    const long NoutTot = (long)p.B * Lout;
    constexpr int BN4 = BN / 4;
    for (int e = tid; e < BM * BN4; e += NT) {
        const int row = e / BN4;
        const int c4 = (e - row * BN4) * 4;
        const int s = c4 >> p.lshift;
        if (s >= nvalid) continue;
        const int m = m0 + row;
        float4 v = *reinterpret_cast<const float4*>(E + row * ES + c4);
        float y[4] = {v.x, v.y, v.z, v.w};
        if (has_gn) {
            const int pr = (row / p.cpg) * SPT + s;
            const float mu = stat_mean[pr], rs = stat_rstd[pr];
            const float ga = p.gamma[m], be = p.beta[m];
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = mish_f32((y[k] - mu) * rs * ga + be);
        }
        if (p.temb != nullptr) {
            const float tv = p.temb[m];
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] += tv;
        }
        const int l = c4 & (Lout - 1);
        if (!p.interleave) {
            const long off = (long)m * NoutTot + (long)(s0 + s) * Lout + l;
            if (p.res != nullptr) {
                const float4 r = *reinterpret_cast<const float4*>(p.res + off);
                y[0] += r.x; y[1] += r.y; y[2] += r.z; y[3] += r.w;
            }
            *reinterpret_cast<float4*>(p.dst + off) = make_float4(y[0], y[1], y[2], y[3]);
        } else {
            const int half = M >> 1;
            const int phase = m >= half;
            const int ch = m - phase * half;
            float* d = p.dst + (long)ch * (2 * NoutTot) + (long)(s0 + s) * (2 * Lout) + 2 * l + phase;
            d[0] = y[0]; d[2] = y[1]; d[4] = y[2]; d[6] = y[3];
        }
    }
}

}  // namespace dad
