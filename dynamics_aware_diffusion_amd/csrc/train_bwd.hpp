// train_bwd.hpp — the kernels of the backward pass that are not a convolution themselves.
//
// The reference trains with loss.backward() through TemporalUnet (m_diffuser/utils/training.py:144-156,
// m_diffuser/models/diffusion.py:253-290): autograd walks Conv1d / ConvTranspose1d / GroupNorm / Mish /
// Linear.  Here the data gradients of the convs run on the forward conv-GEMM kernel with transposed,
// tap-flipped weight images (dad_lib.hip, "backward plan"); this file holds the rest:
//
//   conv_wgrad<TAPS>   dW[m][c][k] = sum_{b,l} dH[b,l,m] * X[b, l*stride + k - pad, c]     (MFMA GEMM, K = B*L)
//   gn_mish_bwd        dH = d(conv + bias) of  Mish(GroupNorm(h)) (+ time embedding), per (sample, group),
//                      with the per-sample partial sums of d gamma, d beta, d bias and d(time projection)
//   row_partial_sums / col_sums / sum_slabs / add_inplace      deterministic reductions (fixed order, no atomics)
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
};

constexpr int WG_TILE = 64;                  // block tile: 64 (m) x 64 (c), 4 waves of 32 x 32 x TAPS
constexpr int WG_THREADS = 256;
constexpr int WG_ROWS = 64;                  // G rows per staged chunk: spc = max(1, 64 / Lg) whole samples

__host__ __device__ inline int wgrad_segz(int Lz, int taps, int pad) { return Lz + pad + (taps - 1 - pad); }
__host__ __device__ inline size_t wgrad_lds_floats(int spc, int Lg, int Lz, int taps, int pad) {
    return (size_t)spc * Lg * WG_TILE + (size_t)spc * wgrad_segz(Lz, taps, pad) * WG_TILE;
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

template <int TAPS>
__global__ __launch_bounds__(WG_THREADS) void conv_wgrad(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int l32 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * WG_TILE, c0 = blockIdx.y * WG_TILE;
    const int ks = blockIdx.z;
    const int Lg = p.Lg, Lz = p.Lz;
    const int SEGZ = wgrad_segz(Lz, TAPS, p.pad);
    const int spc = p.spc;
    float* const Gs = smem;                                  // [spc * Lg][64]
    float* const Zs = smem + spc * Lg * WG_TILE;             // [spc * SEGZ][64]
    const int s_lo = ks * p.samples_per_split;
    const int s_hi = min(p.B, s_lo + p.samples_per_split);
    const int Ctot = p.C0 + p.C1;
    const bool gvec = (p.ldg & 3) == 0, z0vec = (p.ldz0 & 3) == 0, z1vec = (p.ldz1 & 3) == 0;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // A chunk = spc whole samples: G rows [spc * Lg][64] and Z rows with their zero halo [spc * SEGZ][64].
    // The loads of chunk i + 1 are issued before the MFMA loop of chunk i and land under it (registers:
    // up to WG_GI + WG_ZI float4 per thread), then go to LDS behind the barrier.
    constexpr int WG_GI = 8, WG_ZI = 10;                     // covers 128 G rows / 160 Z rows per chunk
    const int n_g = spc * Lg * (WG_TILE / 4), n_z = spc * SEGZ * (WG_TILE / 4);
    float4 gr[WG_GI], zr[WG_ZI];
    auto chunk_load = [&](int sb) {
#pragma unroll
        for (int k = 0; k < WG_GI; ++k) {
            const int i = tid + k * WG_THREADS;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n_g) {
                const int row = i >> 4, q = i & 15;
                const int smp = sb + (row >> p.lg_shift), l = row & (Lg - 1);
                if (smp < s_hi) v = wg_load4(p.G + (long)(smp * Lg + l) * p.ldg, m0 + 4 * q, p.M, gvec);
            }
            gr[k] = v;
        }
#pragma unroll
        for (int k = 0; k < WG_ZI; ++k) {
            const int i = tid + k * WG_THREADS;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n_z) {
                const int row = i >> 4, q = i & 15;
                const int sl = row / SEGZ, pz = row - sl * SEGZ - p.pad;
                const int smp = sb + sl;
                const int c = c0 + 4 * q;
                if (smp < s_hi && pz >= 0 && pz < Lz && c < Ctot) {
                    if (c < p.C0) v = wg_load4(p.Z0 + (long)(smp * Lz + pz) * p.ldz0, c, p.C0, z0vec);
                    else v = wg_load4(p.Z1 + (long)(smp * Lz + pz) * p.ldz1, c - p.C0, p.C1, z1vec);
                }
            }
            zr[k] = v;
        }
    };
    chunk_load(s_lo);
    for (int sb = s_lo; sb < s_hi; sb += spc) {
        __syncthreads();                                     // the previous chunk's fragment reads are done
#pragma unroll
        for (int k = 0; k < WG_GI; ++k) {
            const int i = tid + k * WG_THREADS;
            if (i < n_g) *reinterpret_cast<float4*>(Gs + (i >> 4) * WG_TILE + 4 * (i & 15)) = gr[k];
        }
#pragma unroll
        for (int k = 0; k < WG_ZI; ++k) {
            const int i = tid + k * WG_THREADS;
            if (i < n_z) *reinterpret_cast<float4*>(Zs + (i >> 4) * WG_TILE + 4 * (i & 15)) = zr[k];
        }
        __syncthreads();
        if (sb + spc < s_hi) chunk_load(sb + spc);           // in flight under the MFMAs below
        // ---- K loop over the chunk's rows: k = row (lane half h takes row kk + h)
        const int nrows = spc * Lg;
        for (int kk = 0; kk < nrows; kk += 2) {
            const int row = kk + h;
            const int sl = row >> p.lg_shift, l = row & (Lg - 1);
            const float a = Gs[row * WG_TILE + wm * 32 + l32];
            const float* zb = Zs + (sl * SEGZ + l * p.stride) * WG_TILE + wn * 32 + l32;
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, zb[t * WG_TILE], acc[t], 0, 0, 0);
        }
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

// out[i] = sum_k slab[k][i], k in order
__global__ void sum_slabs_kernel(float* out, const float* slab, long n, int ks) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = slab[i];
    for (int k = 1; k < ks; ++k) v += slab[(long)k * n + i];
    out[i] = v;
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

// the same for three [B][C] arrays at once (blockIdx.y picks one): d gamma, d beta, d bias of one conv
struct ColSums3 { float* out[3]; const float* part[3]; };
__global__ __launch_bounds__(256) void col_sums3_kernel(const ColSums3 a, int B, int stride, int C) {
    __shared__ float red[8][33];
    const float* part = blockIdx.y == 0 ? a.part[0] : (blockIdx.y == 1 ? a.part[1] : a.part[2]);
    float* out = blockIdx.y == 0 ? a.out[0] : (blockIdx.y == 1 ? a.out[1] : a.out[2]);
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
    const float inv_n = 1.0f / (float)(cpg * L);

    // pass 1: the two pair sums
    float s1 = 0.0f, s2 = 0.0f;
    for (int l = lg; l < L; l += nlg) {
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
    for (int l = lg; l < L; l += nlg) {
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

}  // namespace dad
