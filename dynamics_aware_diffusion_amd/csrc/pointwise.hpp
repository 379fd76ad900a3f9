// pointwise.hpp — the bandwidth-bound kernels around the conv GEMMs:
//   * final 1x1 conv + reverse-diffusion posterior step in one pass
//       (temporal_unet.py:196 final_conv[1]; diffusion.py:159-223; policies.py:84-110)
//   * counter-based normal generator (Philox4x32-10 + Box-Muller) replacing torch.randn
//   * dynamics projection gather -> x@P -> blend -> scatter (policies.py:409-485)
//   * one-off time-embedding table builders (temporal_unet.py:19-32,97-100,155-160)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_gemm.hpp"

namespace dad {

// ------------------------------------------------------------------------------ Philox
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0;
        const uint64_t p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Standard normal for global element index `e` of draw `draw`:
//   quad = e >> 2 picks the Philox block, e & 3 picks one of its four Box-Muller outputs.
__device__ __forceinline__ float philox_normal(uint64_t e, uint64_t draw, uint64_t seed) {
    uint32_t r[4];
    const uint64_t q = e >> 2;
    philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)draw, (uint32_t)(draw >> 32),
                  (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const int pair = (int)((e >> 1) & 1);
    const uint32_t ra = r[2 * pair], rb = r[2 * pair + 1];
    const float u1 = ((float)(ra >> 9) + 0.5f) * (1.0f / 8388608.0f);   // (0,1), exact
    const float u2 = (float)(rb >> 8) * (1.0f / 16777216.0f);            // [0,1), exact
    const float rad = sqrtf(-2.0f * logf(u1));
    const float ang = 6.28318530717958647692f * u2;
    return (e & 1) ? rad * sinf(ang) : rad * cosf(ang);
}

__global__ void set_u64_kernel(unsigned long long* dst, unsigned long long v) { *dst = v; }

__global__ void fill_normal_kernel(float* x, long n_elems, uint64_t elem_offset, uint64_t draw,
                                   uint64_t seed) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_elems) x[i] = philox_normal(elem_offset + (uint64_t)i, draw, seed);
}

// dst[r][0:c0] = a[r][:], dst[r][c0:c0+c1] = b[r][:]  (float4 granularity; c0, c1 multiples of 4).
// torch.cat([x, skip], dim=1) made real — only for the identity residual of a decoder block whose
// input concat is as wide as its output; every other concat stays virtual (two source pointers).
__global__ void concat_rows_kernel(float* dst, const float* a, const float* b, long rows, int c0, int c1) {
    const int q = (c0 + c1) >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * q) return;
    const long r = i / q;
    const int c = (int)(i - r * q) * 4;
    const float4 v = c < c0 ? *reinterpret_cast<const float4*>(a + r * c0 + c)
                            : *reinterpret_cast<const float4*>(b + r * c1 + (c - c0));
    *reinterpret_cast<float4*>(dst + r * (c0 + c1) + c) = v;
}

// -------------------------------------------------------------- final conv + posterior
struct FinalParams {
    const float* act;        // [B*H][dim] channels-last output of final_conv[0]
    const float* w;          // [td][dim] final_conv[1].weight (k=1)
    const float* bias;       // [td]
    float* x;                // (B,H,td) trajectory, updated in place unless x_out_disabled
    const float* noise;      // (B,H,td) or nullptr -> Philox
    const float* cond0;      // inpainting for horizon step 0 or nullptr
    const float* guide;      // (B,H,td) guide gradient or nullptr
    float* mean_out;         // optional
    float* eps_out;          // optional
    int32_t dim, td, B, H;
    int32_t Hact;            // rows per sample of `act` (the zero-padded horizon; == H without padding)
    int32_t cond_per_row;
    int32_t predict_epsilon, clip_denoised;
    int32_t x_out_disabled;  // 1: only eps_out / mean_out are produced
    float c_recip, c_recipm1, coef1, coef2;   // schedule scalars at t (diffusion.py:164-178)
    float sigma;             // [t != 0] * exp(0.5 * log_var_t)
    float guide_scale;       // guide_weight * exp(log_var_t)      (policies.py:97)
    uint64_t seed, elem_offset, draw;
    const unsigned long long* seed_dev;   // when set, the Philox key is read from device memory
                                          // (graph replays take a fresh seed without re-capture)
};

constexpr int FINAL_COLS = 32;   // trajectory positions per block (256 blocks at B*H = 8192)
// a block keeps the weight rows of ITS output columns (gy = gridDim.y column groups), the biases and
// the activation tile
__host__ __device__ inline int final_rows_local(int td, int gy) {
    constexpr int JG = 256 / FINAL_COLS;
    const int col_groups = (td + JG - 1) / JG;
    return (col_groups + gy - 1) / gy * JG;
}
__host__ __device__ inline size_t final_lds_floats(int td, int dim, int gy) {
    return (size_t)final_rows_local(td, gy) * dim + ((td + 3) & ~3) + (size_t)FINAL_COLS * (dim + 4);
}

// K output columns j0, j0 + jstep, ... of one position: dot products of the activation row with K
// weight rows (local rows of the block, JG apart), every read of the activation row shared.
template <int K>
__device__ __forceinline__ void final_dots(const float* wrow0, const float* arow, int dim, int rstep, float* acc) {
    const float* wr[K];
    float a[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        wr[k] = wrow0 + (long)k * rstep;                 // the thread's next column: JG local rows further
        a[k][0] = a[k][1] = a[k][2] = a[k][3] = 0.0f;
    }
    for (int c = 0; c < dim; c += 4) {
        const float4 av = *reinterpret_cast<const float4*>(arow + c);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float4 wv = *reinterpret_cast<const float4*>(wr[k] + c);
            a[k][0] = fmaf(wv.x, av.x, a[k][0]); a[k][1] = fmaf(wv.y, av.y, a[k][1]);
            a[k][2] = fmaf(wv.z, av.z, a[k][2]); a[k][3] = fmaf(wv.w, av.w, a[k][3]);
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = (a[k][0] + a[k][1]) + (a[k][2] + a[k][3]);     // four chains, summed pairwise
}

// One block = 32 (b, l) positions.  The [32][dim] activation slab and the [td][dim] weights
// are staged in LDS with coalesced float4 loads (rows padded to dim+4 floats: 16-byte aligned and
// an odd number of 16-byte slots, conflict-free ds_read_b128); thread (col, jg) then produces
// outputs j = jg, jg+8, ... of its position and applies the posterior update to them.
__global__ __launch_bounds__(256) void final_posterior_kernel(const FinalParams p) {
    extern __shared__ __attribute__((aligned(16))) float ws[];
    const int td = p.td, dim = p.dim;
    const int rs = dim + 4;                // tile row stride
    constexpr int JG = 256 / FINAL_COLS;
    const int rows_local = final_rows_local(td, (int)gridDim.y);
    float* wl = ws;                        // [rows_local][dim]: column j of this block at row (j / (JG * gy)) * JG + j % JG
    float* bl = wl + rows_local * dim;     // [td] (+ pad to 4)
    float* tile = bl + ((td + 3) & ~3);    // [32][dim+4]
    const long N = (long)p.B * p.H;
    const long n0 = (long)blockIdx.x * FINAL_COLS;
    const int dq = dim >> 2;
    for (int e = threadIdx.x; e < FINAL_COLS * dq; e += blockDim.x) {
        const int row = e / dq, q = e - row * dq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0 + row < N) {
            long ar = n0 + row;                      // row of the activation buffer: sample * Hact + position
            if (p.Hact != p.H) { const long sb = ar / p.H; ar = sb * p.Hact + (ar - sb * p.H); }
            v = *reinterpret_cast<const float4*>(p.act + ar * dim + q * 4);
        }
        *reinterpret_cast<float4*>(tile + row * rs + q * 4) = v;
    }
    // blockIdx.y owns the output columns j with (j / JG) % gridDim.y == blockIdx.y: wide transitions
    // (and short grids) spread their columns over more blocks; each block stages only its own rows
    for (int i = threadIdx.x; i < rows_local * dq; i += blockDim.x) {
        const int lr = i / dq, q = i - lr * dq;
        const int j = ((lr / JG) * (int)gridDim.y + (int)blockIdx.y) * JG + lr % JG;     // global column of local row lr
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < td) v = *reinterpret_cast<const float4*>(p.w + (long)j * dim + q * 4);
        *reinterpret_cast<float4*>(wl + i * 4) = v;
    }
    for (int i = threadIdx.x; i < td; i += blockDim.x) bl[i] = p.bias[i];
    __syncthreads();

    const int col = threadIdx.x & (FINAL_COLS - 1);
    const int jg = threadIdx.x / FINAL_COLS;
    const long n = n0 + col;
    if (n >= N) return;
    const int b = (int)(n / p.H);
    const int l = (int)(n - (long)b * p.H);
    const float* arow = tile + col * rs;
    // posterior update of one output (diffusion.py:159-223, policies.py:84-110)
    auto finish = [&](int j, float acc, float xv) {
        const long idx = n * td + j;
        const float out = acc + bl[j];
        if (p.eps_out != nullptr) p.eps_out[idx] = out;
        if (p.x_out_disabled && p.mean_out == nullptr) return;
        float x0 = p.predict_epsilon ? p.c_recip * xv - p.c_recipm1 * out : out;
        if (p.clip_denoised) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        float mean = p.coef1 * x0 + p.coef2 * xv;
        if (p.guide != nullptr) mean = mean + p.guide_scale * p.guide[idx];
        if (p.mean_out != nullptr) p.mean_out[idx] = mean;
        if (p.x_out_disabled) return;
        const float z = (p.noise != nullptr)
                            ? p.noise[idx]
                            : philox_normal(p.elem_offset + (uint64_t)idx, p.draw,
                                            p.seed_dev != nullptr ? (uint64_t)*p.seed_dev : p.seed);
        float xn = mean + p.sigma * z;
        if (l == 0 && p.cond0 != nullptr) xn = p.cond0[(p.cond_per_row ? (long)b * td : 0) + j];
        p.x[idx] = xn;
    };
    // up to JB of the thread's columns share every read of its activation row (the weight rows are
    // wave-wide broadcasts: the activation row is the LDS traffic); per output the same four chains.
    // The block-uniform count of live columns picks the instantiation (PointMaze: one).
    constexpr int JB = 4;
    const int jstep = JG * (int)gridDim.y;
    for (int jb = (int)blockIdx.y * JG; jb < td; jb += JB * jstep) {
        const int j0 = jb + jg;
        const int live = min(JB, (td - 1 - jb) / jstep + 1);      // columns jb + k * jstep < td
        float acc[JB], xv[JB];
        // x_t of the thread's outputs: loaded ahead of the dot products (each element of x is read and
        // written by exactly one thread of the grid)
#pragma unroll
        for (int k = 0; k < JB; ++k) xv[k] = p.x[n * td + min(j0 + k * jstep, td - 1)];
        const float* wrow0 = wl + (long)(((jb / JG) / (int)gridDim.y) * JG + jg) * dim;   // local row of column j0
        switch (live) {
            case 1: final_dots<1>(wrow0, arow, dim, JG * dim, acc); break;
            case 2: final_dots<2>(wrow0, arow, dim, JG * dim, acc); break;
            case 3: final_dots<3>(wrow0, arow, dim, JG * dim, acc); break;
            default: final_dots<4>(wrow0, arow, dim, JG * dim, acc); break;
        }
#pragma unroll
        for (int k = 0; k < JB; ++k)
            if (k < live && j0 + k * jstep < td) finish(j0 + k * jstep, acc[k], xv[k]);
    }
}

// -------------------------------------------------------------------------- projection
struct ProjParams {
    const float* P;          // [D][D]
    const float* obs_mean; const float* obs_std; const float* act_mean; const float* act_std;
    float* x;                // (B,H,od+m) in place
    float* xout;             // project_gemm_kernel: where the projected trajectories go (a scratch copy:
                             // other blocks still gather from x); the caller copies it back over x
    int32_t B, H, n, od, m, D;
    float alpha, one_minus_alpha;
    float* violation;        // when set: x is left untouched and violation[b] = sum_d (v - vP)_d^2 in
                             // physical units (losses/__init__.py:161-186, ProjectionLoss.compute)
};

// One block = RB trajectories, KP waves.  v@P is latency-bound (P = D*D*4 bytes streams through
// every CU from L2, one dependent-free load per FMA), so the work is spread as wide as it goes:
// lane dl of wave kp accumulates column d = 64 s + dl over the kp-th slice of k; the KP slices
// meet in LDS and are added in fixed order.  (4 waves: 11.8 us per call at B=256, D=196.)
template <int RB, int KP>
__global__ __launch_bounds__(64 * KP) void project_kernel(const ProjParams p) {
    extern __shared__ __attribute__((aligned(16))) float v[];    // [RB][D] rows, then [KP][RB][D] partials
    const int D = p.D, H = p.H, n = p.n, m = p.m, td = p.od + p.m;
    float* part = v + RB * D;
    const int b0 = blockIdx.x * RB;
    const int nstate = (H + 1) * n;
    // gather + de-normalise: [s_0..s_{H-1}, s_{H-1}, a_0..a_{H-1}]  (policies.py:434-448)
    for (int e = threadIdx.x; e < RB * D; e += blockDim.x) {
        const int r = e / D, d = e - r * D;
        const int b = b0 + r;
        float val = 0.0f;
        if (b < p.B) {
            const float* xb = p.x + (long)b * H * td;
            if (d < nstate) {
                const int ts = d / n, k = d - ts * n;
                const int tsrc = ts < H ? ts : H - 1;
                val = xb[tsrc * td + k] * p.obs_std[k] + p.obs_mean[k];
            } else {
                const int dd = d - nstate;
                const int ts = dd / m, k = dd - ts * m;
                val = xb[ts * td + p.od + k] * p.act_std[k] + p.act_mean[k];
            }
        }
        v[e] = val;
    }
    __syncthreads();
    const int dl = threadIdx.x & 63, kp = threadIdx.x >> 6;
    const int kq = (D + KP - 1) / KP;
    const int k0 = kp * kq, k1 = min(D, k0 + kq);
    for (int d = dl; d < D; d += 64) {
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = 0.0f;
        const float* pc = p.P + d;
#pragma unroll 8
        for (int k = k0; k < k1; ++k) {
            const float pk = pc[(long)k * D];
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[r] = fmaf(v[r * D + k], pk, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) part[(kp * RB + r) * D + d] = acc[r];
    }
    __syncthreads();
    if (p.violation != nullptr) {
        // ProjectionLoss: squared distance from the dynamics-consistent subspace, summed per row in a
        // fixed order (wave 0, strided partials then a shuffle tree): deterministic
        if (threadIdx.x < 64) {
            for (int r = 0; r < RB; ++r) {
                float acc = 0.0f;
                for (int d = threadIdx.x; d < D; d += 64) {
                    float proj = 0.0f;
#pragma unroll
                    for (int q = 0; q < KP; ++q) proj += part[(q * RB + r) * D + d];
                    const float diff = v[r * D + d] - proj;
                    acc = fmaf(diff, diff, acc);
                }
                for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
                if (threadIdx.x == 0 && b0 + r < p.B) p.violation[b0 + r] = acc;
            }
        }
        return;
    }
    for (int e = threadIdx.x; e < RB * D; e += blockDim.x) {
        const int r = e / D, d = e - r * D;
        const int b = b0 + r;
        if (b >= p.B) continue;
        float proj = 0.0f;
#pragma unroll
        for (int q = 0; q < KP; ++q) proj += part[(q * RB + r) * D + d];
        const float blended = p.alpha * proj + p.one_minus_alpha * v[r * D + d];
        float* xb = p.x + (long)b * H * td;
        if (d < nstate) {
            const int ts = d / n, k = d - ts * n;
            if (ts < H) xb[ts * td + k] = (blended - p.obs_mean[k]) / p.obs_std[k];
        } else {
            const int dd = d - nstate;
            const int ts = dd / m, k = dd - ts * m;
            xb[ts * td + p.od + k] = (blended - p.act_mean[k]) / p.act_std[k];
        }
    }
    // observation channels beyond the physical state are zero-padded (policies.py:475-480)
    if (p.od > n) {
        const int extra = p.od - n;
        for (int e = threadIdx.x; e < RB * H * extra; e += blockDim.x) {
            const int r = e / (H * extra);
            const int rem = e - r * H * extra;
            const int ts = rem / extra, k = rem - ts * extra;
            const int b = b0 + r;
            if (b < p.B) p.x[((long)b * H + ts) * td + n + k] = 0.0f;
        }
    }
}

// The same projection as a GEMM (batches of 32+ trajectories, any D): C[b][d] = sum_k v[b][k] P[k][d] on
// v_mfma_f32_32x32x2_f32.  One block = 32 trajectories x 32 columns of P; its four waves split K in
// chunks of 64 and meet in LDS (fixed order).  P is read once per 32 trajectories (straight from L2 /
// Infinity Cache into the B operand: lane (column, k half) reads P[k][d0 + column], 128 contiguous bytes
// per half wave) instead of once per trajectory; v is gathered + de-normalised from x chunk by chunk into
// a wave-private LDS tile.  The epilogue blends and scatters its 32 x 32 outputs
// (guides/policies.py:451-483) into a scratch copy of the batch (other blocks are still gathering from
// x); the host copies it back.  No atomics: bit-reproducible.
constexpr int PG_THREADS = 256;
constexpr int PG_KC = 64;                     // k values per staged chunk
constexpr int PG_AS = 33;                     // LDS row stride of the staged v chunk [k][32 rows]
__host__ __device__ inline size_t project_gemm_lds_floats() { return (size_t)4 * PG_KC * PG_AS + 64; }

// element d of trajectory row `xb` in physical units (policies.py:434-448)
__device__ __forceinline__ float proj_gather(const ProjParams& p, const float* xb, int d, int nstate, int td) {
    if (d < nstate) {
        const int ts = d / p.n, k = d - ts * p.n;
        const int tsrc = ts < p.H ? ts : p.H - 1;
        return xb[tsrc * td + k] * p.obs_std[k] + p.obs_mean[k];
    }
    const int dd = d - nstate;
    const int ts = dd / p.m, k = dd - ts * p.m;
    return xb[ts * td + p.od + k] * p.act_std[k] + p.act_mean[k];
}

__global__ __launch_bounds__(PG_THREADS) void project_gemm_kernel(const ProjParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, h = lane >> 5;
    const int D = p.D, H = p.H, td = p.od + p.m;
    const int nstate = (H + 1) * p.n;
    const int b0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    float* const As = sm + wave * (PG_KC * PG_AS);          // this wave's [64 k][32 rows (+1)]
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int nchunks = (D + PG_KC - 1) / PG_KC;
    const int dcol = min(d0 + l32, D - 1);                  // (columns past D read column D-1 and are dropped)
    for (int ch = wave; ch < nchunks; ch += 4) {
        const int k0 = ch * PG_KC;
        // B operand: 32 k-pairs of this chunk, all loads in flight before the gather
        float bv[PG_KC / 2];
#pragma unroll
        for (int j = 0; j < PG_KC / 2; ++j) {
            const int k = min(k0 + 2 * j + h, D - 1);       // (rows past D: masked by the zero A operand)
            bv[j] = p.P[(long)k * D + dcol];
        }
        // A operand: lane = k of the chunk, loop over the 32 trajectories
        {
            const int k = k0 + lane;
            for (int r = 0; r < 32; ++r) {
                const int b = b0 + r;
                float v = 0.0f;
                if (k < D && b < p.B) v = proj_gather(p, p.x + (long)b * H * td, k, nstate, td);
                As[lane * PG_AS + r] = v;
            }
        }
        // (wave-private tile: the wave's own LDS writes are ordered before its reads by the waitcnt the
        // compiler inserts; no block barrier inside the K loop)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int j = 0; j < PG_KC / 2; ++j)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(2 * j + h) * PG_AS + l32], bv[j], acc, 0, 0, 0);
    }
    __syncthreads();
    float* const E = sm;                                    // [4 waves][32 rows][33]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        E[(wave * 32 + row) * PG_AS + l32] = acc[r];
    }
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += PG_THREADS) {
        const int r = e >> 5, cidx = e & 31;
        const int b = b0 + r, d = d0 + cidx;
        if (b >= p.B || d >= D) continue;
        const float proj = ((E[r * PG_AS + cidx] + E[(32 + r) * PG_AS + cidx]) + E[(64 + r) * PG_AS + cidx]) +
                           E[(96 + r) * PG_AS + cidx];
        const float vv = proj_gather(p, p.x + (long)b * H * td, d, nstate, td);
        const float blended = p.alpha * proj + p.one_minus_alpha * vv;
        float* xb = p.xout + (long)b * H * td;
        if (d < nstate) {
            const int ts = d / p.n, k = d - ts * p.n;
            if (ts < H) xb[ts * td + k] = (blended - p.obs_mean[k]) / p.obs_std[k];
        } else {
            const int dd = d - nstate;
            const int ts = dd / p.m, k = dd - ts * p.m;
            xb[ts * td + p.od + k] = (blended - p.act_mean[k]) / p.act_std[k];
        }
    }
    // observation channels beyond the physical state are zero-padded (policies.py:475-480)
    if (p.od > p.n && blockIdx.y == 0) {
        const int extra = p.od - p.n;
        for (int e = tid; e < 32 * H * extra; e += PG_THREADS) {
            const int r = e / (H * extra);
            const int rem = e - r * H * extra;
            const int ts = rem / extra, k = rem - ts * extra;
            if (b0 + r < p.B) p.xout[((long)(b0 + r) * H + ts) * td + p.n + k] = 0.0f;
        }
    }
}

// y = Mish(x) through the conv epilogue's device function (mish_fast_f32) — dad_debug_mish: lets a
// test push torch.nn.Mish's golden grid (|x| > 20, -100, ...) through the arithmetic the fused
// epilogue runs, without having to construct a network around it.
__global__ void mish_probe_kernel(const float* in, float* out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mish_fast_f32(in[i]);
}

// ------------------------------------------------------------- time-embedding tables
// out[t][m] = b[m] + sum_k W[m][k] * f(in[t][k]),  f = Mish when mish_in (the reference's
// nn.Sequential(Mish, Linear) / Linear -> Mish -> Linear chains), one thread per output.
__global__ void table_linear_kernel(const float* in, const float* W, const float* b, float* out,
                                    int T, int K, int M, int out_stride, int mish_in) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)T * M) return;
    const int t = (int)(i / M), m = (int)(i - (long)t * M);
    const float* x = in + (long)t * K;
    const float* w = W + (long)m * K;
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float v = mish_in ? mish_f32(x[k]) : x[k];
        acc = fmaf(w[k], v, acc);
    }
    out[(long)t * out_stride + m] = acc + b[m];
}

}  // namespace dad
