// dad_lib.hip — host side of libdad_hip.so: model state, weight packing, launch plan and
// the C ABI declared in include/dad.h.  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <mutex>
#include <set>
#include <tuple>

#include "../../include/dad.h"
#include "host_plan.hpp"
#include "conv_gemm.hpp"
#include "pointwise.hpp"
#include "conv_cc.hpp"
#include "conv_ccw.hpp"
#include "train_bwd.hpp"
#include "conv_chain.hpp"

using namespace dadhost;

namespace {

#ifdef DAD_STAMPS
unsigned long long* g_stamps = nullptr;
#endif

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(DAD_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                            \
    } while (0)

// Everything a captured loop freezes: pointers, sizes and the scalars baked into its launches.
struct GraphKey {
    const void* x; const void* noise; const void* cond; const void* ws;
    const void* P; const void* obs_mean; const void* obs_std; const void* act_mean; const void* act_std;
    const void* proj_scratch;
    int n_steps, batch, cond_per_row, state_dim, observation_dim, action_dim;
    int force_tile, flags;    // tile / split-K / fusion hooks change the captured launches
    uint64_t row_offset;
    uint64_t alpha_hash;      // projection strengths are baked into the captured launches
    bool operator<(const GraphKey& o) const {
        return std::memcmp(this, &o, sizeof(GraphKey)) < 0;
    }
};

}  // namespace

struct dad_model : HostModel {
    bool finalized = false;
    int device = -1;                                           // device the parameters live on
    std::vector<float> sched[5];                               // host schedule scalars
    bool have_sched = false;
    std::vector<float> emb_override;                           // dad_model_load_time_embedding
    // device
    float* d_emb = nullptr;           // [T][dim]      SinusoidalPosEmb
    float* d_temb = nullptr;          // [T][time_dim] time_mlp output
    float* d_temb_table = nullptr;    // [T][temb_width] every block's Mish -> Linear
    float* d_final_w = nullptr;       // [td][dim]
    float* d_final_b = nullptr;
    uint64_t* d_rng = nullptr;
    unsigned* d_counters = nullptr;   // split-K arrival tickets (zero between launches)
    float* d_zero = nullptr;          // zeros: bias row of the data-gradient launches (training)
    std::vector<dad::ChainConv> chain;       // level-0 chain (conv_chain.hpp): its five convs, or empty
    std::vector<int> chain_src;              // plan index of each chain conv (its standard image feeds chain_repack_kernel)
    int chain_C = 0, chain_last = -1;
    std::map<std::string, float*> d_time;    // time-MLP tensors as uploaded (dad_model_refresh_weights re-derives the tables)
    float* d_h1 = nullptr;            // [T][4 time_dim] scratch of the table builder
    std::vector<void*> owned;         // every hipMalloc to free
    void* d_repack = nullptr;         // dad_model_refresh_weights: descriptor table of the last refresh (device)
    size_t repack_cap = 0;
    std::vector<char> repack_host;    //   and its host copy (re-uploaded only when it changes)
    bool tables_stale = false;        // time-MLP tensors changed since the per-timestep tables were built
    // All parameters, tables and flags live in ONE device allocation: a conv launch touches a
    // handful of pages instead of one page per tensor (cold address translations used to cost
    // ~1 us at the start of every kernel).
    char* arena = nullptr;
    size_t arena_cap = 0, arena_used = 0;
    // profiling
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    double prof_flops = 0;
    int64_t prof_launches = 0;
    // graphs
    std::map<GraphKey, hipGraphExec_t> graphs;
    hipStream_t cap_stream = nullptr;
};

namespace {

using dad::ConvParams;

int arena_alloc(dad_model* m, size_t bytes, void** out) {
    const size_t aligned = (bytes + 255) / 256 * 256;
    if (m->arena_used + aligned > m->arena_cap)
        return fail(DAD_E_STATE, "parameter arena exhausted (%zu + %zu > %zu)", m->arena_used, aligned,
                    m->arena_cap);
    *out = m->arena + m->arena_used;
    m->arena_used += aligned;
    return DAD_OK;
}

int upload(dad_model* m, const std::vector<float>& host, float** dev) {
    void* p = nullptr;
    const int rc = arena_alloc(m, std::max<size_t>(host.size(), 1) * sizeof(float), &p);
    if (rc != DAD_OK) return rc;
    HIP_TRY(hipMemcpy(p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    *dev = (float*)p;
    return DAD_OK;
}

void free_device(dad_model* m) {
    for (auto& kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
    m->graphs.clear();
    for (void* p : m->owned) (void)hipFree(p);
    m->owned.clear();
    m->d_repack = nullptr; m->repack_cap = 0; m->repack_host.clear(); m->tables_stale = false;
    m->arena = nullptr;
    m->arena_cap = m->arena_used = 0;
    m->d_emb = m->d_temb = m->d_temb_table = nullptr;
    m->d_final_w = m->d_final_b = nullptr;
    m->d_rng = nullptr;
    m->d_counters = nullptr;
    m->d_zero = nullptr;
    m->chain.clear(); m->chain_src.clear(); m->chain_C = 0; m->chain_last = -1;
    for (Plan* plan : {&m->plan, &m->tplan})
        for (auto& op : plan->convs) op.d_w = op.d_bias = op.d_gamma = op.d_beta = op.d_rbias = nullptr;
    for (auto& b : m->bconvs) for (int k = 0; k < b.n; ++k) b.op[k].d_w = b.op[k].d_bias = nullptr;
    m->bfinal.d_w = m->bfinal.d_bias = nullptr;
}

// ---------------------------------------------------------------------- kernel registry
// Every conv-GEMM instantiation the planner can ask for, keyed by what plan_launch decides.
using KernFn = void (*)(const ConvParams);
using KernKey = std::tuple<int, int, int, bool, bool, bool, bool, bool>;   // cfg, taps, stride, x3, bdir, ragged, res, padded
using KernTable = std::map<KernKey, KernFn>;

template <int CFG> struct Tile {
    static constexpr int BM = kTiles[CFG].BM, BN = kTiles[CFG].BN, SK = kTiles[CFG].SK, KC = kTiles[CFG].KC;
};

template <int CFG, int TAPS, int STRIDE, bool X3, bool BDIR, bool RES, bool PADDED = false>
void reg_kernel(KernTable& t) {
    using T = Tile<CFG>;
    constexpr int KC = eff_kc(T::KC, T::BM, TAPS, T::SK, X3, BDIR, T::BN);
    t[KernKey(CFG, TAPS, STRIDE, X3, BDIR, false, RES, PADDED)] =
        dad::conv_gemm_f32<T::BM, T::BN, T::SK, KC, TAPS, STRIDE, false, X3, BDIR, RES, PADDED>;
    if constexpr (!BDIR && STRIDE == 1 && (TAPS & 1) == 1)             // general staging path
        t[KernKey(CFG, TAPS, STRIDE, X3, BDIR, true, RES, PADDED)] =
            dad::conv_gemm_f32<T::BM, T::BN, T::SK, KC, TAPS, STRIDE, true, X3, BDIR, RES, PADDED>;
}
// Zero-padded nets (dad_model_set_horizon / dad_model_set_group_channels): fp32, no ride, on the tiles the heuristic
// picks (kPaddedTiles of host_plan.hpp)
template <int CFG>
void reg_tile_padded(KernTable& t) {
    reg_kernel<CFG, 5, 1, false, false, false, true>(t);
    reg_kernel<CFG, 3, 1, false, false, false, true>(t);
    reg_kernel<CFG, 7, 1, false, false, false, true>(t);
    reg_kernel<CFG, 3, 2, false, false, false, true>(t);
    if constexpr (Tile<CFG>::KC >= 16)                      // backward of Upsample1d
        reg_kernel<CFG, 5, 2, false, false, false, true>(t);
    reg_kernel<CFG, 2, 1, false, false, false, true>(t);
    reg_kernel<CFG, 1, 1, false, false, false, true>(t);
    if constexpr (Tile<CFG>::KC < 16) {
        reg_kernel<CFG, 5, 1, false, true, false, true>(t);
        reg_kernel<CFG, 3, 1, false, true, false, true>(t);
        reg_kernel<CFG, 7, 1, false, true, false, true>(t);
    }
}
template <int CFG>
void reg_tile(KernTable& t) {
    reg_kernel<CFG, 5, 1, false, false, false>(t);
    reg_kernel<CFG, 3, 1, false, false, false>(t);      // TemporalUnet(kernel_size=3 / 7) (temporal_unet.py:139): fp32 only
    reg_kernel<CFG, 7, 1, false, false, false>(t);
    reg_kernel<CFG, 3, 2, false, false, false>(t);
    if constexpr (Tile<CFG>::KC >= 16)                      // backward of Upsample1d: 5-tap stride-2 conv
        reg_kernel<CFG, 5, 2, false, false, false>(t);
    reg_kernel<CFG, 2, 1, false, false, false>(t);
    reg_kernel<CFG, 1, 1, false, false, false>(t);
    if constexpr (Tile<CFG>::KC >= 16) {
        reg_kernel<CFG, 5, 1, false, false, true>(t);      // + the riding 1x1 residual conv
        reg_kernel<CFG, 3, 1, false, false, true>(t);
        reg_kernel<CFG, 7, 1, false, false, true>(t);
        reg_kernel<CFG, 5, 1, true, false, false>(t);      // split-f16 variants (16-channel granules)
        reg_kernel<CFG, 3, 2, true, false, false>(t);
        reg_kernel<CFG, 2, 1, true, false, false>(t);
        reg_kernel<CFG, 1, 1, true, false, false>(t);
    } else {                                               // wide tile: direct-B kernels of the GroupNorm'd 5-tap convs
        reg_kernel<CFG, 5, 1, true, true, false>(t);
        reg_kernel<CFG, 5, 1, false, true, false>(t);
        reg_kernel<CFG, 3, 1, true, true, false>(t);       // kernel_size 3 / 7 (an LDS weight stage of 7 taps x 256 rows
        reg_kernel<CFG, 3, 1, false, true, false>(t);      //   would not fit twice)
        reg_kernel<CFG, 7, 1, true, true, false>(t);
        reg_kernel<CFG, 7, 1, false, true, false>(t);
    }
}
const KernTable& kernel_table() {
    static const KernTable table = [] {
        KernTable t;
        reg_tile<0>(t); reg_tile<1>(t); reg_tile<2>(t); reg_tile<3>(t);
        reg_tile<4>(t); reg_tile<5>(t); reg_tile<6>(t); reg_tile<7>(t);
        reg_tile<8>(t); reg_tile<9>(t);
        reg_tile_padded<0>(t); reg_tile_padded<1>(t); reg_tile_padded<2>(t); reg_tile_padded<3>(t);
        reg_tile_padded<4>(t); reg_tile_padded<8>(t); reg_tile_padded<9>(t);
        return t;
    }();
    return table;
}

// Small-batch conv kernels (conv_cc.hpp) by (taps, riding 1x1 conv, an input with 9..16 partial
// slabs, rows per tile).
template <bool BIG, int NR>
const void* cc_kernel_t(int taps, bool ride) {
    if (taps == 5) return ride ? (const void*)dad::conv_cc<5, 1, true, BIG, NR> : (const void*)dad::conv_cc<5, 1, false, BIG, NR>;
    if (ride) return nullptr;
    if (taps == 3) return (const void*)dad::conv_cc<3, 2, false, BIG, NR>;
    if (taps == 2) return (const void*)dad::conv_cc<2, 1, false, BIG, NR>;
    return nullptr;
}
const void* cc_kernel(int taps, bool ride, bool big, int rows, bool windowed = false) {
    if (windowed) {
        if (taps != 5 || big || rows != 32) return nullptr;
        return ride ? (const void*)dad::conv_cc<5, 1, true, false, 32, 6, true> : (const void*)dad::conv_cc<5, 1, false, false, 32, 6, true>;
    }
    if (rows == 16) return big ? cc_kernel_t<true, 16>(taps, ride) : cc_kernel_t<false, 16>(taps, ride);
    return big ? cc_kernel_t<true, 32>(taps, ride) : cc_kernel_t<false, 32>(taps, ride);
}

// conv_ccw.hpp instantiations: (taps, this launch carries a riding 1x1 conv, an input has ride slabs,
// rows per tile)
template <bool RIDE, int NR>
const void* ccw_kernel_t(int taps, bool res) {
    if (taps == 5) return res ? (const void*)dad::conv_ccw<5, 1, true, RIDE, NR> : (const void*)dad::conv_ccw<5, 1, false, RIDE, NR>;
    if (res) return nullptr;
    if (taps == 3) return (const void*)dad::conv_ccw<3, 2, false, RIDE, NR>;
    if (taps == 2) return (const void*)dad::conv_ccw<2, 1, false, RIDE, NR>;
    if (taps == 1) return (const void*)dad::conv_ccw<1, 1, false, RIDE, NR>;
    return nullptr;
}
const void* ccw_kernel(int taps, bool res, bool ride_in, int rows) {
    if (rows == 16) return ride_in ? ccw_kernel_t<true, 16>(taps, res) : ccw_kernel_t<false, 16>(taps, res);
    return ride_in ? ccw_kernel_t<true, 32>(taps, res) : ccw_kernel_t<false, 32>(taps, res);
}

// Every kernel may use up to the full 160 KiB of LDS; the dynamic-LDS limit is a per-device
// function attribute, raised once per device (not lazily per launch, so that nothing but launches
// happens under hipGraph capture).
// conv_wgrad instantiations by (taps, block tile): tile 0 = 64 x 64 (two K-groups), 1 = 64 x 32 (four), 2 = 32 x 32
// (eight); tile 3 = 32 x 32 with the general staging path (operands that are not whole aligned float4 rows)
constexpr int kWgradTiles = 4;
template <int TAPS>
const void* wgrad_kernel_t(int tile) {
    return tile == 0 ? (const void*)dad::conv_wgrad<TAPS, 2, 2, true>
         : tile == 1 ? (const void*)dad::conv_wgrad<TAPS, 2, 1, true>
         : tile == 2 ? (const void*)dad::conv_wgrad<TAPS, 1, 1, true>
                     : (const void*)dad::conv_wgrad<TAPS, 1, 1, false>;
}
const void* wgrad_kernel(int taps, int tile) {
    switch (taps) {
        case 1: return wgrad_kernel_t<1>(tile);
        case 3: return wgrad_kernel_t<3>(tile);
        case 4: return wgrad_kernel_t<4>(tile);
        case 5: return wgrad_kernel_t<5>(tile);
        case 7: return wgrad_kernel_t<7>(tile);
    }
    return nullptr;
}

int configure_kernels() {
    static std::mutex lock;
    static std::set<int> done;
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> hold(lock);
    if (done.count(dev)) return DAD_OK;
    for (const auto& kv : kernel_table())
        HIP_TRY(hipFuncSetAttribute((const void*)kv.second, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)dad::kLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)dad::final_posterior_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)dad::kLdsBytes));
    for (int taps : {5, 3, 2})
        for (int ride = 0; ride < 2; ++ride)
            for (int big = 0; big < 2; ++big)
                for (int rows : {16, 32})
                    if (const void* k = cc_kernel(taps, ride != 0, big != 0, rows))
                        HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dad::kLdsBytes));
    for (int ride = 0; ride < 2; ++ride)
        HIP_TRY(hipFuncSetAttribute(cc_kernel(5, ride != 0, false, 32, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dad::kLdsBytes));
    for (int taps : {5, 3, 2, 1})
        for (int res = 0; res < 2; ++res)
            for (int ride = 0; ride < 2; ++ride)
                for (int rows : {16, 32})
                    if (const void* k = ccw_kernel(taps, res != 0, ride != 0, rows))
                        HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dad::kLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)dad::final_cc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)dad::kLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)dad::chain_l0_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dad::kLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)dad::chain_l0_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dad::kLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)dad::chain_l0_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dad::kLdsBytes));
    for (int taps : {1, 3, 4, 5, 7})
        for (int tile = 0; tile < kWgradTiles; ++tile)
            HIP_TRY(hipFuncSetAttribute(wgrad_kernel(taps, tile), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dad::kLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)dad::project_kernel<4, 16>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)dad::kLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)dad::project_kernel<1, 16>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)dad::kLdsBytes));
    done.insert(dev);
    return DAD_OK;
}

// Operands of one conv-GEMM launch as raw pointers (the forward plans name buffers; the backward walk
// passes gradient tensors).
struct ConvIO {
    const float* src0 = nullptr; const float* src1 = nullptr;
    float* dst = nullptr; const float* res = nullptr; float* rdst = nullptr;
    const float* temb = nullptr; const int32_t* trow = nullptr;
    float* pre = nullptr; float* stats = nullptr;      // training forward: pre-normalisation output, pair statistics
    float* slab = nullptr;                             // split-K scratch
};
int launch_conv(dad_model* m, const ConvOp& op, int batch, const ConvIO& io, hipStream_t st);

// `plan`: m->plan (sampling: buffers shared by lifetime) or m->tplan (training: every tensor kept, plus
// the pre-activation / statistics buffers of the GroupNorm'd convs).  `temb_rows`: (B, temb_width) time
// projections of the training forward (indexed through trow), instead of the per-timestep table.
int run_conv(dad_model* m, const Plan& plan, const ConvOp& op, const float* xext, float* ws, int batch, int t,
             hipStream_t st, const int32_t* trow = nullptr, const float* temb_rows = nullptr) {
    auto buf = [&](int id) -> float* {
        return id >= 0 ? ws + plan.bufs[id].offset * (long)batch : nullptr;
    };
    if (op.cat0 >= 0) {     // rows of [cat0 | cat1] side by side into the residual buffer
        const long rows = (long)batch * op.Lout;
        const long n4 = rows * ((op.cat_c0 + op.cat_c1) / 4);
        hipLaunchKernelGGL(dad::concat_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                           buf(op.res), buf(op.cat0), buf(op.cat1), rows, op.cat_c0, op.cat_c1);
        HIP_TRY(hipGetLastError());
    }
    ConvIO io;
    io.src0 = op.src0 == -2 ? xext : buf(op.src0);
    io.src1 = buf(op.src1);
    // shared timestep: row t of the table; per-row timesteps: row 0 + trow[b] * stride in the kernel
    if (op.temb_off >= 0)
        io.temb = temb_rows != nullptr ? temb_rows + op.temb_off
                                       : m->d_temb_table + (trow ? 0L : (long)t * m->plan.temb_width) + op.temb_off;
    // per-row timesteps only where there is a time embedding to index: the kernels' fallback operand
    // for an absent embedding is the bias row, which must not be offset by trow[b] * stride
    io.trow = io.temb != nullptr ? trow : nullptr;
    io.res = op.res == -2 ? xext : buf(op.res);       // identity residual of the trajectory itself (td == C)
    io.dst = buf(op.dst);
    io.rdst = buf(op.rdst);
    io.pre = buf(op.pre); io.stats = buf(op.stats);
    io.slab = ws + plan.floats_per_sample * (long)batch;     // scratch behind the activations
    return launch_conv(m, op, batch, io, st);
}

int launch_conv(dad_model* m, const ConvOp& op, int batch, const ConvIO& io, hipStream_t st) {
    LaunchGeom g;
    int rc = plan_launch(*m, op, batch, g);
    if (rc != DAD_OK) return rc;
    ConvParams p{};
    p.src0 = io.src0;
    p.src1 = io.src1;
    p.w = op.d_w; p.bias = op.d_bias; p.gamma = op.d_gamma; p.beta = op.d_beta;
    p.temb = io.temb;
    p.trow = io.trow; p.temb_stride = m->plan.temb_width;
    p.res = io.res;
    p.dst = io.dst;
    p.pre = io.pre; p.stats = io.stats;
    p.cin0 = op.cin0; p.cin1 = op.cin1; p.cin_pad = op.cin_pad;
    p.M = op.M; p.cpg = op.norm.empty() ? 0 : op.cout / 8; p.cpg_real = op.gn_real;
    p.lreal = op.lreal; p.src_len = op.src_len;
    p.B = batch; p.Lin = op.Lin; p.Lout = op.Lout; p.lshift = ilog2(op.Lout);
    p.lshift_in = ilog2(op.Lin);
    p.interleave = op.kind == CONV_UP;
    p.ntiles_n = g.ntiles_n;
    p.xcd_gn = g.xcd_gn; p.xcd_mts = g.xcd_mts; p.xcd_ntn = g.xcd_ntn;
    p.kslices = g.split.kslices;
    p.chunks_per_slice = g.split.chunks_per_slice;
    p.slab = io.slab;
    p.counters = m->d_counters;
    p.c1 = op.c1; p.c2 = op.c2;
    p.xswz = g.xswz;
    p.wtaps = op.wtaps();
    p.rbias = g.fused ? op.d_rbias : nullptr;
    p.rdst = g.fused ? io.rdst : nullptr;
    if (g.split.kslices > 1 && io.slab == nullptr)
        return fail(DAD_E_WORKSPACE, "%s: split-K launch without scratch", op.name.c_str());
    static const bool trace = getenv("DAD_TRACE_TILES") != nullptr;     // tuning aid
    if (trace)
        fprintf(stderr, "[dad] %-34s B=%d M=%d K=%dx%d L=%d tile=%d (%dx%d SK%d) kslices=%d%s\n", op.name.c_str(),
                batch, op.M, op.taps, op.cin0 + op.cin1, op.Lout, g.cfg, kTiles[g.cfg].BM, kTiles[g.cfg].BN,
                kTiles[g.cfg].SK, g.split.kslices, g.fused ? " +res1x1" : "");
#ifdef DAD_STAMPS
    p.stamps = (g_stamps && &op >= &m->plan.convs[0] && &op < &m->plan.convs[0] + m->plan.convs.size())
                   ? g_stamps + (size_t)(&op - &m->plan.convs[0]) * 4096 * 8 : nullptr;
#endif
    const auto& table = kernel_table();
    const auto it = table.find(KernKey(g.cfg, op.taps, op.stride, op.x3, op.bdir, g.ragged, g.fused, g.padded));
    if (it == table.end())
        return fail(DAD_E_INVALID, "no kernel for %s (tile %d taps=%d stride=%d x3=%d bdir=%d ragged=%d res=%d)",
                    op.name.c_str(), g.cfg, op.taps, op.stride, (int)op.x3, (int)op.bdir, (int)g.ragged, (int)g.fused);
    void* args[] = {&p};
    HIP_TRY(hipLaunchKernel((const void*)it->second, dim3(g.gx, g.gy, g.gz), dim3(g.threads), args,
                            g.lds_bytes, st));
    return DAD_OK;
}

int build_time_tables(dad_model* m, hipStream_t st);
// The per-timestep tables follow the time-MLP tensors lazily (dad_model_refresh_weights marks them stale): every
// entry point that reads them calls this first, on the stream it launches on.
int ensure_tables(dad_model* m, hipStream_t st) {
    if (!m->tables_stale) return DAD_OK;
    const int rc = build_time_tables(m, st);
    if (rc == DAD_OK) m->tables_stale = false;
    return rc;
}

// rows per sample of the external tensors (x, noise, guide, means): the horizon before zero-padding
inline int traj_horizon(const dad_model* m) { return m->real_horizon > 0 ? m->real_horizon : m->cfg.horizon; }

int check_ready(const dad_model* m, int batch, int t, size_t ws_bytes) {
    if (!m) return fail(DAD_E_INVALID, "null model");
    if (!m->finalized) return fail(DAD_E_STATE, "dad_model_finalize has not been called");
    if (batch <= 0) return fail(DAD_E_INVALID, "batch must be positive (got %d)", batch);
    if (t < 0 || t >= m->cfg.n_timesteps)
        return fail(DAD_E_RANGE, "index %d is out of bounds for the schedule of size %d", t,
                    m->cfg.n_timesteps);
    const size_t need = workspace_bytes(*m, batch);
    if (ws_bytes < need)
        return fail(DAD_E_WORKSPACE, "workspace has %zu bytes, batch %d needs %zu", ws_bytes, batch,
                    need);
    return DAD_OK;
}

// ------------------------------------------------------------------ small-batch (CC) launches
// The tensor `in` names, as the consumer must read it (conv_cc.hpp).
dad::CcSrc cc_source(dad_model* m, const CcPlan& cc, const CcInput& in, int channels, const float* xext,
                     float* ws, int batch, int t) {
    dad::CcSrc s{};
    auto buf = [&](int id) -> float* { return ws + m->plan.bufs[id].offset * (long)batch; };
    float* slabs = ws + m->plan.floats_per_sample * (long)batch;
    s.C = channels;
    if (in.kind == 1) { s.data = xext; s.rows = batch * m->cfg.horizon; return s; }
    if (in.kind == 2) { s.data = buf(in.buf); return s; }
    const ConvOp& q = m->plan.convs[in.producer];
    const CcOp& qo = cc.ops[in.producer];
    s.data = slabs + qo.oslab;
    s.nsl = qo.kslices;
    s.C = qo.out_cols;
    s.rows = qo.out_rows;
    s.bias = q.d_bias;
    if (!q.norm.empty()) { s.gamma = q.d_gamma; s.beta = q.d_beta; s.cpg = q.cout / 8; }
    if (q.temb_off >= 0) s.temb = m->d_temb_table + (long)t * m->plan.temb_width + q.temb_off;
    if (qo.res_kind == 1) s.res = xext;
    else if (qo.res_kind == 2) s.res = buf(qo.res_buf);
    else if (qo.res_kind == 3) {
        const CcOp& r = cc.ops[qo.res_ride];
        s.rslab = slabs + r.orslab; s.nrs = r.kslices; s.rbias = m->plan.convs[qo.res_ride].d_rbias;
    } else if (qo.res_kind == 4) {                        // the block's stand-alone 1x1 residual conv
        const CcOp& r = cc.ops[qo.res_ride];
        s.rslab = slabs + r.oslab; s.nrs = r.kslices; s.rbias = m->plan.convs[qo.res_ride].d_bias;
    }
    s.mat = buf(in.buf);
    return s;
}

int run_conv_cc(dad_model* m, const CcPlan& cc, int i, const float* xext, float* ws, int batch, int t,
                hipStream_t st) {
    const ConvOp& op = m->plan.convs[i];
    const CcOp& o = cc.ops[i];
    dad::CcParams p{};
    p.src0 = cc_source(m, cc, o.in0, op.cin0, xext, ws, batch, t);
    if (o.in1.kind != 0) p.src1 = cc_source(m, cc, o.in1, op.cin1, xext, ws, batch, t);
    p.w = op.d_w; p.wtaps = op.wtaps();
    p.cin0 = op.cin0; p.cin1 = op.cin1; p.M = op.M;
    p.B = batch; p.Lin = op.Lin; p.Lout = op.Lout;
    p.lshift = ilog2(op.Lout); p.lshift_in = ilog2(op.Lin);
    p.interleave = op.kind == CONV_UP;
    p.slice_ch = o.slice_ch;
    float* slabs = ws + m->plan.floats_per_sample * (long)batch;
    p.oslab = slabs + o.oslab;
    p.orslab = o.orslab >= 0 ? slabs + o.orslab : nullptr;
    p.out_rows = o.out_rows;
    const bool shape_ok = (op.taps == 5 && op.stride == 1) || (op.taps == 3 && op.stride == 2) || (op.taps == 2 && op.stride == 1);
    const bool big = p.src0.nsl > dad::CC_MAX_SLABS || p.src0.nrs > dad::CC_MAX_SLABS ||
                     p.src1.nsl > dad::CC_MAX_SLABS || p.src1.nrs > dad::CC_MAX_SLABS;
    const void* kern = shape_ok ? cc_kernel(op.taps, op.ride, big, o.tile_rows, op.Lout > 32) : nullptr;
    if (o.wide) {
        if (big) return fail(DAD_E_INVALID, "wide small-batch conv %s: more than %d partial slabs", op.name.c_str(), dad::CC_MAX_SLABS);
        kern = ccw_kernel(op.taps, op.ride, p.src0.rslab != nullptr || p.src1.rslab != nullptr, o.tile_rows);
    }
    if (!kern) return fail(DAD_E_INVALID, "no small-batch kernel for %s (taps=%d stride=%d)", op.name.c_str(), op.taps, op.stride);
    static const bool trace = getenv("DAD_TRACE_TILES") != nullptr;
    if (trace)
        fprintf(stderr, "[dad] %-34s B=%d M=%d K=%dx%d L=%d cc slice=%d kslices=%d ntiles=%d%s\n", op.name.c_str(), batch,
                op.M, op.taps, op.cin0 + op.cin1, op.Lout, o.slice_ch, o.kslices, o.ntiles, op.ride ? " +res1x1" : "");
    if (trace && o.wide) fprintf(stderr, "[dad]   (wide: %d-row tiles, %zu B LDS)\n", o.tile_rows, o.lds_bytes);
#ifdef DAD_STAMPS
    p.stamps = g_stamps ? g_stamps + (size_t)i * 16 : nullptr;
#endif
    void* args[] = {&p};
    HIP_TRY(hipLaunchKernel(kern, dim3(o.kslices, op.M / 32, o.ntiles), dim3(dad::CC_THREADS), args, o.lds_bytes, st));
    return DAD_OK;
}

int run_unet(dad_model* m, const float* x, int t, int batch, float* ws, hipStream_t st,
             const int32_t* trow = nullptr, const CcPlan* cc = nullptr, bool train = false,
             const float* temb_rows = nullptr) {
    // Profiling brackets the whole run of conv-GEMM launches of one denoiser evaluation with
    // ONE pair of HIP events on the launch stream (events between individual launches would
    // break the back-to-back dispatch they are meant to time).
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (m->profile) {
        if (m->ev_used == m->ev_pool.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            m->ev_pool.push_back({a, b});
        }
        e0 = m->ev_pool[m->ev_used].first;
        e1 = m->ev_pool[m->ev_used].second;
        ++m->ev_used;
        HIP_TRY(hipEventRecord(e0, st));
    }
    const Plan& plan = train ? m->tplan : m->plan;
    const std::vector<ConvOp>& convs = plan.convs;
    if (cc != nullptr) {                                  // small batch: consumer-combine kernels
        for (size_t i = 0; i < convs.size(); ++i) {
            if (m->profile) m->prof_flops += convs[i].flops_per_sample * batch;
            if (!cc->ops[i].launched) continue;
            const int rc = run_conv_cc(m, *cc, (int)i, x, ws, batch, t, st);
            if (rc != DAD_OK) return rc;
            if (m->profile) ++m->prof_launches;
        }
        if (m->profile) HIP_TRY(hipEventRecord(e1, st));
        return DAD_OK;
    }
    // level-0 chain: the first six plan entries as ONE launch (inference plan, shared timestep, large batch)
    int first = 0;
    if (!train && trow == nullptr && (int)m->chain.size() == dad::CH_N && m->chain_enabled && batch >= m->chain_min_batch) {
        dad::ChainParams cp{};
        cp.x = x;
        cp.out = ws + plan.bufs[convs[m->chain_last].dst].offset * (long)batch;
        cp.temb_row = m->d_temb_table + (long)t * m->plan.temb_width;
        cp.B = batch; cp.td = m->cfg.transition_dim;
        for (int i = 0; i < dad::CH_N; ++i) cp.c[i] = m->chain[i];
        switch (m->chain_C) {
            case 32: hipLaunchKernelGGL(dad::chain_l0_kernel<32>, dim3(batch), dim3(dad::ChainShape<32>::NT),
                                        dad::ChainShape<32>::LDS_FLOATS * sizeof(float), st, cp); break;
            case 64: hipLaunchKernelGGL(dad::chain_l0_kernel<64>, dim3(batch), dim3(dad::ChainShape<64>::NT),
                                        dad::ChainShape<64>::LDS_FLOATS * sizeof(float), st, cp); break;
            default: hipLaunchKernelGGL(dad::chain_l0_kernel<128>, dim3(batch), dim3(dad::ChainShape<128>::NT),
                                        dad::ChainShape<128>::LDS_FLOATS * sizeof(float), st, cp); break;
        }
        HIP_TRY(hipGetLastError());
        first = m->chain_last + 1;
        if (m->profile) {
            for (int i = 0; i < first; ++i) m->prof_flops += convs[i].flops_per_sample * batch;
            ++m->prof_launches;
        }
    }
    for (size_t oi = (size_t)first; oi < convs.size(); ++oi) {
        const ConvOp& op = convs[oi];
        // a residual 1x1 conv whose block's first conv carries it at this batch is not launched
        const bool rides = op.rider_of >= 0 && fused_at(*m, convs[op.rider_of], batch);
        if (m->profile) m->prof_flops += op.flops_per_sample * batch;
        if (rides) continue;
        const int rc = run_conv(m, plan, op, x, ws, batch, t, st, trow, temb_rows);
        if (rc != DAD_OK) return rc;
        if (m->profile) ++m->prof_launches;
    }
    if (m->profile) HIP_TRY(hipEventRecord(e1, st));
    return DAD_OK;
}

int run_final(dad_model* m, float* x, const float* x_ro, int t, int batch, const dad_step_args* a,
              int x_out_disabled, float* eps_only, float* ws, hipStream_t st,
              bool seed_from_device = false, const CcPlan* cc = nullptr, bool train = false) {
    const dad_cfg& c = m->cfg;
    dad::FinalParams p{};
    const Plan& plan = train ? m->tplan : m->plan;
    p.act = ws + plan.bufs[plan.final_act].offset * (long)batch;
    p.w = m->d_final_w; p.bias = m->d_final_b;
    p.dim = c.dim; p.td = c.transition_dim; p.B = batch; p.H = traj_horizon(m); p.Hact = c.horizon;
    p.predict_epsilon = c.predict_epsilon; p.clip_denoised = c.clip_denoised;
    if (eps_only) {
        p.x = const_cast<float*>(x_ro);
        p.eps_out = eps_only;
        p.x_out_disabled = 1;
    } else {
        p.x = x;
        p.noise = a->noise; p.cond0 = a->cond0; p.cond_per_row = a->cond_per_row;
        p.guide = (a->guide_grad && a->guide_weight > 0.0f) ? a->guide_grad : nullptr;
        p.mean_out = a->mean_out; p.eps_out = a->eps_out;
        p.x_out_disabled = x_out_disabled;
        const float lv = m->sched[4][t];
        p.c_recip = m->sched[0][t]; p.c_recipm1 = m->sched[1][t];
        p.coef1 = m->sched[2][t]; p.coef2 = m->sched[3][t];
        p.sigma = t == 0 ? 0.0f : expf(0.5f * lv);
        p.guide_scale = a->guide_weight * expf(lv);
        p.seed = a->seed;
        p.elem_offset = a->row_offset * (uint64_t)traj_horizon(m) * (uint64_t)c.transition_dim;
        p.draw = a->draw;
        p.seed_dev = seed_from_device ? (const unsigned long long*)m->d_rng : nullptr;
    }
    if (cc != nullptr) {
        dad::FinalCcParams fp{};
        CcInput in; in.kind = 3; in.producer = cc->final_producer; in.buf = m->plan.final_act;
        fp.src = cc_source(m, *cc, in, c.dim, x_ro ? x_ro : x, ws, batch, t);
        fp.src.mat = nullptr;
        fp.f = p;
        const size_t lds_cc = dad::final_cc_lds_floats(c.transition_dim, c.dim, c.horizon) * sizeof(float);
        if (lds_cc > dad::kLdsBytes)
            return fail(DAD_E_INVALID, "final 1x1 conv does not fit LDS (td=%d, dim=%d)", c.transition_dim, c.dim);
        // one block per (sample, group of output columns): enough columns per block to occupy its
        // 512 threads once, as long as the grid stays within one wave of blocks
        const int want = (c.horizon * c.transition_dim + dad::CC_THREADS - 1) / dad::CC_THREADS;
        const int gy = std::max(1, std::min({want, c.transition_dim, 256 / std::max(batch, 1)}));
        hipLaunchKernelGGL(dad::final_cc_kernel, dim3(batch, gy), dim3(dad::CC_THREADS), lds_cc, st, fp);
        HIP_TRY(hipGetLastError());
        return DAD_OK;
    }
    const long N = (long)batch * traj_horizon(m);
    // columns of the transition are spread over gridDim.y when the row tiles alone leave CUs idle
    // (a block stages only the weight rows of its own columns), and further until a block fits LDS
    const long row_tiles = (N + dad::FINAL_COLS - 1) / dad::FINAL_COLS;
    const int jg = 256 / dad::FINAL_COLS;
    const long col_groups = (c.transition_dim + jg - 1) / jg;
    long gy = std::max(1L, std::min(col_groups, 512 / row_tiles));
    while (gy < col_groups && dad::final_lds_floats(c.transition_dim, c.dim, (int)gy) * sizeof(float) > dad::kLdsBytes) ++gy;
    const size_t lds = dad::final_lds_floats(c.transition_dim, c.dim, (int)gy) * sizeof(float);
    if (lds > dad::kLdsBytes)
        return fail(DAD_E_INVALID, "final 1x1 conv does not fit LDS (td=%d, dim=%d)", c.transition_dim, c.dim);
    hipLaunchKernelGGL(dad::final_posterior_kernel, dim3((unsigned)row_tiles, (unsigned)gy), dim3(256), lds, st, p);
    HIP_TRY(hipGetLastError());
    return DAD_OK;
}

int run_project(const dad_project_args* pa, float alpha, float* x, int batch, int horizon,
                hipStream_t st, float* violation = nullptr) {
    if (!pa || !pa->P) return fail(DAD_E_INVALID, "projection arguments missing");
    if (alpha <= 0.0f && !violation) return DAD_OK;       // policies.py:428-429
    {
        const int rc0 = configure_kernels();
        if (rc0 != DAD_OK) return rc0;
    }
    dad::ProjParams p{};
    p.P = pa->P; p.obs_mean = pa->obs_mean; p.obs_std = pa->obs_std;
    p.act_mean = pa->act_mean; p.act_std = pa->act_std;
    p.x = x; p.B = batch; p.H = horizon; p.n = pa->state_dim; p.od = pa->observation_dim;
    p.m = pa->action_dim;
    p.D = (horizon + 1) * p.n + horizon * p.m;
    p.alpha = alpha;
    p.one_minus_alpha = (float)(1.0 - (double)alpha);
    p.violation = violation;
    // batches of 32+ trajectories with scratch for the projected copy: v @ P as an MFMA GEMM (P read once
    // per 32 trajectories, not once per trajectory)
    const size_t x_bytes = (size_t)batch * horizon * (pa->observation_dim + pa->action_dim) * sizeof(float);
    // (at PointMaze size, D = 196, the per-trajectory kernel is the faster one: 10 us against 16.6 us for
    // 256 plans — 56 GEMM blocks leave most of the chip idle; the GEMM takes over from D = 512)
    const size_t row_lds_probe = (size_t)(1 + 16) * p.D * sizeof(float);
    const bool gemm_pays = p.D >= 512 || row_lds_probe > dad::kLdsBytes;
    if (!violation && gemm_pays && batch >= 32 && pa->scratch != nullptr && pa->scratch_bytes >= x_bytes) {
        p.xout = pa->scratch;
        const dim3 grid((unsigned)((batch + 31) / 32), (unsigned)((p.D + 31) / 32));
        hipLaunchKernelGGL(dad::project_gemm_kernel, grid, dim3(dad::PG_THREADS),
                           dad::project_gemm_lds_floats() * sizeof(float), st, p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(x, pa->scratch, x_bytes, hipMemcpyDeviceToDevice, st));
        return DAD_OK;
    }
    p.xout = x;
    // rows per block: one while the batch fits one wave of blocks (every CU streams P once),
    // four beyond that (P is then re-used by four rows per pass) if four rows fit LDS
    const size_t row_lds = (size_t)(1 + 16) * p.D * sizeof(float);    // a row + its 16 partial sets
    const int rb = (batch > 512 && 4 * row_lds <= dad::kLdsBytes) ? 4 : 1;
    const size_t lds = rb * row_lds;
    if (lds > dad::kLdsBytes)
        return fail(DAD_E_INVALID, "projection dimension D=%d: one trajectory's partial sums exceed LDS; pass "
                    "dad_project_args.scratch and a batch of at least 32 for the GEMM form", p.D);
    if (rb == 1)
        hipLaunchKernelGGL((dad::project_kernel<1, 16>), dim3(batch), dim3(1024), lds, st, p);
    else
        hipLaunchKernelGGL((dad::project_kernel<4, 16>), dim3((batch + 3) / 4), dim3(1024), lds, st, p);
    HIP_TRY(hipGetLastError());
    return DAD_OK;
}

// The level-0 chain's weight images from the standard packed images of its convs (device side: also after
// dad_model_refresh_weights).
int repack_chain(dad_model* m, hipStream_t st) {
    for (size_t k = 0; k < m->chain.size(); ++k) {
        const ConvOp& op = m->plan.convs[m->chain_src[k]];
        const int nch = (op.cin0 + 15) / 16;
        const long n4 = (long)nch * 2 * op.wtaps() * (op.M / 32) * 64;
        hipLaunchKernelGGL(dad::chain_repack_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                           const_cast<float*>(m->chain[k].w), op.d_w, nch, op.wtaps(), op.M);
        HIP_TRY(hipGetLastError());
    }
    return DAD_OK;
}

// Per-timestep tables from the device copies of the time-MLP tensors: time_mlp (Linear -> Mish -> Linear
// on the sinusoid rows) and every block's Mish -> Linear (temporal_unet.py:97-100,155-160).
int build_time_tables(dad_model* m, hipStream_t st) {
    const dad_cfg& c = m->cfg;
    const int T = c.n_timesteps, dim = c.dim, tdm = c.time_dim;
    auto linear = [&](const float* in, const std::string& key, float* out, int K, int M, int stride, int mish_in) -> int {
        auto w = m->d_time.find(key + ".weight"), b = m->d_time.find(key + ".bias");
        if (w == m->d_time.end() || b == m->d_time.end()) return fail(DAD_E_KEY, "missing key '%s'", key.c_str());
        const long total = (long)T * M;
        hipLaunchKernelGGL(dad::table_linear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256),
                           0, st, in, w->second, b->second, out, T, K, M, stride, mish_in);
        HIP_TRY(hipGetLastError());
        return DAD_OK;
    };
    int rc;
    if ((rc = linear(m->d_emb, "time_mlp.1", m->d_h1, dim, 4 * tdm, 4 * tdm, 0)) != DAD_OK) return rc;
    if ((rc = linear(m->d_h1, "time_mlp.3", m->d_temb, 4 * tdm, tdm, tdm, 1)) != DAD_OK) return rc;
    for (const ConvOp& op : m->plan.convs) {
        if (op.temb_off < 0) continue;
        std::string base = op.name.substr(0, op.name.size() - std::strlen(".blocks.0.block.0"));
        if ((rc = linear(m->d_temb, base + ".time_mlp.1", m->d_temb_table + op.temb_off, tdm, op.cout,
                         m->plan.temb_width, 1)) != DAD_OK) return rc;
    }
    return DAD_OK;
}

}  // namespace

// ===================================================================================== ABI
extern "C" {

const char* dad_last_error(void) { return g_err; }
const char* dad_version(void) { return "dad-hip 0.1 (gfx950, fp32 MFMA)"; }

int dad_model_create(const dad_cfg* cfg, dad_model** out) {
    if (!cfg || !out) return fail(DAD_E_INVALID, "null argument");
    int rc = check_cfg(cfg);
    if (rc != DAD_OK) return rc;
    std::unique_ptr<dad_model> m(new dad_model());
    m->cfg = *cfg;
    // A/B switches for tuning runs (read once, per model)
    m->xswz_enabled = getenv("DAD_NO_XSWZ") == nullptr;
    m->xcd_order = getenv("DAD_NO_XCD_ORDER") == nullptr;
    m->fuse_residual = getenv("DAD_NO_FUSE_RES") == nullptr;
    if (const char* v = getenv("DAD_SPLIT_TARGET")) m->split_target = std::max(1, atoi(v));
    m->cc_enabled = getenv("DAD_NO_CC") == nullptr;
    m->chain_enabled = getenv("DAD_CHAIN") != nullptr;
    if (const char* v = getenv("DAD_CC_MAX_ROWS")) m->cc_max_rows = std::max(0, atoi(v));
    if (const char* v = getenv("DAD_WGRAD_BLOCKS")) m->wgrad_blocks = std::max(1, atoi(v));
    if ((rc = build_plan(m.get())) != DAD_OK) return rc;
    *out = m.release();
    return DAD_OK;
}

void dad_model_destroy(dad_model* m) {
    if (!m) return;
    free_device(m);
    if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
    for (auto& e : m->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    delete m;
}

int dad_model_load_weight(dad_model* m, const char* key, const float* data, const int64_t* shape,
                          int32_t ndim) {
    if (!m || !key || !data || !shape) return fail(DAD_E_INVALID, "null argument");
    auto it = m->expected.find(key);
    if (it == m->expected.end()) return fail(DAD_E_KEY, "unexpected key '%s'", key);
    if ((int)it->second.size() != ndim) return fail(DAD_E_KEY, "'%s': rank %d, expected %zu", key, ndim, it->second.size());
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] != it->second[i])
            return fail(DAD_E_KEY, "'%s': size mismatch at dim %d (%lld vs %lld)", key, i,
                        (long long)shape[i], (long long)it->second[i]);
        n *= (size_t)shape[i];
    }
    HostTensor& t = m->raw[key];
    t.shape.assign(shape, shape + ndim);
    t.data.assign(data, data + n);
    m->finalized = false;
    return DAD_OK;
}

int dad_model_load_schedule(dad_model* m, const float* a, const float* b, const float* c1,
                            const float* c2, const float* lv) {
    if (!m || !a || !b || !c1 || !c2 || !lv) return fail(DAD_E_INVALID, "null argument");
    const float* src[5] = {a, b, c1, c2, lv};
    for (int i = 0; i < 5; ++i) m->sched[i].assign(src[i], src[i] + m->cfg.n_timesteps);
    m->have_sched = true;
    return DAD_OK;
}

int dad_model_load_time_embedding(dad_model* m, const float* emb, int32_t n_timesteps, int32_t dim) {
    if (!m || !emb) return fail(DAD_E_INVALID, "null argument");
    if (n_timesteps != m->cfg.n_timesteps || dim != m->cfg.dim)
        return fail(DAD_E_INVALID, "time embedding must be (%d, %d), got (%d, %d)", m->cfg.n_timesteps,
                    m->cfg.dim, n_timesteps, dim);
    m->emb_override.assign(emb, emb + (size_t)n_timesteps * dim);
    m->finalized = false;
    return DAD_OK;
}

int dad_model_set_precision(dad_model* m, int32_t precision) {
    if (!m) return fail(DAD_E_INVALID, "null model");
    if (precision != DAD_PREC_FP32 && precision != DAD_PREC_F16X3)
        return fail(DAD_E_INVALID, "unknown precision %d (DAD_PREC_FP32 = 0, DAD_PREC_F16X3 = 1)", precision);
    if (precision != m->precision) m->finalized = false;       // weights must be re-packed
    m->precision = precision;
    decide_kernel_families(m);
    return DAD_OK;
}

int dad_model_set_horizon(dad_model* m, int32_t real_horizon) {
    if (!m) return fail(DAD_E_INVALID, "null model");
    const int padded = m->cfg.horizon, down = 1 << (m->cfg.n_levels - 1);
    if (real_horizon < down || real_horizon > padded || real_horizon % down != 0)
        return fail(DAD_E_INVALID, "horizon %d: need a multiple of 2^(levels-1) = %d (every level halves the length) of at most the "
                    "padded horizon %d", real_horizon, down, padded);
    m->real_horizon = real_horizon == padded ? 0 : real_horizon;
    const int rc = build_plan(m);                 // the same plan; every conv now knows how many of its rows exist
    if (rc != DAD_OK) return rc;
    m->finalized = false;
    if (m->training) if (const char* why = training_refusal(*m)) return fail(DAD_E_INVALID, "training: %s", why);
    return DAD_OK;
}

int dad_model_set_group_channels(dad_model* m, const int32_t* real_channels, int32_t n_levels) {
    if (!m || !real_channels) return fail(DAD_E_INVALID, "null argument");
    if (n_levels != m->cfg.n_levels) return fail(DAD_E_INVALID, "%d levels given, the model has %d", n_levels, m->cfg.n_levels);
    for (int i = 0; i < n_levels; ++i) {
        const int real = real_channels[i], padded = m->cfg.channels[i];
        if (real < 8 || real % 8 != 0 || real > padded)
            return fail(DAD_E_INVALID, "level %d: %d real channels in %d (need a multiple of 8, at most the padded width)", i, real, padded);
    }
    for (int i = 0; i < DAD_MAX_LEVELS; ++i) m->real_channels[i] = i < n_levels ? real_channels[i] : 0;
    const int rc = build_plan(m);                 // the same plan; every GroupNorm'd conv now knows its real group width
    if (rc != DAD_OK) return rc;
    m->finalized = false;
    if (m->training) if (const char* why = training_refusal(*m)) return fail(DAD_E_INVALID, "training: %s", why);
    return DAD_OK;
}

int dad_model_finalize(dad_model* m, dad_stream_t stream) {
    if (!m) return fail(DAD_E_INVALID, "null model");
    if (!m->have_sched) return fail(DAD_E_STATE, "schedule not loaded");
    for (auto& kv : m->expected)
        if (!m->raw.count(kv.first)) return fail(DAD_E_KEY, "missing key '%s'", kv.first.c_str());
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipStreamSynchronize(st));
    {
        const int rc0 = configure_kernels();
        if (rc0 != DAD_OK) return rc0;
    }
    free_device(m);
    const dad_cfg& c = m->cfg;
    {
        void* a = nullptr;
        m->arena_cap = arena_bytes_needed(*m);
        HIP_TRY(hipMalloc(&a, m->arena_cap));
        m->owned.push_back(a);
        m->arena = (char*)a;
        m->arena_used = 0;
    }

    HIP_TRY(hipGetDevice(&m->device));
    for (ConvOp& op : m->plan.convs) {
        PackedOp packed;
        int rc = pack_op(m, op, packed);
        if (rc != DAD_OK) return rc;
        if ((rc = upload(m, packed.w, &op.d_w)) != DAD_OK) return rc;
        if ((rc = upload(m, packed.bias, &op.d_bias)) != DAD_OK) return rc;
        if (op.ride && (rc = upload(m, packed.rbias, &op.d_rbias)) != DAD_OK) return rc;
        if (!op.norm.empty()) {
            if ((rc = upload(m, m->raw[op.norm + ".weight"].data, &op.d_gamma)) != DAD_OK) return rc;
            if ((rc = upload(m, m->raw[op.norm + ".bias"].data, &op.d_beta)) != DAD_OK) return rc;
        }
    }
    int rc;
    for (size_t i = 0; i < m->plan.convs.size(); ++i) {        // the training plan launches the same images
        const ConvOp& a = m->plan.convs[i];
        ConvOp& b = m->tplan.convs[i];
        b.d_w = a.d_w; b.d_bias = a.d_bias; b.d_gamma = a.d_gamma; b.d_beta = a.d_beta; b.d_rbias = a.d_rbias;
        b.c1 = a.c1; b.c2 = a.c2;
    }
    {   // level-0 chain table (conv_chain.hpp)
        const ChainPlan cp = chain_plan(*m);
        if (cp.ok) {
            const std::vector<ConvOp>& v = m->plan.convs;
            std::vector<dad::ChainConv> tab;
            auto entry = [&](const ConvOp& op, int src, int dst, int add_t0, int ride) {
                dad::ChainConv e{};
                void* img = nullptr;                        // the chain's own image of this conv (filled below)
                const int nch = (op.cin0 + 15) / 16;
                if (arena_alloc(m, (size_t)nch * op.wtaps() * op.M * 16 * sizeof(float), &img) != DAD_OK) return;
                m->chain_src.push_back((int)(&op - &v[0]));
                e.w = (const float*)img; e.bias = op.d_bias; e.gamma = op.d_gamma; e.beta = op.d_beta; e.rbias = ride ? op.d_rbias : nullptr;
                e.temb_off = op.temb_off; e.cin = op.cin0; e.taps = op.taps; e.stride = op.stride; e.wtaps = op.wtaps();
                e.src = src; e.dst = dst; e.add_t0 = add_t0; e.ride = ride;
                tab.push_back(e);
            };
            entry(v[0], 0, 2, 0, 1);      // x -> T1, ride -> T0
            entry(v[2], 2, 1, 1, 0);      // T1 -> T0 (+ T0)
            entry(v[3], 1, 2, 0, 0);      // T0 -> T1
            entry(v[4], 2, 1, 1, 0);      // T1 -> T0 (+ T0: identity residual)
            entry(v[5], 1, 3, 0, 0);      // T0 -> memory
            if ((int)tab.size() != dad::CH_N) return fail(DAD_E_STATE, "parameter arena exhausted (level-0 chain images)");
            m->chain = tab;
            m->chain_C = cp.C; m->chain_last = cp.last;
            if ((rc = repack_chain(m, st)) != DAD_OK) return rc;
        }
    }
    if (m->training) {
        if (const char* why = training_refusal(*m)) return fail(DAD_E_INVALID, "training: %s", why);
        std::vector<float> zeros((size_t)std::max(m->max_bwd_m, 2 * m->max_cout) + 64, 0.0f);
        if ((rc = upload(m, zeros, &m->d_zero)) != DAD_OK) return rc;
        std::vector<float> img;
        for (size_t i = 0; i < m->bconvs.size(); ++i)
            for (int k = 0; k < m->bconvs[i].n; ++k) {
                if ((rc = pack_bwd_op(m, m->tplan.convs[i], m->bconvs[i], k, img)) != DAD_OK) return rc;
                if ((rc = upload(m, img, &m->bconvs[i].op[k].d_w)) != DAD_OK) return rc;
                m->bconvs[i].op[k].d_bias = m->d_zero;
            }
        if ((rc = pack_bwd_final(m, img)) != DAD_OK) return rc;
        if ((rc = upload(m, img, &m->bfinal.d_w)) != DAD_OK) return rc;
        m->bfinal.d_bias = m->d_zero;
    }
    if ((rc = upload(m, m->raw["final_conv.1.weight"].data, &m->d_final_w)) != DAD_OK) return rc;
    if ((rc = upload(m, m->raw["final_conv.1.bias"].data, &m->d_final_b)) != DAD_OK) return rc;

    // ---- time-embedding tables: every t in [0, T) at once --------------------------------
    const int T = c.n_timesteps, dim = c.dim, tdm = c.time_dim;
    // SinusoidalPosEmb (temporal_unet.py:27-31): the caller's table when one was handed over
    // (dad_model_load_time_embedding: the Python mirror evaluates the reference's own torch
    // expression), else the same formula with the C library's expf/sinf/cosf
    const std::vector<float> emb = m->emb_override.size() == (size_t)T * dim ? m->emb_override
                                                                            : sinusoid_table(T, dim);
    float *d_emb, *d_h1, *d_temb;
    (void)c; (void)dim;
    if ((rc = upload(m, emb, &d_emb)) != DAD_OK) return rc;
    std::vector<float> zeros((size_t)T * 4 * tdm, 0.0f);
    if ((rc = upload(m, zeros, &d_h1)) != DAD_OK) return rc;
    zeros.resize((size_t)T * tdm);
    if ((rc = upload(m, zeros, &d_temb)) != DAD_OK) return rc;
    zeros.assign((size_t)T * std::max(1, m->plan.temb_width), 0.0f);
    if ((rc = upload(m, zeros, &m->d_temb_table)) != DAD_OK) return rc;
    m->d_emb = d_emb; m->d_temb = d_temb; m->d_h1 = d_h1;
    m->d_time.clear();
    for (const auto& kv : m->raw)                        // every time-MLP tensor keeps a device copy
        if (kv.first.find("time_mlp.") != std::string::npos) {
            float* dptr = nullptr;
            if ((rc = upload(m, kv.second.data, &dptr)) != DAD_OK) return rc;
            m->d_time[kv.first] = dptr;
        }
    if ((rc = build_time_tables(m, st)) != DAD_OK) return rc;
    void* rng = nullptr;
    if ((rc = arena_alloc(m, 64, &rng)) != DAD_OK) return rc;
    m->d_rng = (uint64_t*)rng;
    void* cnt = nullptr;
    if ((rc = arena_alloc(m, kMaxSplitTiles * sizeof(unsigned), &cnt)) != DAD_OK) return rc;
    HIP_TRY(hipMemsetAsync(cnt, 0, kMaxSplitTiles * sizeof(unsigned), st));
    m->d_counters = (unsigned*)cnt;
    HIP_TRY(hipStreamSynchronize(st));
    m->raw.clear();
    m->finalized = true;
    return DAD_OK;
}

int dad_workspace_bytes(const dad_model* m, int32_t batch, size_t* bytes) {
    if (!m || !bytes || batch <= 0) return fail(DAD_E_INVALID, "bad argument");
    *bytes = workspace_bytes(*m, batch);
    return DAD_OK;
}

int dad_unet_forward(dad_model* m, const float* x, int32_t t, float* out, int32_t batch,
                     void* workspace, size_t workspace_bytes, dad_stream_t stream) {
    int rc = check_ready(m, batch, t, workspace_bytes);
    if (rc != DAD_OK) return rc;
    if (!x || !out || !workspace) return fail(DAD_E_INVALID, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_tables(m, st)) != DAD_OK) return rc;
    const CcPlan cc = cc_plan(*m, batch);
    const CcPlan* ccp = cc.ok ? &cc : nullptr;
    if ((rc = run_unet(m, x, t, batch, (float*)workspace, st, nullptr, ccp)) != DAD_OK) return rc;
    return run_final(m, nullptr, x, t, batch, nullptr, 1, out, (float*)workspace, st, false, ccp);
}

int dad_unet_forward_rows(dad_model* m, const float* x, const int32_t* t_rows, float* out, int32_t batch,
                          void* workspace, size_t workspace_bytes, dad_stream_t stream) {
    int rc = check_ready(m, batch, 0, workspace_bytes);
    if (rc != DAD_OK) return rc;
    if (!x || !out || !workspace || !t_rows) return fail(DAD_E_INVALID, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_tables(m, st)) != DAD_OK) return rc;
    if ((rc = run_unet(m, x, 0, batch, (float*)workspace, st, t_rows)) != DAD_OK) return rc;
    return run_final(m, nullptr, x, 0, batch, nullptr, 1, out, (float*)workspace, st);
}

int dad_denoise_step(dad_model* m, float* x, int32_t t, int32_t batch, const dad_step_args* args,
                     int32_t x_out_disabled, void* workspace, size_t workspace_bytes,
                     dad_stream_t stream) {
    int rc = check_ready(m, batch, t, workspace_bytes);
    if (rc != DAD_OK) return rc;
    if (!x || !args || !workspace) return fail(DAD_E_INVALID, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_tables(m, st)) != DAD_OK) return rc;
    const CcPlan cc = cc_plan(*m, batch);
    const CcPlan* ccp = cc.ok ? &cc : nullptr;
    if ((rc = run_unet(m, x, t, batch, (float*)workspace, st, nullptr, ccp)) != DAD_OK) return rc;
    return run_final(m, x, nullptr, t, batch, args, x_out_disabled, nullptr, (float*)workspace, st, false, ccp);
}

int dad_project(const dad_project_args* p, float alpha, float* x, int32_t batch, int32_t horizon,
                dad_stream_t stream) {
    if (!x || batch <= 0 || horizon <= 0) return fail(DAD_E_INVALID, "bad argument");
    return run_project(p, alpha, x, batch, horizon, (hipStream_t)stream);
}

int dad_projection_violation(const dad_project_args* p, const float* x, float* violation, int32_t batch,
                             int32_t horizon, dad_stream_t stream) {
    if (!x || !violation || batch <= 0 || horizon <= 0) return fail(DAD_E_INVALID, "bad argument");
    return run_project(p, 1.0f, const_cast<float*>(x), batch, horizon, (hipStream_t)stream, violation);
}

int dad_sample_loop(dad_model* m, float* x, int32_t n_steps, int32_t batch,
                    const float* noise_stack, uint64_t seed, uint64_t row_offset,
                    const float* cond0, int32_t cond_per_row, const dad_project_args* proj,
                    const float* proj_alphas_host, int32_t use_graph, void* workspace,
                    size_t workspace_bytes, dad_stream_t stream) {
    if (n_steps < 1) return fail(DAD_E_INVALID, "n_steps must be positive");
    int rc = check_ready(m, batch, n_steps - 1, workspace_bytes);
    if (rc != DAD_OK) return rc;
    if (!x || !workspace) return fail(DAD_E_INVALID, "null pointer");
    if (proj && !proj_alphas_host) return fail(DAD_E_INVALID, "projection needs per-step alphas");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_tables(m, st)) != DAD_OK) return rc;      // (before any capture: the replayed loop reads the tables)
    const long step_elems = (long)batch * traj_horizon(m) * m->cfg.transition_dim;

    const bool seed_dev = use_graph && !m->profile && noise_stack == nullptr;
    const CcPlan cc = cc_plan(*m, batch);
    const CcPlan* ccp = cc.ok ? &cc : nullptr;
    auto enqueue_all = [&](hipStream_t st) -> int {
        for (int j = 0; j < n_steps; ++j) {
            const int t = n_steps - 1 - j;
            dad_step_args a{};
            a.noise = noise_stack ? noise_stack + (long)j * step_elems : nullptr;
            a.seed = seed; a.row_offset = row_offset; a.draw = (uint64_t)(j + 1);
            a.cond0 = cond0; a.cond_per_row = cond_per_row;
            int r = run_unet(m, x, t, batch, (float*)workspace, st, nullptr, ccp);
            if (r != DAD_OK) return r;
            if ((r = run_final(m, x, nullptr, t, batch, &a, 0, nullptr, (float*)workspace, st,
                               seed_dev, ccp)) != DAD_OK)
                return r;
            if (proj && (r = run_project(proj, proj_alphas_host[t], x, batch, traj_horizon(m), st)) != DAD_OK)
                return r;
        }
        return DAD_OK;
    };

    if (!use_graph || m->profile) return enqueue_all(st);

    // Graph replay: the whole T-step loop is one hipGraph keyed by every frozen pointer and
    // scalar.  With in-kernel noise the Philox key is read from device memory, written by a
    // tiny kernel ahead of the replay, so a new seed does not need a new capture.
    if (seed_dev) {
        hipLaunchKernelGGL(dad::set_u64_kernel, dim3(1), dim3(1), 0, st,
                           (unsigned long long*)m->d_rng, (unsigned long long)seed);
        HIP_TRY(hipGetLastError());
    }
    GraphKey key{};
    key.x = x; key.noise = noise_stack; key.cond = cond0; key.ws = workspace;
    if (proj) {
        key.P = proj->P; key.obs_mean = proj->obs_mean; key.obs_std = proj->obs_std;
        key.act_mean = proj->act_mean; key.act_std = proj->act_std;
        key.proj_scratch = proj->scratch;
        key.state_dim = proj->state_dim; key.observation_dim = proj->observation_dim;
        key.action_dim = proj->action_dim;
    }
    key.n_steps = n_steps; key.batch = batch; key.cond_per_row = cond_per_row;
    key.force_tile = m->force_tile;
    key.flags = (m->split_enabled ? 1 : 0) | (m->fuse_residual ? 2 : 0) | (m->xswz_enabled ? 4 : 0) |
                (m->xcd_order ? 8 : 0) | (ccp ? 16 : 0) | (m->chain_enabled ? 32 : 0) | (m->split_target << 8);
    key.row_offset = row_offset;
    if (proj) {
        uint64_t hsh = 1469598103934665603ull;            // FNV-1a over the per-step alphas
        for (int i = 0; i < n_steps; ++i) {
            uint32_t bits;
            std::memcpy(&bits, &proj_alphas_host[i], 4);
            hsh = (hsh ^ bits) * 1099511628211ull;
        }
        key.alpha_hash = hsh;
    }
    auto it = m->graphs.find(key);
    if (it == m->graphs.end()) {
        if (m->graphs.size() >= 16) {                 // bounded cache: drop everything, re-capture
            // a replay may still be in flight on the caller's stream (or on another one the caller
            // used earlier): wait for the device before destroying executable graphs
            HIP_TRY(hipDeviceSynchronize());
            for (auto& kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
            m->graphs.clear();
        }
        // capture on a private stream: the caller's stream may be the null stream, which
        // cannot be captured; nothing executes during capture.
        if (!m->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeRelaxed));
        rc = enqueue_all(m->cap_stream);
        hipError_t e = hipStreamEndCapture(m->cap_stream, &graph);
        if (rc != DAD_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) return fail(DAD_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) return fail(DAD_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
        it = m->graphs.emplace(key, exec).first;
    }
    HIP_TRY(hipGraphLaunch(it->second, st));
    return DAD_OK;
}

// ---------------------------------------------------------------------------------- training
int dad_model_set_training(dad_model* m, int32_t on) {
    if (!m) return fail(DAD_E_INVALID, "null model");
    if (on) if (const char* why = training_refusal(*m)) return fail(DAD_E_INVALID, "training: %s", why);
    if ((on != 0) != m->training) m->finalized = false;       // the data-gradient images are packed at finalize
    m->training = on != 0;
    return DAD_OK;
}

int dad_model_refresh_weights(dad_model* m, int32_t n, const char* const* keys, const float* const* tensors,
                              dad_stream_t stream) {
    if (!m || n < 0 || (n > 0 && (!keys || !tensors))) return fail(DAD_E_INVALID, "bad argument");
    if (!m->finalized) return fail(DAD_E_STATE, "dad_model_finalize has not been called");
    if (m->precision != DAD_PREC_FP32)
        return fail(DAD_E_STATE, "the split-f16 images are scaled per layer on the host: load the weights and finalize again");
    hipStream_t st = (hipStream_t)stream;
    // A training loop refreshes after every optimiser step: the ~70 images and ~110 small tensors of a PointMaze
    // net were ~200 launches of ~5 us; they are collected here and go out as one repack launch (descriptor table in
    // device memory, re-uploaded only when a source address changed) and one copy launch per COPY_MAX tensors.
    std::vector<dad::RepackParams> reps;
    std::vector<std::tuple<float*, const float*, size_t>> copies;
    auto repack = [&](float* dst, const ConvOp& op, int mode, const float* w, const float* ride, int CO, int CI, int K,
                      int c_lo, int c_n) -> int {
        dad::RepackParams p{};
        p.dst = dst; p.w = w; p.ride = ride;
        p.kg = op.bdir ? 16 : std::min(op.kc, 16);
        p.wtaps = op.wtaps(); p.M = op.M;
        p.n = (long)op.cin_pad * p.wtaps * op.M;
        p.mode = mode; p.CO = CO; p.CI = CI; p.K = K; p.c_lo = c_lo; p.c_n = c_n;
        reps.push_back(p);
        return DAD_OK;
    };
    auto copy = [&](float* dst, const float* src, size_t floats) -> int {
        if (floats >= (1u << 31)) return fail(DAD_E_INVALID, "refresh: a tensor of %zu floats", floats);
        copies.emplace_back(dst, src, floats);
        return DAD_OK;
    };
    std::map<std::string, const float*> given;
    for (int i = 0; i < n; ++i) {
        if (!keys[i] || !tensors[i]) return fail(DAD_E_INVALID, "null key or tensor at index %d", i);
        if (!m->expected.count(keys[i])) return fail(DAD_E_KEY, "unexpected key '%s'", keys[i]);
        given[keys[i]] = tensors[i];
    }
    auto has = [&](const std::string& k) -> const float* { auto it = given.find(k); return it == given.end() ? nullptr : it->second; };
    bool tables_dirty = false;
    int rc;
    for (auto& kv : given)
        if (kv.first.find("time_mlp.") != std::string::npos) {
            const auto& shape = m->expected[kv.first];
            size_t fl = 1;
            for (int64_t d : shape) fl *= (size_t)d;
            if ((rc = copy(m->d_time.at(kv.first), kv.second, fl)) != DAD_OK) return rc;
            tables_dirty = true;
        }
    std::vector<ConvOp>& convs = m->plan.convs;
    for (size_t i = 0; i < convs.size(); ++i) {
        ConvOp& op = convs[i];
        const int cin = op.cin0 + op.cin1;
        const float* w = has(op.name + ".weight");
        const float* rw = op.ride ? has(op.rname + ".weight") : nullptr;
        if (w || rw) {
            // the image holds this conv's taps and, when a 1x1 residual conv rides along, that conv's weights
            // as an extra tap: both are needed to rebuild it
            if (op.ride && (!w || !rw))
                return fail(DAD_E_KEY, "'%s.weight' and '%s.weight' share one packed image: refresh them together",
                            op.name.c_str(), op.rname.c_str());
            rc = op.kind == CONV_UP ? repack(op.d_w, op, dad::RP_FWD_UP, w, nullptr, op.cout, cin, 4, 0, 0)
                                    : repack(op.d_w, op, dad::RP_FWD, w, rw, op.cout, cin, op.taps, 0, 0);
            if (rc != DAD_OK) return rc;
            if (m->training && w) {
                const HostModel::BwdConv& b = m->bconvs[i];
                for (int k = 0; k < b.n; ++k) {
                    const ConvOp& bo = b.op[k];
                    const int mode = op.kind == CONV_DOWN ? dad::RP_BWD_DOWN : op.kind == CONV_UP ? dad::RP_BWD_UP : dad::RP_BWD_CONV;
                    if ((rc = repack(bo.d_w, bo, mode, w, nullptr, op.cout, cin, op.taps, b.c_lo[k], b.c_n[k])) != DAD_OK) return rc;
                }
            }
        }
        if (const float* bsrc = has(op.name + ".bias")) {
            if ((rc = copy(op.d_bias, bsrc, op.cout)) != DAD_OK) return rc;
            if (op.kind == CONV_UP && (rc = copy(op.d_bias + op.cout, bsrc, op.cout)) != DAD_OK) return rc;
        }
        if (op.ride)
            if (const float* rb = has(op.rname + ".bias"))
                if ((rc = copy(op.d_rbias, rb, op.cout)) != DAD_OK) return rc;
        if (!op.norm.empty()) {
            if (const float* g = has(op.norm + ".weight")) if ((rc = copy(op.d_gamma, g, op.cout)) != DAD_OK) return rc;
            if (const float* be = has(op.norm + ".bias")) if ((rc = copy(op.d_beta, be, op.cout)) != DAD_OK) return rc;
        }
    }
    if (const float* fw = has("final_conv.1.weight")) {
        if ((rc = copy(m->d_final_w, fw, (size_t)m->cfg.transition_dim * m->cfg.dim)) != DAD_OK) return rc;
        if (m->training &&
            (rc = repack(m->bfinal.d_w, m->bfinal, dad::RP_BWD_FINAL, fw, nullptr, m->cfg.transition_dim, m->cfg.dim, 1, 0, 0)) != DAD_OK)
            return rc;
    }
    if (const float* fb = has("final_conv.1.bias"))
        if ((rc = copy(m->d_final_b, fb, m->cfg.transition_dim)) != DAD_OK) return rc;
    for (size_t at = 0; at < copies.size(); at += dad::COPY_MAX) {
        dad::CopyMany cm{};
        const int k = (int)std::min<size_t>(dad::COPY_MAX, copies.size() - at);
        size_t widest = 1;
        for (int i = 0; i < k; ++i) {
            cm.dst[i] = std::get<0>(copies[at + i]); cm.src[i] = std::get<1>(copies[at + i]);
            cm.n[i] = (int32_t)std::get<2>(copies[at + i]);
            widest = std::max(widest, std::get<2>(copies[at + i]));
        }
        const unsigned gx = (unsigned)std::min<size_t>(64, (widest + 1023) / 1024);
        hipLaunchKernelGGL(dad::copy_many_kernel, dim3(gx, (unsigned)k), dim3(256), 0, st, cm);
        HIP_TRY(hipGetLastError());
    }
    if (!reps.empty()) {
        std::vector<int> first(reps.size() + 1, 0);
        for (size_t i = 0; i < reps.size(); ++i) {
            const long blocks = (reps[i].n + 255) / 256;
            if (first[i] + blocks >= (1L << 31)) return fail(DAD_E_INVALID, "refresh: too many image elements for one launch");
            first[i + 1] = first[i] + (int)blocks;
        }
        const size_t desc_bytes = reps.size() * sizeof(dad::RepackParams), first_bytes = first.size() * sizeof(int);
        const bool same = m->repack_host.size() == desc_bytes && std::memcmp(m->repack_host.data(), reps.data(), desc_bytes) == 0;
        if (!same) {
            if (m->repack_cap < desc_bytes + first_bytes) {       // (grows once; freed with the model's other allocations)
                void* a = nullptr;
                HIP_TRY(hipMalloc(&a, desc_bytes + first_bytes));
                m->owned.push_back(a);
                m->d_repack = a; m->repack_cap = desc_bytes + first_bytes;
            }
            HIP_TRY(hipStreamSynchronize(st));                    // a previous refresh may still read the table
            HIP_TRY(hipMemcpy(m->d_repack, reps.data(), desc_bytes, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy((char*)m->d_repack + desc_bytes, first.data(), first_bytes, hipMemcpyHostToDevice));
            m->repack_host.assign((const char*)reps.data(), (const char*)reps.data() + desc_bytes);
        }
        hipLaunchKernelGGL(dad::repack_many_kernel, dim3((unsigned)first.back()), dim3(256), 0, st,
                           (const dad::RepackParams*)m->d_repack, (const int*)((const char*)m->d_repack + desc_bytes), (int)reps.size());
        HIP_TRY(hipGetLastError());
    }
    // the per-timestep tables belong to the sampler: a training loop never reads them, so they are re-derived by the
    // next inference entry point (ensure_tables), not after every optimiser step (14 launches, 0.35 ms on PointMaze)
    if (tables_dirty) m->tables_stale = true;
    if (!m->chain.empty() && (rc = repack_chain(m, st)) != DAD_OK) return rc;
    return DAD_OK;
}

int dad_train_grad_count(const dad_model* m, int32_t* count, int64_t* total_floats) {
    if (!m) return fail(DAD_E_INVALID, "null model");
    if (count) *count = (int32_t)m->grad_slots.size();
    if (total_floats) *total_floats = m->grad_numel;
    return DAD_OK;
}

int dad_train_grad_info(const dad_model* m, int32_t i, const char** key, int64_t* offset, int64_t* numel) {
    if (!m || i < 0 || i >= (int32_t)m->grad_slots.size()) return fail(DAD_E_INVALID, "gradient slot %d out of range", i);
    if (key) *key = m->grad_slots[i].key.c_str();
    if (offset) *offset = m->grad_slots[i].offset;
    if (numel) *numel = m->grad_slots[i].numel;
    return DAD_OK;
}

}  // extern "C"  (helpers of the training entry points follow)

namespace {

struct WgradGeom { int spc, ksplit, sps, tile, tm, tn; unsigned gx, gy; size_t lds; };
// Block tile: the largest of 64 x 64 / 64 x 32 / 32 x 32 that still gives the layer 32 tiles (the smaller tiles
// split K inside the block instead of over the grid: fewer partial slabs to write and add); the batch is then split
// over blockIdx.z until `target` blocks exist (one block = 8 waves = two per SIMD).
WgradGeom wgrad_geom(int M, int Ctot, int B, int Lg, int Lz, int taps, int pad, int target = 256, bool ragged = false) {
    WgradGeom g{};
    g.spc = std::max(1, dad::WG_ROWS / Lg);
    while (g.spc > 1 && g.spc * dad::wgrad_segz(Lz, taps, pad) > dad::WG_MAX_ZROWS) g.spc /= 2;
    static const int tms[3] = {2, 2, 1}, tns[3] = {2, 1, 1};
    long tiles = 0;
    for (g.tile = ragged ? 2 : 0; g.tile < 3; ++g.tile) {
        g.tm = tms[g.tile]; g.tn = tns[g.tile];
        g.gx = (unsigned)((M + 32 * g.tm - 1) / (32 * g.tm));
        g.gy = (unsigned)((Ctot + 32 * g.tn - 1) / (32 * g.tn));
        tiles = (long)g.gx * g.gy;
        const int kgroups = 8 / (g.tm * g.tn);
        if ((tiles >= 32 && (g.spc * Lg) % (4 * kgroups) == 0) || g.tile == 2) break;
    }
    const int chunks = (B + g.spc - 1) / g.spc;
    int want = (int)std::max(1L, target / tiles);
    want = std::min(want, chunks);
    g.sps = (chunks + want - 1) / want * g.spc;                // samples per split: whole chunks
    g.ksplit = (B + g.sps - 1) / g.sps;
    g.lds = dad::wgrad_lds_floats(g.spc, Lg, Lz, taps, pad, g.tm, g.tn) * sizeof(float);
    if (ragged) g.tile = 3;
    return g;
}

// scratch of dad_unet_backward, in floats: gradient mirror of the training plan | per-sample partial sums
// | wgrad split slabs | padded d x | staging of a down-sampling conv's data gradient | split-K slabs
struct TrainScratch { long mirror, part, wslab, dxpad, tmp, bslab, total; };
TrainScratch train_scratch(const dad_model& m, int B) {
    TrainScratch t{};
    const Plan& P = m.tplan;
    const int H = m.cfg.horizon, td = m.cfg.transition_dim;
    t.mirror = P.floats_per_sample * (long)B;
    t.part = 0;                                            // every layer's per-sample partial sums, side by side
    for (const ConvOp& f : P.convs) t.part += (f.norm.empty() ? 1L : 3L) * B * round_up(f.cout, 4);
    t.part += (long)B * round_up(td, 4);
    long ws = 0, tmp = 0, bs = 0;
    auto wg = [&](int M, int C0, int C1, int Lg, int Lz, int taps, int pad, long numel) {
        const WgradGeom g = wgrad_geom(M, C0 + C1, B, Lg, Lz, taps, pad, m.wgrad_blocks, ((M | C0 | C1) & 3) != 0);
        if (g.ksplit > 1) ws = std::max(ws, (long)g.ksplit * numel);
    };
    for (size_t i = 0; i < P.convs.size(); ++i) {
        const ConvOp& f = P.convs[i];
        const int cin = f.cin0 + f.cin1;
        switch (f.kind) {
            case CONV_K5: case CONV_1X1: wg(f.cout, f.cin0, f.cin1, f.Lin, f.Lin, f.taps, f.taps / 2, (long)f.cout * cin * f.taps); break;
            case CONV_DOWN: wg(f.cout, cin, 0, f.Lout, f.Lin, 3, 1, (long)f.cout * cin * 3);
                tmp = std::max(tmp, (long)B * f.Lin * cin); break;
            case CONV_UP: wg(cin, f.cout, 0, f.Lin, 2 * f.Lin, 4, 1, (long)cin * f.cout * 4); break;
        }
        for (int k = 0; k < m.bconvs[i].n; ++k) {
            const ConvOp& b = m.bconvs[i].op[k];
            const int cfg = choose_tile(m, b, B);
            if (cfg >= 0) bs = std::max(bs, plan_split(m, b, cfg, B).slab_floats);
        }
    }
    wg(td, m.cfg.dim, 0, H, H, 1, 0, (long)td * m.cfg.dim);
    {
        const int cfg = choose_tile(m, m.bfinal, B);
        if (cfg >= 0) bs = std::max(bs, plan_split(m, m.bfinal, cfg, B).slab_floats);
    }
    t.wslab = ws; t.tmp = tmp; t.bslab = bs;
    t.dxpad = (long)B * H * round_up(td, 32);
    if (m.real_horizon > 0 && m.real_horizon != H) t.dxpad += 2L * B * H * round_up(td, 4);      // zero-padded copies of x and d out
    auto al = [](long v) { return (v + 63) / 64 * 64; };
    t.mirror = al(t.mirror); t.part = al(t.part); t.wslab = al(t.wslab); t.dxpad = al(t.dxpad);
    t.tmp = al(t.tmp); t.bslab = al(t.bslab);
    t.total = t.mirror + t.part + t.wslab + t.dxpad + t.tmp + t.bslab;
    return t;
}

size_t train_saved_bytes(const dad_model& m, int B) {
    return ((size_t)m.tplan.floats_per_sample * (size_t)B + (size_t)slab_floats_for(m, B)) * sizeof(float);
}

// dst[b][l][c] (l < Hp) = l < Hr ? src[b][l][c] : 0      (the trajectory into the zero-padded layout)
__global__ void pad_rows_kernel(float* dst, const float* src, long B, int Hp, int Hr, int cols) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Hp * cols) return;
    const int c = (int)(i % cols);
    const long r = i / cols;
    const int l = (int)(r % Hp);
    const long b = r / Hp;
    dst[i] = l < Hr ? src[(b * Hr + l) * cols + c] : 0.0f;
}
// dst[b][l][c] (l < Hr, c < cols) = src[(b * Hp + l) * ld + c]
__global__ void slice_rows_cols_kernel(float* dst, const float* src, long B, int Hp, int Hr, int cols, int ld) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Hr * cols) return;
    const int c = (int)(i % cols);
    const long r = i / cols;
    const int l = (int)(r % Hr);
    const long b = r / Hr;
    dst[i] = src[(b * Hp + l) * ld + c];
}
__global__ void slice_cols_kernel(float* dst, const float* src, long rows, int cols, int ld) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const long r = i / cols;
    dst[i] = src[r * ld + (i - r * cols)];
}

int check_train(const dad_model* m, int batch) {
    if (!m) return fail(DAD_E_INVALID, "null model");
    if (!m->training) return fail(DAD_E_STATE, "dad_model_set_training(m, 1) has not been called");
    if (!m->finalized) return fail(DAD_E_STATE, "dad_model_finalize has not been called");
    if (batch <= 0) return fail(DAD_E_INVALID, "batch must be positive (got %d)", batch);
    return DAD_OK;
}

}  // namespace

extern "C" {

int dad_train_workspace_bytes(const dad_model* m, int32_t batch, size_t* saved_bytes, size_t* scratch_bytes) {
    if (!m || batch <= 0) return fail(DAD_E_INVALID, "bad argument");
    if (saved_bytes) *saved_bytes = train_saved_bytes(*m, batch);
    if (scratch_bytes) *scratch_bytes = (size_t)train_scratch(*m, batch).total * sizeof(float);
    return DAD_OK;
}

int dad_unet_forward_train(dad_model* m, const float* x, const int32_t* row_index, const float* temb_rows,
                           float* out, int32_t batch, void* saved, size_t saved_bytes, dad_stream_t stream) {
    int rc = check_train(m, batch);
    if (rc != DAD_OK) return rc;
    if (!x || !row_index || !temb_rows || !out || !saved) return fail(DAD_E_INVALID, "null pointer");
    if (saved_bytes < train_saved_bytes(*m, batch))
        return fail(DAD_E_WORKSPACE, "saved-activation buffer has %zu bytes, batch %d needs %zu", saved_bytes, batch,
                    train_saved_bytes(*m, batch));
    hipStream_t st = (hipStream_t)stream;
    if ((rc = run_unet(m, x, 0, batch, (float*)saved, st, row_index, nullptr, true, temb_rows)) != DAD_OK) return rc;
    return run_final(m, nullptr, x, 0, batch, nullptr, 1, out, (float*)saved, st, false, nullptr, true);
}

int dad_unet_backward(dad_model* m, const float* x, const float* d_out, float* d_x, float* d_temb_rows,
                      float* const* grad_tensors, int32_t n_grad_tensors,
                      int32_t batch, void* saved_v, size_t saved_bytes, void* scratch_v, size_t scratch_bytes,
                      dad_stream_t stream) {
    int rc = check_train(m, batch);
    if (rc != DAD_OK) return rc;
    if (!x || !d_out || !d_temb_rows || !grad_tensors || !saved_v || !scratch_v) return fail(DAD_E_INVALID, "null pointer");
    if (n_grad_tensors != (int32_t)m->grad_slots.size())
        return fail(DAD_E_INVALID, "%d gradient tensors passed, the model has %zu (dad_train_grad_count)", n_grad_tensors,
                    m->grad_slots.size());
    for (int32_t i = 0; i < n_grad_tensors; ++i)
        if (!grad_tensors[i]) return fail(DAD_E_INVALID, "gradient tensor %d ('%s') is null", i, m->grad_slots[i].key.c_str());
    const int B = batch;
    const TrainScratch ts = train_scratch(*m, B);
    if (saved_bytes < train_saved_bytes(*m, B) || scratch_bytes < (size_t)ts.total * sizeof(float))
        return fail(DAD_E_WORKSPACE, "backward workspaces too small (saved %zu / %zu, scratch %zu / %zu bytes)", saved_bytes,
                    train_saved_bytes(*m, B), scratch_bytes, (size_t)ts.total * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    const Plan& P = m->tplan;
    const std::vector<ConvOp>& convs = P.convs;
    const dad_cfg& c = m->cfg;
    const int H = c.horizon, td = c.transition_dim, tdp = round_up(td, 32);
    float* const saved = (float*)saved_v;
    float* const mirror = (float*)scratch_v;
    float* const part = mirror + ts.mirror;
    float* const wslab = part + ts.part;
    float* const dxpad = wslab + ts.wslab;
    float* const tmp = dxpad + ts.dxpad;
    float* const bslab = tmp + ts.tmp;
    auto act = [&](int id) -> float* { return id >= 0 ? saved + P.bufs[id].offset * (long)B : nullptr; };
    std::vector<int> alias(P.bufs.size(), -1);       // gradient of this buffer IS the gradient of that one
    std::vector<char> written(P.bufs.size(), 0);
    auto resolve = [&](int id) { while (alias[id] >= 0) id = alias[id]; return id; };
    auto grd = [&](int id) -> float* { return mirror + P.bufs[resolve(id)].offset * (long)B; };
    std::vector<int> owner(P.bufs.size(), -1);
    for (size_t i = 0; i < convs.size(); ++i) owner[convs[i].dst] = (int)i;
    bool dx_written = false;
    auto G = [&](const std::string& key) -> float* { return grad_tensors[m->grad_index.at(key)]; };
    // zero-padded horizon: the trajectory and d loss / d out arrive in their real shape (B, H_real, td); the pass runs on
    // copies in the padded layout (zero rows behind the real ones), d x goes back through the same row map
    const int Hr = traj_horizon(m);
    if (Hr != H) {
        float* const xpad = dxpad + (long)B * H * tdp;
        float* const dopad = xpad + (long)B * H * round_up(td, 4);
        const long n = (long)B * H * td;
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, xpad, x, (long)B, H, Hr, td);
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dopad, d_out, (long)B, H, Hr, td);
        HIP_TRY(hipGetLastError());
        x = xpad; d_out = dopad;
    }

    // y += x over n floats (n a multiple of 4), or y = x when y holds nothing yet
    auto accumulate = [&](float* y, const float* xs, long n, bool have) -> int {
        if (!have) { HIP_TRY(hipMemcpyAsync(y, xs, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st)); return DAD_OK; }
        const long n4 = n / 4;
        hipLaunchKernelGGL(dad::add_inplace_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, y, xs, n4);
        HIP_TRY(hipGetLastError());
        return DAD_OK;
    };
    // per-sample partial sums of every layer live side by side in `part`; ONE launch at the end reduces them all
    std::vector<std::tuple<float*, const float*, int>> sums;      // (out[C], part[B][C], C)
    long part_used = 0;
    auto part_take = [&](int C) -> float* { float* q = part + part_used; part_used += (long)B * round_up(C, 4); return q; };
    auto bias_grad = [&](float* out, const float* g, int rows_per_sample, int C) -> int {
        float* q = part_take(C);
        hipLaunchKernelGGL(dad::row_partial_sums_kernel, dim3(B), dim3(256), 0, st, q, g, rows_per_sample, C, C);
        HIP_TRY(hipGetLastError());
        sums.emplace_back(out, q, C);
        return DAD_OK;
    };
    auto wgrad = [&](const float* Gp, int ldg, int M, const float* Z0, int C0, const float* Z1, int C1, float* out,
                     int taps, int stride, int pad, int Lg, int Lz) -> int {
        const bool ragged = ((ldg | C0 | C1 | M) & 3) != 0;      // rows that are not whole aligned float4s (ld == width everywhere)
        const WgradGeom g = wgrad_geom(M, C0 + C1, B, Lg, Lz, taps, pad, m->wgrad_blocks, ragged);
        const int kgroups = 8 / (g.tm * g.tn);
        if (g.lds > dad::kLdsBytes || g.spc * Lg > dad::WG_MAX_GROWS || g.spc * dad::wgrad_segz(Lz, taps, pad) > dad::WG_MAX_ZROWS ||
            (g.spc * Lg) % (4 * kgroups) != 0)
            return fail(DAD_E_INVALID, "wgrad: a chunk of %d samples x %d rows does not fit the kernel's staging", g.spc, Lz);
        const void* fn = wgrad_kernel(taps, g.tile);
        if (fn == nullptr) return fail(DAD_E_INVALID, "wgrad: %d taps", taps);
        dad::WgradParams p{};
        p.G = Gp; p.ldg = ldg; p.M = M;
        p.Z0 = Z0; p.ldz0 = C0; p.C0 = C0; p.Z1 = Z1; p.ldz1 = C1; p.C1 = C1;
        p.out_numel = (long)M * (C0 + C1) * taps;
        p.out = g.ksplit > 1 ? wslab : out;
        p.B = B; p.Lg = Lg; p.Lz = Lz; p.lg_shift = ilog2(Lg); p.stride = stride; p.pad = pad;
        p.ksplit = g.ksplit; p.samples_per_split = g.sps; p.spc = g.spc;
        p.zero = m->d_zero;
        const dim3 grid(g.gx, g.gy, (unsigned)g.ksplit);
        void* args[] = {&p};
        HIP_TRY(hipLaunchKernel(fn, grid, dim3(dad::WG_THREADS), args, g.lds, st));
        if (g.ksplit > 1) {
            if (p.out_numel % 4 != 0) return fail(DAD_E_INVALID, "wgrad: %ld gradient elements (not a multiple of 4)", p.out_numel);
            const long n4 = p.out_numel / 4;
            hipLaunchKernelGGL(dad::sum_slabs_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, out, wslab, n4, g.ksplit);
            HIP_TRY(hipGetLastError());
        }
        return DAD_OK;
    };
    // data gradient `bop` of dH into buffer `target` (-2: the trajectory), adding to what is there
    auto dgrad = [&](const ConvOp& bop, const float* dH, int target, long target_floats) -> int {
        float* dst;
        bool have;
        if (target == -2) { dst = dxpad; have = dx_written; dx_written = true; }
        else { dst = grd(target); have = written[resolve(target)]; written[resolve(target)] = 1; }
        ConvIO io;
        io.src0 = dH; io.slab = bslab;
        if (bop.kind == CONV_UP && have) {                 // the interleaving store has no residual operand
            io.dst = tmp;
            int r = launch_conv(m, bop, B, io, st);
            if (r != DAD_OK) return r;
            return accumulate(dst, tmp, target_floats, true);
        }
        io.dst = dst;
        io.res = have ? dst : nullptr;
        return launch_conv(m, bop, B, io, st);
    };

    // ---- final_conv[1] (1x1, dim -> td; forward in final_posterior_kernel)
    {
        if ((rc = bias_grad(G("final_conv.1.bias"), d_out, H, td)) != DAD_OK) return rc;
        if ((rc = wgrad(d_out, td, td, act(P.final_act), c.dim, nullptr, 0, G("final_conv.1.weight"), 1, 1, 0, H, H)) != DAD_OK)
            return rc;
        if ((rc = dgrad(m->bfinal, d_out, P.final_act, (long)B * H * c.dim)) != DAD_OK) return rc;
    }
    for (int i = (int)convs.size() - 1; i >= 0; --i) {
        const ConvOp& f = convs[i];
        if (!written[resolve(f.dst)])
            return fail(DAD_E_STATE, "backward: no gradient reached the output of %s", f.name.c_str());
        const float* gout = grd(f.dst);
        const int out_rows = f.kind == CONV_UP ? 2 * f.Lout : f.Lout;      // rows per sample of the output
        const long out_floats = (long)B * out_rows * f.cout;
        const float* dH = gout;
        if (!f.norm.empty()) {
            if (f.res == -2) {                             // identity residual of the trajectory itself (td == C)
                if ((rc = accumulate(dxpad, gout, out_floats, dx_written)) != DAD_OK) return rc;
                dx_written = true;
            } else if (f.cat0 >= 0) {                      // identity residual over [cat0 | cat1]: each side takes its columns
                const long rows = (long)B * out_rows;
                const int ids[2] = {f.cat0, f.cat1}, cs[2] = {f.cat_c0, f.cat_c1};
                for (int k = 0, off = 0; k < 2; off += cs[k], ++k) {
                    const int r = resolve(ids[k]);
                    const long n4 = rows * (cs[k] / 4);
                    hipLaunchKernelGGL(dad::take_cols_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                                       grd(ids[k]), gout, rows, cs[k], f.cout, off, (int)written[r]);
                    written[r] = 1;
                }
                HIP_TRY(hipGetLastError());
            } else if (f.res >= 0) {
                const int q = owner[f.res];
                const bool conv_out = q >= 0 && convs[q].kind == CONV_1X1 && convs[q].norm.empty();
                if (conv_out) { alias[f.res] = resolve(f.dst); }         // the 1x1 residual conv's output gradient
                else {
                    const int r = resolve(f.res);
                    if ((rc = accumulate(grd(f.res), gout, out_floats, written[r])) != DAD_OK) return rc;
                    written[r] = 1;
                }
            }
            dad::GnBwdParams gp{};
            gp.dA = gout; gp.h = act(f.pre); gp.stats = act(f.stats);
            gp.gamma = f.d_gamma; gp.beta = f.d_beta;
            gp.dH = mirror + P.bufs[f.pre].offset * (long)B;
            gp.part_dgamma = part_take(f.cout); gp.part_dbeta = part_take(f.cout); gp.part_dbias = part_take(f.cout);
            gp.dtemb = f.temb_off >= 0 ? d_temb_rows + f.temb_off : nullptr;
            gp.temb_stride = P.temb_width;
            gp.C = f.cout; gp.L = f.Lout; gp.cpg = f.cout / 8; gp.lreal = f.lreal; gp.cpg_real = f.gn_real;
            gp.B = B;
            {   // one wave per (sample, group) pair while the pair fits its registers, else one block per pair
                const int f4 = gp.cpg / 4 * gp.L;
                const dim3 wgrid((unsigned)((B * 8 + 3) / 4));
                if (f4 <= 64) hipLaunchKernelGGL(dad::gn_mish_bwd_wave_kernel<1>, wgrid, dim3(256), 0, st, gp);
                else if (f4 <= 128) hipLaunchKernelGGL(dad::gn_mish_bwd_wave_kernel<2>, wgrid, dim3(256), 0, st, gp);
                else if (f4 <= 256) hipLaunchKernelGGL(dad::gn_mish_bwd_wave_kernel<4>, wgrid, dim3(256), 0, st, gp);
                else if (f4 <= 512) hipLaunchKernelGGL(dad::gn_mish_bwd_wave_kernel<8>, wgrid, dim3(256), 0, st, gp);
                else if (f4 <= 1024) hipLaunchKernelGGL(dad::gn_mish_bwd_wave_kernel<16>, wgrid, dim3(256), 0, st, gp);
                else hipLaunchKernelGGL(dad::gn_mish_bwd_kernel, dim3(B, 8), dim3(dad::GNB_THREADS), 0, st, gp);
            }
            HIP_TRY(hipGetLastError());
            sums.emplace_back(G(f.norm + ".weight"), gp.part_dgamma, f.cout);
            sums.emplace_back(G(f.norm + ".bias"), gp.part_dbeta, f.cout);
            sums.emplace_back(G(f.name + ".bias"), gp.part_dbias, f.cout);
            dH = gp.dH;
        } else {
            if ((rc = bias_grad(G(f.name + ".bias"), dH, out_rows, f.cout)) != DAD_OK) return rc;
        }
        const float* s0 = f.src0 == -2 ? x : act(f.src0);
        const float* s1 = act(f.src1);
        float* gw = G(f.name + ".weight");
        switch (f.kind) {
            case CONV_K5: case CONV_1X1:
                rc = wgrad(dH, f.cout, f.cout, s0, f.cin0, s1, f.cin1, gw, f.taps, 1, f.taps / 2, f.Lin, f.Lin); break;
            case CONV_DOWN:
                rc = wgrad(dH, f.cout, f.cout, s0, f.cin0, nullptr, 0, gw, 3, 2, 1, f.Lout, f.Lin); break;
            case CONV_UP:
                rc = wgrad(s0, f.cin0, f.cin0, dH, f.cout, nullptr, 0, gw, 4, 2, 1, f.Lin, 2 * f.Lin); break;
        }
        if (rc != DAD_OK) return rc;
        const HostModel::BwdConv& b = m->bconvs[i];
        for (int k = 0; k < b.n; ++k) {
            const int target = k == 0 ? f.src0 : f.src1;
            if ((rc = dgrad(b.op[k], dH, target, (long)B * f.Lin * b.c_n[k])) != DAD_OK) return rc;
        }
    }
    if (part_used > ts.part) return fail(DAD_E_WORKSPACE, "backward: partial sums overran their region (%ld > %ld floats)", part_used, ts.part);
    for (size_t at = 0; at < sums.size(); at += dad::COLS_MAX) {
        dad::ColSumsMany cs{};
        const int n = (int)std::min<size_t>(dad::COLS_MAX, sums.size() - at);
        int widest = 0;
        for (int k = 0; k < n; ++k) {
            cs.out[k] = std::get<0>(sums[at + k]); cs.part[k] = std::get<1>(sums[at + k]); cs.C[k] = std::get<2>(sums[at + k]);
            widest = std::max(widest, cs.C[k]);
        }
        hipLaunchKernelGGL(dad::col_sums_many_kernel, dim3((unsigned)((widest + 31) / 32), (unsigned)n), dim3(256), 0, st, cs, B);
        HIP_TRY(hipGetLastError());
    }
    if (d_x != nullptr) {
        if (!dx_written) return fail(DAD_E_STATE, "backward: no gradient reached the trajectory");
        const long n = (long)B * Hr * td;
        if (Hr != H)
            hipLaunchKernelGGL(slice_rows_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_x, dxpad, (long)B, H, Hr, td, tdp);
        else
            hipLaunchKernelGGL(slice_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_x, dxpad, (long)B * H, td, tdp);
        HIP_TRY(hipGetLastError());
    }
    return DAD_OK;
}

int dad_fill_normal(float* x, int32_t batch, int32_t row_elems, uint64_t seed, uint64_t row_offset,
                    uint64_t draw, dad_stream_t stream) {
    if (!x || batch <= 0 || row_elems <= 0) return fail(DAD_E_INVALID, "bad argument");
    const long n = (long)batch * row_elems;
    hipLaunchKernelGGL(dad::fill_normal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, n, row_offset * (uint64_t)row_elems, draw, seed);
    HIP_TRY(hipGetLastError());
    return DAD_OK;
}

#ifdef DAD_STAMPS
int dad_debug_stamps(void* buf) { g_stamps = (unsigned long long*)buf; return DAD_OK; }
#endif
// 1 when the host's statement of which conv-GEMM kernels exist (kernel_registered) equals the registry
int dad_debug_kernel_table_consistent(void) {
    const KernTable& t = kernel_table();
    size_t hits = 0;
    for (int cfg = 0; cfg < kNumTiles; ++cfg)
        for (int taps = 1; taps <= 7; ++taps)
            for (int stride = 1; stride <= 2; ++stride)
                for (int f = 0; f < 32; ++f) {
                    const bool x3 = f & 1, bdir = f & 2, ragged = f & 4, res = f & 8, padded = f & 16;
                    const bool have = t.count(KernKey(cfg, taps, stride, x3, bdir, ragged, res, padded)) != 0;
                    if (have != kernel_registered(cfg, taps, stride, x3, bdir, ragged, res, padded)) {
                        fail(DAD_E_INVALID, "kernel table mismatch at tile %d taps=%d stride=%d x3=%d bdir=%d ragged=%d res=%d padded=%d (registry %d)",
                             cfg, taps, stride, (int)x3, (int)bdir, (int)ragged, (int)res, (int)padded, (int)have);
                        return 0;
                    }
                    hits += have;
                }
    return hits == t.size() ? 1 : 0;
}

#ifdef DAD_WG_STAMPS
extern "C" int dad_debug_wgrad_stamps(unsigned long long* host32) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(host32, HIP_SYMBOL(dad::g_wg_stamps), 32 * sizeof(unsigned long long)));
    return DAD_OK;
}
#endif
#ifdef DAD_CHAIN_STAMPS
int dad_debug_chain_stamps(unsigned long long* host32) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(host32, HIP_SYMBOL(dad::g_chain_stamps), 32 * sizeof(unsigned long long)));
    return DAD_OK;
}
#endif

int dad_debug_set_tile(dad_model* m, int32_t cfg) {
    if (!m) return fail(DAD_E_INVALID, "null model");
    // cfg >= 100: same, with grid-level split-K disabled (cfg - 100 is the tile, 99 = heuristic)
    m->split_enabled = cfg < 99;
    m->force_tile = cfg >= 99 ? cfg - 100 : cfg;
    return DAD_OK;
}

int dad_debug_set_option(dad_model* m, const char* name, int32_t value) {
    if (!m || !name) return fail(DAD_E_INVALID, "null argument");
    const std::string key(name);
    if (key == "fuse_residual") m->fuse_residual = value != 0;
    else if (key == "xswz") m->xswz_enabled = value != 0;
    else if (key == "xcd_order") m->xcd_order = value != 0;
    else if (key == "split_target") m->split_target = std::max(1, (int)value);
    else if (key == "cc") m->cc_enabled = value != 0;
    else if (key == "cc_max_rows") m->cc_max_rows = std::max(0, (int)value);
    else if (key == "ccw_max_rows") m->ccw_max_rows = std::max(0, (int)value);
    else if (key == "ccw_min_blocks") m->ccw_min_blocks = std::max(1, (int)value);
    else if (key == "ccw_prefer16") m->ccw_prefer16 = value != 0;
    else if (key == "chain") m->chain_enabled = value != 0;
    else if (key == "chain_min_batch") m->chain_min_batch = std::max(1, (int)value);
    else if (key == "wgrad_blocks") m->wgrad_blocks = std::max(1, (int)value);
    else return fail(DAD_E_INVALID, "unknown option '%s'", name);
    // every option changes which launches a captured loop holds, and not all of them are part of the
    // graph key: drop the cache (a replay may still be in flight: wait for the device first)
    if (!m->graphs.empty()) {
        HIP_TRY(hipDeviceSynchronize());
        for (auto& kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
        m->graphs.clear();
    }
    return DAD_OK;
}

int dad_debug_read_table(dad_model* m, int32_t which, int32_t t, float* host_out, int32_t capacity,
                         int32_t* width_out) {
    if (!m || !host_out) return fail(DAD_E_INVALID, "null argument");
    if (!m->finalized) return fail(DAD_E_STATE, "dad_model_finalize has not been called");
    if (t < 0 || t >= m->cfg.n_timesteps)
        return fail(DAD_E_RANGE, "index %d is out of bounds for the schedule of size %d", t, m->cfg.n_timesteps);
    { const int rc = ensure_tables(m, nullptr); if (rc != DAD_OK) return rc; }      // (null stream: the copy below follows it)
    const float* base = nullptr;
    int width = 0;
    switch (which) {
        case DAD_TABLE_SINUSOID: base = m->d_emb; width = m->cfg.dim; break;
        case DAD_TABLE_TIME_MLP: base = m->d_temb; width = m->cfg.time_dim; break;
        case DAD_TABLE_BLOCKS: base = m->d_temb_table; width = m->plan.temb_width; break;
        default: return fail(DAD_E_INVALID, "unknown table %d", which);
    }
    if (width_out) *width_out = width;
    if (capacity < width)
        return fail(DAD_E_INVALID, "table %d has rows of %d floats, the buffer holds %d", which, width, capacity);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_out, base + (size_t)t * width, (size_t)width * sizeof(float), hipMemcpyDeviceToHost));
    return DAD_OK;
}

int dad_debug_mish(const float* in, float* out, int64_t n, dad_stream_t stream) {
    if (!in || !out || n <= 0) return fail(DAD_E_INVALID, "bad argument");
    hipLaunchKernelGGL(dad::mish_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, in, out, (long)n);
    HIP_TRY(hipGetLastError());
    return DAD_OK;
}

int dad_debug_small_batch_plan(dad_model* m, int32_t batch, int32_t* launches_out, int32_t* wide_out) {
    if (!m || batch <= 0) return fail(DAD_E_INVALID, "bad argument");
    const CcPlan cc = cc_plan(*m, batch);
    int launches = 0, wide = 0;
    if (cc.ok)
        for (const CcOp& o : cc.ops) {
            launches += o.launched;
            wide += o.launched && o.wide;
        }
    static const bool trace = getenv("DAD_TRACE_TILES") != nullptr;
    if (trace && !cc.ok) fprintf(stderr, "[dad] batch %d: no small-batch plan (%s)\n", batch, cc.why.c_str());
    if (launches_out) *launches_out = launches;
    if (wide_out) *wide_out = wide;
    return DAD_OK;
}

int dad_profile_enable(dad_model* m, int32_t on) {
    if (!m) return fail(DAD_E_INVALID, "null model");
    m->profile = on != 0;
    m->ev_used = 0;
    m->prof_flops = 0;
    m->prof_launches = 0;
    return DAD_OK;
}

int dad_profile_read(dad_model* m, double* conv_ms, int64_t* conv_launches, double* conv_flops) {
    if (!m) return fail(DAD_E_INVALID, "null model");
    double ms = 0;
    for (size_t i = 0; i < m->ev_used; ++i) {
        HIP_TRY(hipEventSynchronize(m->ev_pool[i].second));
        float d = 0;
        HIP_TRY(hipEventElapsedTime(&d, m->ev_pool[i].first, m->ev_pool[i].second));
        ms += d;
    }
    if (conv_ms) *conv_ms = ms;
    if (conv_launches) *conv_launches = m->prof_launches;
    if (conv_flops) *conv_flops = m->prof_flops;
    m->ev_used = 0;
    m->prof_flops = 0;
    m->prof_launches = 0;
    return DAD_OK;
}

}  // extern "C"
