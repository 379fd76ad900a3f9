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
    std::vector<void*> owned;         // every hipMalloc to free
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
    m->arena = nullptr;
    m->arena_cap = m->arena_used = 0;
    m->d_emb = m->d_temb = m->d_temb_table = nullptr;
    m->d_final_w = m->d_final_b = nullptr;
    m->d_rng = nullptr;
    m->d_counters = nullptr;
    for (auto& op : m->plan.convs) op.d_w = op.d_bias = op.d_gamma = op.d_beta = op.d_rbias = nullptr;
}

// ---------------------------------------------------------------------- kernel registry
// Every conv-GEMM instantiation the planner can ask for, keyed by what plan_launch decides.
using KernFn = void (*)(const ConvParams);
using KernKey = std::tuple<int, int, int, bool, bool, bool, bool>;   // cfg, taps, stride, x3, bdir, ragged, res
using KernTable = std::map<KernKey, KernFn>;

template <int CFG> struct Tile {
    static constexpr int BM = kTiles[CFG].BM, BN = kTiles[CFG].BN, SK = kTiles[CFG].SK, KC = kTiles[CFG].KC;
};

template <int CFG, int TAPS, int STRIDE, bool X3, bool BDIR, bool RES>
void reg_kernel(KernTable& t) {
    using T = Tile<CFG>;
    constexpr int KC = eff_kc(T::KC, T::BM, TAPS, T::SK, X3, BDIR);
    t[KernKey(CFG, TAPS, STRIDE, X3, BDIR, false, RES)] =
        dad::conv_gemm_f32<T::BM, T::BN, T::SK, KC, TAPS, STRIDE, false, X3, BDIR, RES>;
    if constexpr (!BDIR && STRIDE == 1 && (TAPS == 5 || TAPS == 1))    // general staging path
        t[KernKey(CFG, TAPS, STRIDE, X3, BDIR, true, RES)] =
            dad::conv_gemm_f32<T::BM, T::BN, T::SK, KC, TAPS, STRIDE, true, X3, BDIR, RES>;
}
template <int CFG>
void reg_tile(KernTable& t) {
    reg_kernel<CFG, 5, 1, false, false, false>(t);
    reg_kernel<CFG, 3, 2, false, false, false>(t);
    reg_kernel<CFG, 2, 1, false, false, false>(t);
    reg_kernel<CFG, 1, 1, false, false, false>(t);
    if constexpr (Tile<CFG>::KC >= 16) {
        reg_kernel<CFG, 5, 1, false, false, true>(t);      // + the riding 1x1 residual conv
        reg_kernel<CFG, 5, 1, true, false, false>(t);      // split-f16 variants (16-channel granules)
        reg_kernel<CFG, 3, 2, true, false, false>(t);
        reg_kernel<CFG, 2, 1, true, false, false>(t);
        reg_kernel<CFG, 1, 1, true, false, false>(t);
    } else {                                               // wide tile: direct-B kernels of the GroupNorm'd 5-tap convs
        reg_kernel<CFG, 5, 1, true, true, false>(t);
        reg_kernel<CFG, 5, 1, false, true, false>(t);
    }
}
const KernTable& kernel_table() {
    static const KernTable table = [] {
        KernTable t;
        reg_tile<0>(t); reg_tile<1>(t); reg_tile<2>(t); reg_tile<3>(t);
        reg_tile<4>(t); reg_tile<5>(t); reg_tile<6>(t); reg_tile<7>(t);
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
const void* cc_kernel(int taps, bool ride, bool big, int rows) {
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
    for (int taps : {5, 3, 2, 1})
        for (int res = 0; res < 2; ++res)
            for (int ride = 0; ride < 2; ++ride)
                for (int rows : {16, 32})
                    if (const void* k = ccw_kernel(taps, res != 0, ride != 0, rows))
                        HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dad::kLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)dad::final_cc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)dad::kLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)dad::project_kernel<4, 16>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)dad::kLdsBytes));
    HIP_TRY(hipFuncSetAttribute((const void*)dad::project_kernel<1, 16>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)dad::kLdsBytes));
    done.insert(dev);
    return DAD_OK;
}

int run_conv(dad_model* m, const ConvOp& op, const float* xext, float* ws, int batch, int t,
             hipStream_t st, const int32_t* trow = nullptr) {
    auto buf = [&](int id) -> float* {
        return id >= 0 ? ws + m->plan.bufs[id].offset * (long)batch : nullptr;
    };
    if (op.cat0 >= 0) {     // rows of [cat0 | cat1] side by side into the residual buffer
        const long rows = (long)batch * op.Lout;
        const long n4 = rows * ((op.cat_c0 + op.cat_c1) / 4);
        hipLaunchKernelGGL(dad::concat_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                           buf(op.res), buf(op.cat0), buf(op.cat1), rows, op.cat_c0, op.cat_c1);
        HIP_TRY(hipGetLastError());
    }
    LaunchGeom g;
    int rc = plan_launch(*m, op, batch, g);
    if (rc != DAD_OK) return rc;
    ConvParams p{};
    p.src0 = op.src0 == -2 ? xext : buf(op.src0);
    p.src1 = buf(op.src1);
    p.w = op.d_w; p.bias = op.d_bias; p.gamma = op.d_gamma; p.beta = op.d_beta;
    // shared timestep: row t of the table; per-row timesteps: row 0 + trow[b] * stride in the kernel
    p.temb = op.temb_off >= 0 ? m->d_temb_table + (trow ? 0L : (long)t * m->plan.temb_width) + op.temb_off : nullptr;
    p.trow = trow; p.temb_stride = m->plan.temb_width;
    p.res = op.res == -2 ? xext : buf(op.res);       // identity residual of the trajectory itself (td == C)
    p.dst = buf(op.dst);
    p.cin0 = op.cin0; p.cin1 = op.cin1; p.cin_pad = op.cin_pad;
    p.M = op.M; p.cpg = op.norm.empty() ? 0 : op.cout / 8;
    p.B = batch; p.Lin = op.Lin; p.Lout = op.Lout; p.lshift = ilog2(op.Lout);
    p.lshift_in = ilog2(op.Lin);
    p.interleave = op.kind == CONV_UP;
    p.ntiles_n = g.ntiles_n;
    p.xcd_gn = g.xcd_gn; p.xcd_mts = g.xcd_mts; p.xcd_ntn = g.xcd_ntn;
    p.kslices = g.split.kslices;
    p.chunks_per_slice = g.split.chunks_per_slice;
    p.slab = ws + m->plan.floats_per_sample * (long)batch;     // scratch behind the activations
    p.counters = m->d_counters;
    p.c1 = op.c1; p.c2 = op.c2;
    p.xswz = g.xswz;
    p.wtaps = op.wtaps();
    p.rbias = g.fused ? op.d_rbias : nullptr;
    p.rdst = g.fused ? buf(op.rdst) : nullptr;
    static const bool trace = getenv("DAD_TRACE_TILES") != nullptr;     // tuning aid
    if (trace)
        fprintf(stderr, "[dad] %-34s B=%d M=%d K=%dx%d L=%d tile=%d (%dx%d SK%d) kslices=%d%s\n", op.name.c_str(),
                batch, op.M, op.taps, op.cin0 + op.cin1, op.Lout, g.cfg, kTiles[g.cfg].BM, kTiles[g.cfg].BN,
                kTiles[g.cfg].SK, g.split.kslices, g.fused ? " +res1x1" : "");
#ifdef DAD_STAMPS
    p.stamps = g_stamps ? g_stamps + (size_t)(&op - &m->plan.convs[0]) * 4096 * 8 : nullptr;
#endif
    const auto& table = kernel_table();
    const auto it = table.find(KernKey(g.cfg, op.taps, op.stride, op.x3, op.bdir, g.ragged, g.fused));
    if (it == table.end())
        return fail(DAD_E_INVALID, "no kernel for %s (tile %d taps=%d stride=%d x3=%d bdir=%d ragged=%d res=%d)",
                    op.name.c_str(), g.cfg, op.taps, op.stride, (int)op.x3, (int)op.bdir, (int)g.ragged, (int)g.fused);
    void* args[] = {&p};
    HIP_TRY(hipLaunchKernel((const void*)it->second, dim3(g.gx, g.gy, g.gz), dim3(g.threads), args,
                            g.lds_bytes, st));
    return DAD_OK;
}

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
    const void* kern = shape_ok ? cc_kernel(op.taps, op.ride, big, o.tile_rows) : nullptr;
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
             const int32_t* trow = nullptr, const CcPlan* cc = nullptr) {
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
    const std::vector<ConvOp>& convs = m->plan.convs;
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
    for (const ConvOp& op : convs) {
        // a residual 1x1 conv whose block's first conv carries it at this batch is not launched
        const bool rides = op.rider_of >= 0 && fused_at(*m, convs[op.rider_of], batch);
        if (m->profile) m->prof_flops += op.flops_per_sample * batch;
        if (rides) continue;
        const int rc = run_conv(m, op, x, ws, batch, t, st, trow);
        if (rc != DAD_OK) return rc;
        if (m->profile) ++m->prof_launches;
    }
    if (m->profile) HIP_TRY(hipEventRecord(e1, st));
    return DAD_OK;
}

int run_final(dad_model* m, float* x, const float* x_ro, int t, int batch, const dad_step_args* a,
              int x_out_disabled, float* eps_only, float* ws, hipStream_t st,
              bool seed_from_device = false, const CcPlan* cc = nullptr) {
    const dad_cfg& c = m->cfg;
    dad::FinalParams p{};
    p.act = ws + m->plan.bufs[m->plan.final_act].offset * (long)batch;
    p.w = m->d_final_w; p.bias = m->d_final_b;
    p.dim = c.dim; p.td = c.transition_dim; p.B = batch; p.H = c.horizon;
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
        p.elem_offset = a->row_offset * (uint64_t)c.horizon * (uint64_t)c.transition_dim;
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
    const long N = (long)batch * c.horizon;
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
    // rows per block: one while the batch fits one wave of blocks (every CU streams P once),
    // four beyond that (P is then re-used by four rows per pass) if four rows fit LDS
    const size_t row_lds = (size_t)(1 + 16) * p.D * sizeof(float);    // a row + its 16 partial sets
    const int rb = (batch > 512 && 4 * row_lds <= dad::kLdsBytes) ? 4 : 1;
    const size_t lds = rb * row_lds;
    if (lds > dad::kLdsBytes) return fail(DAD_E_INVALID, "projection dimension D=%d too large", p.D);
    if (rb == 1)
        hipLaunchKernelGGL((dad::project_kernel<1, 16>), dim3(batch), dim3(1024), lds, st, p);
    else
        hipLaunchKernelGGL((dad::project_kernel<4, 16>), dim3((batch + 3) / 4), dim3(1024), lds, st, p);
    HIP_TRY(hipGetLastError());
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
    if (const char* v = getenv("DAD_CC_MAX_ROWS")) m->cc_max_rows = std::max(0, atoi(v));
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
    if ((rc = upload(m, m->raw["final_conv.1.weight"].data, &m->d_final_w)) != DAD_OK) return rc;
    if ((rc = upload(m, m->raw["final_conv.1.bias"].data, &m->d_final_b)) != DAD_OK) return rc;

    // ---- time-embedding tables: every t in [0, T) at once --------------------------------
    const int T = c.n_timesteps, dim = c.dim, tdm = c.time_dim;
    // SinusoidalPosEmb (temporal_unet.py:27-31): the caller's table when one was handed over
    // (dad_model_load_time_embedding: the Python mirror evaluates the reference's own torch
    // expression), else the same formula with the C library's expf/sinf/cosf
    const std::vector<float> emb = m->emb_override.size() == (size_t)T * dim ? m->emb_override
                                                                            : sinusoid_table(T, dim);
    float *d_emb, *d_h1, *d_temb, *d_w, *d_b;
    (void)c;
    if ((rc = upload(m, emb, &d_emb)) != DAD_OK) return rc;
    std::vector<float> zeros((size_t)T * 4 * tdm, 0.0f);
    if ((rc = upload(m, zeros, &d_h1)) != DAD_OK) return rc;
    zeros.resize((size_t)T * tdm);
    if ((rc = upload(m, zeros, &d_temb)) != DAD_OK) return rc;
    zeros.assign((size_t)T * std::max(1, m->plan.temb_width), 0.0f);
    if ((rc = upload(m, zeros, &m->d_temb_table)) != DAD_OK) return rc;
    m->d_emb = d_emb; m->d_temb = d_temb;
    auto linear = [&](const float* in, const std::string& key, float* out, int K, int M, int stride,
                      int mish_in) -> int {
        int r;
        if ((r = upload(m, m->raw[key + ".weight"].data, &d_w)) != DAD_OK) return r;
        if ((r = upload(m, m->raw[key + ".bias"].data, &d_b)) != DAD_OK) return r;
        const long total = (long)T * M;
        hipLaunchKernelGGL(dad::table_linear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256),
                           0, st, in, d_w, d_b, out, T, K, M, stride, mish_in);
        HIP_TRY(hipGetLastError());
        return DAD_OK;
    };
    if ((rc = linear(d_emb, "time_mlp.1", d_h1, dim, 4 * tdm, 4 * tdm, 0)) != DAD_OK) return rc;
    if ((rc = linear(d_h1, "time_mlp.3", d_temb, 4 * tdm, tdm, tdm, 1)) != DAD_OK) return rc;
    for (const ConvOp& op : m->plan.convs) {
        if (op.temb_off < 0) continue;
        std::string base = op.name.substr(0, op.name.size() - std::strlen(".blocks.0.block.0"));
        if ((rc = linear(d_temb, base + ".time_mlp.1", m->d_temb_table + op.temb_off, tdm, op.cout,
                         m->plan.temb_width, 1)) != DAD_OK) return rc;
    }
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
    const long step_elems = (long)batch * m->cfg.horizon * m->cfg.transition_dim;

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
            if (proj && (r = run_project(proj, proj_alphas_host[t], x, batch, m->cfg.horizon, st)) != DAD_OK)
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
        key.state_dim = proj->state_dim; key.observation_dim = proj->observation_dim;
        key.action_dim = proj->action_dim;
    }
    key.n_steps = n_steps; key.batch = batch; key.cond_per_row = cond_per_row;
    key.force_tile = m->force_tile;
    key.flags = (m->split_enabled ? 1 : 0) | (m->fuse_residual ? 2 : 0) | (m->xswz_enabled ? 4 : 0) |
                (m->xcd_order ? 8 : 0) | (ccp ? 16 : 0) | (m->split_target << 8);
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
