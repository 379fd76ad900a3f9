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
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dad.h"
#include "conv_gemm.hpp"
#include "pointwise.hpp"

namespace {

thread_local char g_err[1024] = "";
int g_force_tile = -1;      // dad_debug_set_tile: tuning / test hook
bool g_split_enabled = true;
bool g_xswz_enabled = getenv("DAD_NO_XSWZ") == nullptr;     // A/B switch for the LDS row shifts
bool g_xcd_order = getenv("DAD_NO_XCD_ORDER") == nullptr;    // A/B switch for the XCD-aware tile order
#ifdef DAD_STAMPS
unsigned long long* g_stamps = nullptr;
#endif

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(DAD_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                            \
    } while (0)

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

enum ConvKind { CONV_K5 = 0, CONV_1X1 = 1, CONV_DOWN = 2, CONV_UP = 3 };

// One conv-GEMM launch of the plan.  Buffer ids index Plan::bufs; -1 = none,
// -2 = the external (B,H,td) trajectory tensor.
struct ConvOp {
    std::string name;        // weight key prefix, e.g. "downs.0.0.blocks.0.block.0"
    std::string norm;        // GroupNorm key prefix or ""
    ConvKind kind;
    int taps, stride;
    int cin0, cin1, cin_pad;
    int cout;                // real output channels
    int M;                   // GEMM rows (2*cout for CONV_UP)
    int Lin, Lout;           // GEMM per-sample lengths (CONV_UP: Lout == Lin, stores 2*Lin)
    int src0, src1, dst, res;
    int temb_off;            // offset into the per-t table, or -1
    int kc = 16;             // K chunk the weights are packed for (8 when C_out/8 == 256)
    bool x3 = false;         // weights packed as split-f16 images (dad_model_set_precision)
    bool bdir = false;       // x3 on the wide tile: weight fragments go global -> registers
    // identity residual over a channel concat (decoder block whose 2*C_in equals C_out): the two
    // halves are copied side by side into the `res` buffer before this launch
    int cat0 = -1, cat1 = -1, cat_c0 = 0, cat_c1 = 0;
    float c1 = 1.0f, c2 = 0.0f;   // x3: output scales 2^-s and 2^-(s+11)
    // device tensors (owned by the model)
    float* d_w = nullptr;
    float* d_bias = nullptr;
    float* d_gamma = nullptr;
    float* d_beta = nullptr;
    double flops_per_sample = 0;
};

struct Buf {
    long per_sample;   // floats per batch row
    long offset;       // floats per batch row, from workspace start
};

struct Plan {
    std::vector<ConvOp> convs;
    std::vector<Buf> bufs;
    long floats_per_sample = 0;
    int final_act = -1;       // buffer holding final_conv[0] output
    int temb_width = 0;       // sum of C_out over residual blocks
};

constexpr int kMaxSplitTiles = 4096;
struct TileCfg { int BM, BN, SK, KC; };
// Block tile (BM channels x BN positions), SK-way intra-block split-K, K chunk.  Every
// configuration runs 8 waves per block except the last (4 waves, two blocks per CU).
const TileCfg kTiles[] = {
    {32, 64, 4, 32},    // 0: few output tiles -> deepest split-K
    {64, 64, 2, 32},    // 1: the workhorse at batch 256
    {128, 64, 1, 16},   // 2: GroupNorm groups of 128 channels / plentiful tiles
    {256, 32, 1, 8},    // 3: GroupNorm groups of 256 channels (C = 2048)
    {64, 64, 1, 16},    // 4: plentiful tiles, 4 waves
    {32, 64, 2, 16},    // 5: 4 waves, small LDS: several independent blocks per CU
    {32, 64, 1, 16},    // 6: 2 waves
    {64, 64, 2, 16},    // 7: as 1 with the shallower K chunk (two blocks per CU fit)
};


struct GraphKey {
    const void* x; const void* noise; const void* cond; const void* ws; const void* P;
    int n_steps, batch, cond_per_row, pad_ = 0;
    uint64_t row_offset;
    uint64_t alpha_hash;      // projection strengths are baked into the captured launches
    bool operator<(const GraphKey& o) const {
        return std::memcmp(this, &o, sizeof(GraphKey)) < 0;
    }
};

}  // namespace

struct dad_model {
    dad_cfg cfg;
    std::map<std::string, HostTensor> raw;
    std::map<std::string, std::vector<int64_t>> expected;     // key -> shape
    Plan plan;
    bool finalized = false;
    int precision = DAD_PREC_FP32;                             // dad_model_set_precision
    std::vector<float> sched[5];                               // host schedule scalars
    bool have_sched = false;
    // device
    float* d_temb_table = nullptr;    // [T][temb_width]
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

int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }
bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// ----------------------------------------------------------------------------- planning
struct Allocator {
    std::vector<Buf>& bufs;
    std::vector<bool> in_use;
    explicit Allocator(std::vector<Buf>& b) : bufs(b) {}
    int get(long per_sample) {
        int best = -1;
        for (size_t i = 0; i < bufs.size(); ++i)
            if (!in_use[i] && bufs[i].per_sample >= per_sample &&
                (best < 0 || bufs[i].per_sample < bufs[best].per_sample))
                best = (int)i;
        if (best < 0) {
            bufs.push_back({per_sample, 0});
            in_use.push_back(false);
            best = (int)bufs.size() - 1;
        }
        in_use[best] = true;
        return best;
    }
    void put(int id) { if (id >= 0) in_use[id] = false; }
};

void expect(dad_model* m, const std::string& key, std::vector<int64_t> shape) {
    m->expected[key] = std::move(shape);
}

// Emits the launch plan of TemporalUnet.forward (temporal_unet.py:199-241) including the
// reference's always-upsample decoder and unused level-0 skip.
int build_plan(dad_model* m) {
    const dad_cfg& c = m->cfg;
    Plan& P = m->plan;
    Allocator A(P.bufs);
    const int k = c.kernel_size;
    const int tdm = c.time_dim;
    int temb_off = 0;

    auto conv = [&](const std::string& name, const std::string& norm, ConvKind kind, int src0,
                    int src1, int cin0, int cin1, int cout, int Lin, int dst, int res,
                    int toff) {
        ConvOp op;
        op.name = name; op.norm = norm; op.kind = kind;
        op.cin0 = cin0; op.cin1 = cin1;
        op.kc = (!norm.empty() && cout / 8 >= 256) ? 8 : 16;
        const int padto = op.kc == 8 ? 8 : (kind == CONV_1X1 ? 128 : 64);   // deepest K chunk of its kernels
        op.cin_pad = (cin0 + cin1 + padto - 1) / padto * padto;
        op.cout = cout; op.src0 = src0; op.src1 = src1; op.dst = dst; op.res = res;
        op.temb_off = toff; op.Lin = Lin;
        const int cin = cin0 + cin1;
        switch (kind) {
            case CONV_K5: op.taps = k; op.stride = 1; op.M = cout; op.Lout = Lin;
                expect(m, name + ".weight", {cout, cin, k});
                op.flops_per_sample = 2.0 * cout * cin * k * Lin; break;
            case CONV_1X1: op.taps = 1; op.stride = 1; op.M = cout; op.Lout = Lin;
                expect(m, name + ".weight", {cout, cin, 1});
                op.flops_per_sample = 2.0 * cout * cin * Lin; break;
            case CONV_DOWN: op.taps = 3; op.stride = 2; op.M = cout; op.Lout = Lin / 2;
                expect(m, name + ".weight", {cout, cin, 3});
                op.flops_per_sample = 2.0 * cout * cin * 3 * (Lin / 2); break;
            case CONV_UP: op.taps = 2; op.stride = 1; op.M = 2 * cout; op.Lout = Lin;
                expect(m, name + ".weight", {cin, cout, 4});
                op.flops_per_sample = 2.0 * cout * cin * 4 * Lin; break;   // algorithmic
        }
        expect(m, name + ".bias", {cout});
        if (!norm.empty()) {
            expect(m, norm + ".weight", {cout});
            expect(m, norm + ".bias", {cout});
        }
        P.convs.push_back(op);
    };

    auto res_block = [&](const std::string& base, int in0, int in1, int cin0, int cin1, int cout,
                         int L) -> int {
        const int cin = cin0 + cin1;
        const int toff = temb_off;
        temb_off += cout;
        expect(m, base + ".time_mlp.1.weight", {cout, tdm});
        expect(m, base + ".time_mlp.1.bias", {cout});
        const int a0 = A.get((long)cout * L);
        conv(base + ".blocks.0.block.0", base + ".blocks.0.block.1", CONV_K5, in0, in1, cin0, cin1,
             cout, L, a0, -1, toff);
        int res = -1;
        const bool cat_identity = cin == cout && in1 >= 0;   // nn.Identity over torch.cat([x, skip])
        if (cin != cout) {
            res = A.get((long)cout * L);
            conv(base + ".residual_conv", "", CONV_1X1, in0, in1, cin0, cin1, cout, L, res, -1, -1);
        } else if (cat_identity) {
            res = A.get((long)cout * L);
        }
        const int out = A.get((long)cout * L);
        conv(base + ".blocks.1.block.0", base + ".blocks.1.block.1", CONV_K5, a0, -1, cout, 0,
             cout, L, out, res >= 0 ? res : in0, -1);
        if (cat_identity) {
            ConvOp& last = P.convs.back();
            last.cat0 = in0; last.cat1 = in1; last.cat_c0 = cin0; last.cat_c1 = cin1;
        }
        A.put(a0);
        A.put(res);
        return out;
    };

    expect(m, "time_mlp.1.weight", {4 * tdm, c.dim});
    expect(m, "time_mlp.1.bias", {4 * tdm});
    expect(m, "time_mlp.3.weight", {tdm, 4 * tdm});
    expect(m, "time_mlp.3.bias", {tdm});

    const int nl = c.n_levels;
    int L = c.horizon;
    int x = -2, cx = c.transition_dim;
    std::vector<int> skips, skip_ch;
    for (int i = 0; i < nl; ++i) {
        const int co = c.channels[i];
        const std::string b = "downs." + std::to_string(i);
        const int h1 = res_block(b + ".0", x, -1, cx, 0, co, L);
        if (x >= 0) A.put(x);
        const int h2 = res_block(b + ".1", h1, -1, co, 0, co, L);
        A.put(h1);
        skips.push_back(h2);
        skip_ch.push_back(co);
        if (i < nl - 1) {
            const int d = A.get((long)co * (L / 2));
            conv(b + ".2.conv", "", CONV_DOWN, h2, -1, co, 0, co, L, d, -1, -1);
            L /= 2;
            x = d;
            if (i == 0) A.put(h2);   // level-0 skip is pushed but never popped (F8)
        } else {
            x = h2;
        }
        cx = co;
    }
    const int cm = c.channels[nl - 1];
    const bool x_is_skip = true;   // x aliases skips.back() (last level has no downsample)
    const int m1 = res_block("mid_block1", x, -1, cm, 0, cm, L);
    (void)x_is_skip;
    const int m2 = res_block("mid_block2", m1, -1, cm, 0, cm, L);
    A.put(m1);
    x = m2;
    cx = cm;
    for (int j = 0; j < nl - 1; ++j) {
        const int lvl = nl - 1 - j;                 // level whose skip is popped
        const int skip = skips[lvl];
        const int cs = skip_ch[lvl];
        const int co = c.channels[lvl - 1];
        const std::string b = "ups." + std::to_string(j);
        const int u1 = res_block(b + ".0", x, skip, cx, cs, co, L);
        A.put(x);
        A.put(skip);
        const int u2 = res_block(b + ".1", u1, -1, co, 0, co, L);
        A.put(u1);
        const int up = A.get((long)co * (2 * L));
        conv(b + ".2.conv", "", CONV_UP, u2, -1, co, 0, co, L, up, -1, -1);
        A.put(u2);
        L *= 2;
        x = up;
        cx = co;
    }
    if (nl == 1) { /* x == skips[0]; nothing popped */ }
    if (cx != c.dim)
        return fail(DAD_E_INVALID, "final_conv expects %d channels but the decoder ends with %d "
                    "(reference requires dim_mults[0] == 1)", c.dim, cx);
    const int f = A.get((long)c.dim * L);
    conv("final_conv.0.block.0", "final_conv.0.block.1", CONV_K5, x, -1, cx, 0, c.dim, L, f, -1, -1);
    P.final_act = f;
    expect(m, "final_conv.1.weight", {c.transition_dim, c.dim, 1});
    expect(m, "final_conv.1.bias", {c.transition_dim});
    P.temb_width = temb_off;

    long off = 0;
    for (auto& b : P.bufs) {
        b.offset = off;
        off += (b.per_sample + 3) / 4 * 4;
    }
    P.floats_per_sample = off;
    return DAD_OK;
}

// ------------------------------------------------------------------------------ packing
// Conv1d weight (co, ci, k)  ->  [ci_pad/KC][k][M = co][KC]
std::vector<float> pack_conv(const HostTensor& w, int cin_pad, int taps, int kc) {
    const int co = (int)w.shape[0], ci = (int)w.shape[1], k = (int)w.shape[2];
    std::vector<float> out((size_t)cin_pad * taps * co, 0.0f);
    for (int o = 0; o < co; ++o)
        for (int i = 0; i < ci; ++i)
            for (int t = 0; t < k; ++t) {
                const size_t row = ((size_t)(i / kc) * taps + t) * co + o;
                out[row * kc + (i % kc)] = w.data[((size_t)o * ci + i) * k + t];
            }
    return out;
}

// ConvTranspose1d weight (ci, co, 4), stride 2, pad 1:
//   y[co, 2j]   = sum_ci W[ci,co,3] x[ci,j-1] + W[ci,co,1] x[ci,j]
//   y[co, 2j+1] = sum_ci W[ci,co,2] x[ci,j]   + W[ci,co,0] x[ci,j+1]
// packed as a 2-tap conv with M = 2*co columns: columns [0,co) are the even phase (taps at
// positions j-1, j), columns [co,2co) the odd phase (taps at j, j+1 — the kernel shifts the row
// base by one for tiles of that half).
std::vector<float> pack_convT(const HostTensor& w, int cin_pad, int kc) {
    const int ci = (int)w.shape[0], co = (int)w.shape[1];
    const int M = 2 * co;
    std::vector<float> out((size_t)cin_pad * 2 * M, 0.0f);
    auto at = [&](int i, int o, int kk) { return w.data[((size_t)i * co + o) * 4 + kk]; };
    for (int i = 0; i < ci; ++i)
        for (int o = 0; o < co; ++o) {
            auto slot = [&](int tap, int m) -> float& {
                return out[(((size_t)(i / kc) * 2 + tap) * M + m) * kc + (i % kc)];
            };
            slot(0, o) = at(i, o, 3);
            slot(1, o) = at(i, o, 1);
            slot(0, co + o) = at(i, o, 2);
            slot(1, co + o) = at(i, o, 0);
        }
    return out;
}

// Split-f16 image of a packed weight tensor (granules of 16 input channels):
//   [8 words: 16 hi halves | 8 words: 16 lo halves],  w * 2^s ~= hi + lo * 2^-11,
// s chosen per layer so the largest weight lands in [2^9, 2^10) and small ones stay normal halves.
// The kernel reads the words as the 32x32x16 f16 MFMA operand (conv_gemm.hpp, X3).
uint16_t f16_bits(float v) {
    const _Float16 h = (_Float16)v;       // round to nearest even
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
int split_f16_image(std::vector<float>& packed) {
    float amax = 0.0f;
    for (float v : packed) amax = std::max(amax, std::fabs(v));
    int s = 0;
    if (amax > 0.0f && std::isfinite(amax)) {
        int e;
        std::frexp(amax, &e);             // amax = f * 2^e, f in [0.5, 1)
        s = 10 - e;                       // amax * 2^s in [2^9, 2^10)
    }
    s = std::max(-100, std::min(100, s));
    const float up = std::ldexp(1.0f, s);
    for (size_t g = 0; g + 16 <= packed.size(); g += 16) {
        uint16_t hi[16], lo[16];
        for (int j = 0; j < 16; ++j) {
            const float v = packed[g + j] * up;
            const _Float16 h = (_Float16)v;
            hi[j] = f16_bits(v);
            lo[j] = f16_bits((v - (float)h) * 2048.0f);
        }
        std::memcpy(&packed[g], hi, 32);
        std::memcpy(&packed[g + 8], lo, 32);
    }
    return s;
}

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

// Bytes the arena must hold: packed weights, norms, tables, time-MLP weights, flags.
size_t arena_bytes_needed(const dad_model* m) {
    const dad_cfg& c = m->cfg;
    size_t floats = 0, allocs = 0;
    auto add = [&](size_t n) { floats += n + 64; ++allocs; };
    for (const ConvOp& op : m->plan.convs) {
        add((size_t)op.cin_pad * op.taps * op.M);
        add(op.M);
        if (!op.norm.empty()) { add(op.cout); add(op.cout); }
        if (op.temb_off >= 0) { add((size_t)op.cout * c.time_dim); add(op.cout); }
    }
    add((size_t)c.transition_dim * c.dim); add(c.transition_dim);
    const size_t T = c.n_timesteps;
    add(T * c.dim); add(T * 4 * c.time_dim); add(T * c.time_dim);
    add(T * std::max(1, m->plan.temb_width));
    add((size_t)4 * c.time_dim * c.dim); add(4 * c.time_dim);
    add((size_t)c.time_dim * 4 * c.time_dim); add(c.time_dim);
    return floats * sizeof(float) + allocs * 256 + kMaxSplitTiles * sizeof(unsigned) + (1 << 16);
}

void free_device(dad_model* m) {
    for (auto& kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
    m->graphs.clear();
    for (void* p : m->owned) (void)hipFree(p);
    m->owned.clear();
    m->arena = nullptr;
    m->arena_cap = m->arena_used = 0;
    m->d_temb_table = nullptr;
    m->d_final_w = m->d_final_b = nullptr;
    m->d_rng = nullptr;
    m->d_counters = nullptr;
    for (auto& op : m->plan.convs) op.d_w = op.d_bias = op.d_gamma = op.d_beta = nullptr;
}

// ------------------------------------------------------------------------- conv launch
template <int CFG> struct Tile;
template <> struct Tile<0> { static constexpr int BM = 32, BN = 64, SK = 4, KC = 32; };
template <> struct Tile<1> { static constexpr int BM = 64, BN = 64, SK = 2, KC = 32; };
template <> struct Tile<2> { static constexpr int BM = 128, BN = 64, SK = 1, KC = 16; };
template <> struct Tile<3> { static constexpr int BM = 256, BN = 32, SK = 1, KC = 8; };
template <> struct Tile<4> { static constexpr int BM = 64, BN = 64, SK = 1, KC = 16; };
template <> struct Tile<5> { static constexpr int BM = 32, BN = 64, SK = 2, KC = 16; };
template <> struct Tile<6> { static constexpr int BM = 32, BN = 64, SK = 1, KC = 16; };
template <> struct Tile<7> { static constexpr int BM = 64, BN = 64, SK = 2, KC = 16; };

// 1x1 convs have one (tap, group) unit per 8 channels: a deep K chunk keeps enough MFMAs between
// barriers (128 channels; 64 for the 128-row tile, whose stage would not fit LDS twice).
// Split-f16 kernels consume 16 channels per unit: the chunk must give every split-K wave a unit.
constexpr int eff_kc(int cfg_kc, int bm, int taps, int sk = 1, bool x3 = false, bool bd = false) {
    return bd                          ? 32          // wide tile, direct-B kernel (either arithmetic)
           : (taps == 1 && cfg_kc >= 16) ? (bm >= 128 ? 64 : 128)
           : (x3 && cfg_kc < 16 * sk)  ? 16 * sk
                                       : cfg_kc;
}

template <int CFG, int TAPS, int STRIDE, bool X3, bool BDIR = false>
int launch_conv_t(ConvParams& p, hipStream_t st) {
    using T = Tile<CFG>;
    constexpr int KC = eff_kc(T::KC, T::BM, TAPS, T::SK, X3, BDIR);   // BDIR: weight fragments straight from global
    const int cin = p.cin0 + p.cin1;
    const bool ragged = (p.cin0 & 3) != 0 || (p.cin1 & 3) != 0 || p.cin0 % KC != 0 || cin % KC != 0;
    if (ragged && !(STRIDE == 1 && (TAPS == 5 || TAPS == 1)))
        return fail(DAD_E_INVALID, "channel count %d+%d needs the general staging path, which exists "
                    "for stride-1 5-tap and 1x1 convs only", p.cin0, p.cin1);
    if (BDIR && ragged)
        return fail(DAD_E_INVALID, "the direct-B split-f16 kernel needs whole 32-channel chunks (%d+%d)", p.cin0, p.cin1);
    auto kern = BDIR ? dad::conv_gemm_f32<T::BM, T::BN, T::SK, KC, TAPS, STRIDE, false, X3, BDIR>
                : (ragged && STRIDE == 1 && (TAPS == 5 || TAPS == 1))
                    ? dad::conv_gemm_f32<T::BM, T::BN, T::SK, KC, TAPS, 1, true && !BDIR, X3, BDIR>
                    : dad::conv_gemm_f32<T::BM, T::BN, T::SK, KC, TAPS, STRIDE, false, X3, BDIR>;
    const size_t lds = dad::conv_lds_floats(T::BM, T::BN, KC, TAPS, p.Lin, p.Lout, T::SK, BDIR) * sizeof(float);
    const int spt = T::BN / p.Lout;
    p.ntiles_n = (p.B + spt - 1) / spt;
    if (p.ntiles_n > 65535) return fail(DAD_E_INVALID, "batch too large for one launch (%d N tiles)", p.ntiles_n);
    // XCD-aware tile order when every XCD gets the same whole rectangle of tiles: choose the
    // gm x gn arrangement of the 8 XCDs that minimises  gn * (weight bytes) + gm * (activation bytes)
    const int MT = p.M / T::BM, NTn = p.ntiles_n;
    dim3 grid(p.kslices, MT, NTn);
    p.xcd_gn = 0;
    if (g_xcd_order && p.kslices == 1 && (MT & (MT - 1)) == 0 && (long)MT * NTn <= 65535 && ((long)MT * NTn) % 8 == 0) {
        const double wbytes = (double)p.M * TAPS * (p.cin0 + p.cin1);
        const double xbytes = (double)p.B * p.Lin * (p.cin0 + p.cin1);
        double best = -1;
        for (int gm = 1; gm <= 8; gm *= 2) {
            const int gn = 8 / gm;
            if (MT % gm != 0 || NTn % gn != 0) continue;
            const double cost = gn * wbytes + gm * xbytes;
            if (best < 0 || cost < best) {
                best = cost;
                p.xcd_gn = gn; p.xcd_mts = ilog2(MT / gm); p.xcd_ntn = NTn / gn;
            }
        }
        if (p.xcd_gn > 0) grid = dim3(p.kslices, MT * NTn, 1);
    }
    hipLaunchKernelGGL(kern, grid,
                       dim3(64 * (T::BM / 32) * (T::BN / 32) * T::SK), lds, st, p);
    HIP_TRY(hipGetLastError());
    return DAD_OK;
}

// Every kernel may use up to the full 160 KiB of LDS; raise the dynamic-LDS limit once
// (not lazily, so that nothing but launches happens under hipGraph capture).
template <int CFG, int TAPS, int STRIDE, bool X3, bool BDIR = false>
hipError_t raise_lds_limit() {
    using T = Tile<CFG>;
    constexpr int KC = eff_kc(T::KC, T::BM, TAPS, T::SK, X3, BDIR);
    hipError_t e = hipFuncSetAttribute(
        (const void*)dad::conv_gemm_f32<T::BM, T::BN, T::SK, KC, TAPS, STRIDE, false, X3, BDIR>,
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (!BDIR && STRIDE == 1 && (TAPS == 5 || TAPS == 1))
        e = hipFuncSetAttribute(
            (const void*)dad::conv_gemm_f32<T::BM, T::BN, T::SK, KC, TAPS, 1, true && !BDIR, X3, BDIR>,
            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
}
template <int CFG>
hipError_t raise_lds_limit_cfg() {
    hipError_t e;
    if ((e = raise_lds_limit<CFG, 5, 1, false>()) != hipSuccess) return e;
    if ((e = raise_lds_limit<CFG, 3, 2, false>()) != hipSuccess) return e;
    if ((e = raise_lds_limit<CFG, 2, 1, false>()) != hipSuccess) return e;
    if ((e = raise_lds_limit<CFG, 1, 1, false>()) != hipSuccess) return e;
    if constexpr (Tile<CFG>::KC >= 16) {      // split-f16 variants (16-channel granules)
        if ((e = raise_lds_limit<CFG, 5, 1, true>()) != hipSuccess) return e;
        if ((e = raise_lds_limit<CFG, 3, 2, true>()) != hipSuccess) return e;
        if ((e = raise_lds_limit<CFG, 2, 1, true>()) != hipSuccess) return e;
        if ((e = raise_lds_limit<CFG, 1, 1, true>()) != hipSuccess) return e;
    } else {                                   // wide tile: the direct-B kernels of the GroupNorm'd 5-tap convs
        if ((e = raise_lds_limit<CFG, 5, 1, true, true>()) != hipSuccess) return e;
        if ((e = raise_lds_limit<CFG, 5, 1, false, true>()) != hipSuccess) return e;
    }
    return hipSuccess;
}
int configure_kernels() {
    static bool done = false;
    if (done) return DAD_OK;
    HIP_TRY(raise_lds_limit_cfg<0>());
    HIP_TRY(raise_lds_limit_cfg<1>());
    HIP_TRY(raise_lds_limit_cfg<2>());
    HIP_TRY(raise_lds_limit_cfg<3>());
    HIP_TRY(raise_lds_limit_cfg<4>());
    HIP_TRY(raise_lds_limit_cfg<5>());
    HIP_TRY(raise_lds_limit_cfg<6>());
    HIP_TRY(raise_lds_limit_cfg<7>());
    HIP_TRY(hipFuncSetAttribute((const void*)dad::final_posterior_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void*)dad::project_kernel<4, 16>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void*)dad::project_kernel<1, 16>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done = true;
    return DAD_OK;
}

template <int CFG>
int launch_conv_cfg(ConvParams& p, int taps, int stride, bool x3, bool bdir, hipStream_t st) {
    if constexpr (Tile<CFG>::KC == 8) {
        if (bdir && taps == 5 && stride == 1)
            return x3 ? launch_conv_t<CFG, 5, 1, true, true>(p, st) : launch_conv_t<CFG, 5, 1, false, true>(p, st);
    }
    if (bdir) return fail(DAD_E_INVALID, "no direct-B kernel for tile %d taps=%d stride=%d", CFG, taps, stride);
    if constexpr (Tile<CFG>::KC >= 16) {
        if (x3) {
            if (taps == 5 && stride == 1) return launch_conv_t<CFG, 5, 1, true>(p, st);
            if (taps == 3 && stride == 2) return launch_conv_t<CFG, 3, 2, true>(p, st);
            if (taps == 2 && stride == 1) return launch_conv_t<CFG, 2, 1, true>(p, st);
            if (taps == 1 && stride == 1) return launch_conv_t<CFG, 1, 1, true>(p, st);
        }
    }
    if (x3) return fail(DAD_E_INVALID, "no split-f16 kernel for tile %d taps=%d stride=%d", CFG, taps, stride);
    if (taps == 5 && stride == 1) return launch_conv_t<CFG, 5, 1, false>(p, st);
    if (taps == 3 && stride == 2) return launch_conv_t<CFG, 3, 2, false>(p, st);
    if (taps == 2 && stride == 1) return launch_conv_t<CFG, 2, 1, false>(p, st);
    if (taps == 1 && stride == 1) return launch_conv_t<CFG, 1, 1, false>(p, st);
    return fail(DAD_E_INVALID, "unsupported conv taps=%d stride=%d", taps, stride);
}

// Tile choice.  Hard constraints: the tile holds whole GroupNorm groups (BM % (C/8) == 0) and
// whole samples (BN % L == 0), BM divides the columns (each phase half for the transposed
// conv), the K chunk matches the packed weights.  Preference: enough blocks to cover the 256
// CUs; when tiles are scarce, trade tile size for split-K depth.
int choose_tile(const ConvOp& op, int batch) {
    const int Mrows = op.kind == CONV_UP ? op.M / 2 : op.M;
    const int cpg = op.norm.empty() ? 1 : op.cout / 8;
    auto valid = [&](int cfg) {
        const TileCfg& t = kTiles[cfg];
        if ((t.KC == 8) != (op.kc == 8)) return false;
        if (Mrows % t.BM != 0) return false;
        if (!op.norm.empty() && (t.BM % cpg != 0)) return false;
        if (t.BN % op.Lout != 0) return false;
        const int nthreads = 64 * (t.BM / 32) * (t.BN / 32) * t.SK;
        const int f4pl = t.BM * t.BN / 4 / nthreads;
        if (!op.norm.empty() && op.Lout * cpg / 4 < f4pl) return false;   // >= 1 lane per (group, sample)
        return true;
    };
    auto blocks = [&](int cfg) {
        const TileCfg& t = kTiles[cfg];
        const int spt = t.BN / op.Lout;
        return (long)((batch + spt - 1) / spt) * (op.M / t.BM);
    };
    if (op.kc == 8) return valid(3) ? 3 : -1;
    if (g_force_tile >= 0 && g_force_tile < 8 && valid(g_force_tile)) return g_force_tile;
    if (valid(2) && blocks(2) >= 512) return 2;          // plentiful work: big tile
    if (valid(1) && blocks(1) >= 224) return 1;
    if (valid(0)) return 0;
    if (valid(1)) return 1;
    if (valid(2)) return 2;
    if (valid(4)) return 4;
    return -1;
}

// Per-sample slot shifts of the X stage (conv_gemm.hpp, "Activation rows in LDS and bank
// conflicts").  Depth-first over the samples of a block tile: d(s) in [0, 16) such that in every
// 32-row wave tile both 16-lane groups of ds_read_b128 see 16 distinct slots, and no sample is
// pushed onto its neighbour's real rows (d(s) - d(s+1) <= pad * slots-per-row).  Returns 0 (plain
// layout — correct, just slower) when L >= 32, when there is no halo, or when nothing is found.
uint64_t find_xswz(int L, int stride, int pad, int kp4, int BN) {
    if (L >= 32 || pad == 0 || BN / L > 16) return 0;
    static std::map<std::vector<int>, uint64_t> cache;
    static std::mutex cache_lock;
    std::lock_guard<std::mutex> hold(cache_lock);
    const std::vector<int> key{L, stride, pad, kp4, BN};
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    static const int groups[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const int S = BN / L, seg = L * stride + 2 * pad, per = 32 / L;
    std::vector<int> d(S, 0);
    // conflicts among the lanes whose samples are already placed (samples < upto)
    auto ok_prefix = [&](int upto) {
        for (int tn = 0; tn * per < upto; ++tn)
            for (const auto& g : groups) {
                unsigned seen = 0;
                for (int lane : g) {
                    const int n = tn * 32 + lane, sm = n / L, l = n % L;
                    if (sm >= upto) continue;
                    const unsigned bit = 1u << (((sm * seg + l * stride) * kp4 + d[sm]) & 15);
                    if (seen & bit) return false;
                    seen |= bit;
                }
            }
        return true;
    };
    long budget = 300000;                           // node budget: the search is a one-off per shape
    std::function<bool(int)> place = [&](int sm) -> bool {
        if (sm == S) return true;
        for (int v = 0; v < 16; ++v) {
            if (--budget < 0) return false;
            if (sm > 0 && d[sm - 1] - v > pad * kp4) continue;
            d[sm] = v;
            if (ok_prefix(sm + 1) && place(sm + 1)) return true;
        }
        d[sm] = 0;
        return false;
    };
    uint64_t packed = 0;
    if (place(0))
        for (int sm = 0; sm < S; ++sm) packed |= (uint64_t)d[sm] << (4 * sm);
    cache[key] = packed;
    return packed;
}

// Grid-level split-K: when a layer has too few output tiles to cover the chip (small batches;
// the deepest levels of the wide nets), several blocks share a tile and split its K chunks.
struct SplitPlan { int kslices, chunks_per_slice; long slab_floats; };
SplitPlan plan_split(const ConvOp& op, int cfg, int batch) {
    const TileCfg& t = kTiles[cfg];
    const int spt = t.BN / op.Lout;
    const long tiles = (long)((batch + spt - 1) / spt) * (op.M / t.BM);
    const int kc = eff_kc(t.KC, t.BM, op.taps, t.SK, op.x3, op.bdir);
    const int nchunks = (op.cin0 + op.cin1 + kc - 1) / kc;      // chunks holding real channels
    SplitPlan sp{1, nchunks, 0};
    if (tiles >= 160 || nchunks < 2 || tiles > kMaxSplitTiles) return sp;
    static const int target = getenv("DAD_SPLIT_TARGET") ? atoi(getenv("DAD_SPLIT_TARGET")) : 256;   // tuning aid
    int want = (int)((target + tiles - 1) / tiles);
    if (want > nchunks) want = nchunks;
    if (want < 2) return sp;
    sp.chunks_per_slice = (nchunks + want - 1) / want;
    sp.kslices = (nchunks + sp.chunks_per_slice - 1) / sp.chunks_per_slice;
    sp.slab_floats = tiles * sp.kslices * (long)t.BN * t.BM;
    return sp;
}

int choose_tile(const ConvOp& op, int batch);

// floats of split-K scratch a batch needs (max over layers)
long slab_floats_for(const dad_model* m, int batch) {
    long best = 0;
    for (const ConvOp& op : m->plan.convs) {
        const int cfg = choose_tile(op, batch);
        if (cfg < 0) continue;
        best = std::max(best, plan_split(op, cfg, batch).slab_floats);
    }
    return best;
}

int run_conv(dad_model* m, const ConvOp& op, const float* xext, float* ws, int batch, int t,
             hipStream_t st) {
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
    ConvParams p{};
    p.src0 = op.src0 == -2 ? xext : buf(op.src0);
    p.src1 = buf(op.src1);
    p.w = op.d_w; p.bias = op.d_bias; p.gamma = op.d_gamma; p.beta = op.d_beta;
    p.temb = op.temb_off >= 0 ? m->d_temb_table + (long)t * m->plan.temb_width + op.temb_off : nullptr;
    p.res = op.res == -2 ? nullptr : buf(op.res);
    p.dst = buf(op.dst);
    p.cin0 = op.cin0; p.cin1 = op.cin1; p.cin_pad = op.cin_pad;
    p.M = op.M; p.cpg = op.norm.empty() ? 0 : op.cout / 8;
    p.B = batch; p.Lin = op.Lin; p.Lout = op.Lout; p.lshift = ilog2(op.Lout);
    p.lshift_in = ilog2(op.Lin);
    p.interleave = op.kind == CONV_UP;
    if ((long)batch * op.Lout * op.M * (op.kind == CONV_UP ? 1 : 1) >= (1L << 31) ||
        (long)batch * op.Lin * (op.cin0 + op.cin1) >= (1L << 31))
        return fail(DAD_E_INVALID, "batch %d too large: a layer's activation tensor exceeds 2^31 elements", batch);
    const int cfg = choose_tile(op, batch);
    if (cfg < 0)
        return fail(DAD_E_INVALID, "no tile configuration for %s (M=%d, C/8=%d, L=%d)",
                    op.name.c_str(), op.M, op.cout / 8, op.Lout);
    const SplitPlan sp = g_split_enabled ? plan_split(op, cfg, batch)
                                         : SplitPlan{1, op.cin_pad, 0};   // one slice: every chunk
    p.kslices = sp.kslices;
    p.chunks_per_slice = sp.chunks_per_slice;
    p.slab = ws + m->plan.floats_per_sample * (long)batch;     // scratch behind the activations
    p.counters = m->d_counters;
    p.c1 = op.c1; p.c2 = op.c2;
    {
        const TileCfg& tc = kTiles[cfg];
        const int kc = eff_kc(tc.KC, tc.BM, op.taps, tc.SK, op.x3, op.bdir);
        p.xswz = g_xswz_enabled ? find_xswz(op.Lout, op.stride, op.taps / 2, (kc + 4) / 4, tc.BN) : 0;
    }
    static const bool trace = getenv("DAD_TRACE_TILES") != nullptr;     // tuning aid
    if (trace)
        fprintf(stderr, "[dad] %-34s B=%d M=%d K=%dx%d L=%d tile=%d (%dx%d SK%d) kslices=%d\n", op.name.c_str(),
                batch, op.M, op.taps, op.cin0 + op.cin1, op.Lout, cfg, kTiles[cfg].BM, kTiles[cfg].BN,
                kTiles[cfg].SK, sp.kslices);
#ifdef DAD_STAMPS
    p.stamps = g_stamps ? g_stamps + (size_t)(&op - &m->plan.convs[0]) * 4096 * 8 : nullptr;
#endif
    int rc;
    switch (cfg) {
        case 0: rc = launch_conv_cfg<0>(p, op.taps, op.stride, op.x3, op.bdir, st); break;
        case 1: rc = launch_conv_cfg<1>(p, op.taps, op.stride, op.x3, op.bdir, st); break;
        case 2: rc = launch_conv_cfg<2>(p, op.taps, op.stride, op.x3, op.bdir, st); break;
        case 3: rc = launch_conv_cfg<3>(p, op.taps, op.stride, op.x3, op.bdir, st); break;
        case 4: rc = launch_conv_cfg<4>(p, op.taps, op.stride, op.x3, op.bdir, st); break;
        case 5: rc = launch_conv_cfg<5>(p, op.taps, op.stride, op.x3, op.bdir, st); break;
        case 6: rc = launch_conv_cfg<6>(p, op.taps, op.stride, op.x3, op.bdir, st); break;
        default: rc = launch_conv_cfg<7>(p, op.taps, op.stride, op.x3, op.bdir, st); break;
    }
    return rc;
}

int check_ready(const dad_model* m, int batch, int t, size_t ws_bytes) {
    if (!m) return fail(DAD_E_INVALID, "null model");
    if (!m->finalized) return fail(DAD_E_STATE, "dad_model_finalize has not been called");
    if (batch <= 0) return fail(DAD_E_INVALID, "batch must be positive (got %d)", batch);
    if (t < 0 || t >= m->cfg.n_timesteps)
        return fail(DAD_E_RANGE, "index %d is out of bounds for the schedule of size %d", t,
                    m->cfg.n_timesteps);
    const size_t need = ((size_t)m->plan.floats_per_sample * batch + (size_t)slab_floats_for(m, batch)) * sizeof(float);
    if (ws_bytes < need)
        return fail(DAD_E_WORKSPACE, "workspace has %zu bytes, batch %d needs %zu", ws_bytes, batch,
                    need);
    return DAD_OK;
}

int run_unet(dad_model* m, const float* x, int t, int batch, float* ws, hipStream_t st) {
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
    for (const ConvOp& op : m->plan.convs) {
        const int rc = run_conv(m, op, x, ws, batch, t, st);
        if (rc != DAD_OK) return rc;
        if (m->profile) { m->prof_flops += op.flops_per_sample * batch; ++m->prof_launches; }
    }
    if (m->profile) HIP_TRY(hipEventRecord(e1, st));
    return DAD_OK;
}

int run_final(dad_model* m, float* x, const float* x_ro, int t, int batch, const dad_step_args* a,
              int x_out_disabled, float* eps_only, float* ws, hipStream_t st,
              bool seed_from_device = false) {
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
    const size_t lds = dad::final_lds_floats(c.transition_dim, c.dim) * sizeof(float);
    if (lds > 160 * 1024)
        return fail(DAD_E_INVALID, "final 1x1 conv does not fit LDS (td=%d, dim=%d)", c.transition_dim, c.dim);
    const long N = (long)batch * c.horizon;
    hipLaunchKernelGGL(dad::final_posterior_kernel,
                       dim3((unsigned)((N + dad::FINAL_COLS - 1) / dad::FINAL_COLS)), dim3(256), lds, st, p);
    HIP_TRY(hipGetLastError());
    return DAD_OK;
}

int run_project(const dad_project_args* pa, float alpha, float* x, int batch, int horizon,
                hipStream_t st) {
    if (!pa || !pa->P) return fail(DAD_E_INVALID, "projection arguments missing");
    if (alpha <= 0.0f) return DAD_OK;                     // policies.py:428-429
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
    // rows per block: one while the batch fits one wave of blocks (every CU streams P once),
    // four beyond that (P is then re-used by four rows per pass)
    const int rb = batch <= 512 ? 1 : 4;
    const size_t lds = (size_t)(1 + 16) * rb * p.D * sizeof(float);   // rows + 16 partial sets
    if (lds > 160 * 1024) return fail(DAD_E_INVALID, "projection dimension D=%d too large", p.D);
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
    if (cfg->kernel_size != 5) return fail(DAD_E_INVALID, "kernel_size %d unsupported (5 only)", cfg->kernel_size);
    if (cfg->n_levels < 1 || cfg->n_levels > DAD_MAX_LEVELS)
        return fail(DAD_E_INVALID, "n_levels %d out of range", cfg->n_levels);
    if (cfg->transition_dim < 1 || cfg->dim < 4 || (cfg->dim & 1) || cfg->time_dim < 1)
        return fail(DAD_E_INVALID, "bad transition_dim/dim/time_dim");
    if (!is_pow2(cfg->horizon) || (cfg->horizon >> (cfg->n_levels - 1)) < 4)
        return fail(DAD_E_INVALID, "horizon %d must be a power of two with horizon / 2^(levels-1) >= 4",
                    cfg->horizon);
    if (cfg->n_timesteps < 1) return fail(DAD_E_INVALID, "n_timesteps must be positive");
    for (int i = 0; i < cfg->n_levels; ++i) {
        const int ch = cfg->channels[i];
        if (ch < 32 || ch % 32 != 0 || !is_pow2(ch / 8))
            return fail(DAD_E_INVALID, "level %d has %d channels: need a multiple of 32 with C/8 a power of two",
                        i, ch);
    }
    if (cfg->dim % 32 != 0) return fail(DAD_E_INVALID, "dim %d must be a multiple of 32", cfg->dim);
    std::unique_ptr<dad_model> m(new dad_model());
    m->cfg = *cfg;
    const int rc = build_plan(m.get());
    if (rc != DAD_OK) return rc;
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

int dad_model_set_precision(dad_model* m, int32_t precision) {
    if (!m) return fail(DAD_E_INVALID, "null model");
    if (precision != DAD_PREC_FP32 && precision != DAD_PREC_F16X3)
        return fail(DAD_E_INVALID, "unknown precision %d (DAD_PREC_FP32 = 0, DAD_PREC_F16X3 = 1)", precision);
    if (precision != m->precision) m->finalized = false;       // weights must be re-packed
    m->precision = precision;
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
        m->arena_cap = arena_bytes_needed(m);
        HIP_TRY(hipMalloc(&a, m->arena_cap));
        m->owned.push_back(a);
        m->arena = (char*)a;
        m->arena_used = 0;
    }

    for (ConvOp& op : m->plan.convs) {
        const HostTensor& w = m->raw[op.name + ".weight"];
        const HostTensor& b = m->raw[op.name + ".bias"];
        // wide-group layers (op.kc == 8) use the direct-B kernel in either arithmetic: 16-channel
        // granules, whole 32-channel chunks, 5-tap stride-1 only (else the LDS-staged wide kernel)
        const int cin_all = op.cin0 + op.cin1;
        op.bdir = op.kc == 8 && op.kind == CONV_K5 &&
                  (op.cin0 % 32) == 0 && (cin_all % 32) == 0 && op.cin_pad == cin_all;
        const int pack_g = op.bdir ? 16 : op.kc;
        std::vector<float> packed = op.kind == CONV_UP ? pack_convT(w, op.cin_pad, pack_g)
                                                       : pack_conv(w, op.cin_pad, op.taps, pack_g);
        // split-f16 operands where the kernels exist for every tile this layer may get: 16-channel
        // granules, and for the strided / transposed convs (no general staging path) whole
        // 64-channel chunks
        const int cin = op.cin0 + op.cin1;
        op.x3 = (op.bdir && m->precision == DAD_PREC_F16X3) ||
                (m->precision == DAD_PREC_F16X3 && op.kc == 16 &&
                            (op.kind == CONV_K5 || op.kind == CONV_1X1 ||
                             ((op.cin0 & 63) == 0 && (cin & 63) == 0)));
        op.c1 = 1.0f; op.c2 = 0.0f;
        if (op.x3) {
            const int sh = split_f16_image(packed);
            op.c1 = std::ldexp(1.0f, -sh);
            op.c2 = std::ldexp(1.0f, -sh - 11);
        }
        int rc = upload(m, packed, &op.d_w);
        if (rc != DAD_OK) return rc;
        std::vector<float> bias = b.data;
        if (op.kind == CONV_UP) bias.insert(bias.end(), b.data.begin(), b.data.end());
        if ((rc = upload(m, bias, &op.d_bias)) != DAD_OK) return rc;
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
    std::vector<float> emb((size_t)T * dim);
    {   // SinusoidalPosEmb (temporal_unet.py:27-31) in fp32, as torch computes it
        const int half = dim / 2;
        const float scale = (float)(-(std::log(10000.0) / (half - 1)));
        for (int t = 0; t < T; ++t)
            for (int j = 0; j < half; ++j) {
                const float f = std::exp((float)j * scale);
                const float arg = (float)t * f;
                emb[(size_t)t * dim + j] = std::sin(arg);
                emb[(size_t)t * dim + half + j] = std::cos(arg);
            }
    }
    float *d_emb, *d_h1, *d_temb, *d_w, *d_b;
    if ((rc = upload(m, emb, &d_emb)) != DAD_OK) return rc;
    std::vector<float> zeros((size_t)T * 4 * tdm, 0.0f);
    if ((rc = upload(m, zeros, &d_h1)) != DAD_OK) return rc;
    zeros.resize((size_t)T * tdm);
    if ((rc = upload(m, zeros, &d_temb)) != DAD_OK) return rc;
    zeros.assign((size_t)T * std::max(1, m->plan.temb_width), 0.0f);
    if ((rc = upload(m, zeros, &m->d_temb_table)) != DAD_OK) return rc;
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
    *bytes = ((size_t)m->plan.floats_per_sample * (size_t)batch + (size_t)slab_floats_for(m, batch)) * sizeof(float);
    return DAD_OK;
}

int dad_unet_forward(dad_model* m, const float* x, int32_t t, float* out, int32_t batch,
                     void* workspace, size_t workspace_bytes, dad_stream_t stream) {
    int rc = check_ready(m, batch, t, workspace_bytes);
    if (rc != DAD_OK) return rc;
    if (!x || !out || !workspace) return fail(DAD_E_INVALID, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = run_unet(m, x, t, batch, (float*)workspace, st)) != DAD_OK) return rc;
    return run_final(m, nullptr, x, t, batch, nullptr, 1, out, (float*)workspace, st);
}

int dad_denoise_step(dad_model* m, float* x, int32_t t, int32_t batch, const dad_step_args* args,
                     int32_t x_out_disabled, void* workspace, size_t workspace_bytes,
                     dad_stream_t stream) {
    int rc = check_ready(m, batch, t, workspace_bytes);
    if (rc != DAD_OK) return rc;
    if (!x || !args || !workspace) return fail(DAD_E_INVALID, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = run_unet(m, x, t, batch, (float*)workspace, st)) != DAD_OK) return rc;
    return run_final(m, x, nullptr, t, batch, args, x_out_disabled, nullptr, (float*)workspace, st);
}

int dad_project(const dad_project_args* p, float alpha, float* x, int32_t batch, int32_t horizon,
                dad_stream_t stream) {
    if (!x || batch <= 0 || horizon <= 0) return fail(DAD_E_INVALID, "bad argument");
    return run_project(p, alpha, x, batch, horizon, (hipStream_t)stream);
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
    auto enqueue_all = [&](hipStream_t st) -> int {
        for (int j = 0; j < n_steps; ++j) {
            const int t = n_steps - 1 - j;
            dad_step_args a{};
            a.noise = noise_stack ? noise_stack + (long)j * step_elems : nullptr;
            a.seed = seed; a.row_offset = row_offset; a.draw = (uint64_t)(j + 1);
            a.cond0 = cond0; a.cond_per_row = cond_per_row;
            int r = run_unet(m, x, t, batch, (float*)workspace, st);
            if (r != DAD_OK) return r;
            if ((r = run_final(m, x, nullptr, t, batch, &a, 0, nullptr, (float*)workspace, st,
                               seed_dev)) != DAD_OK)
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
    key.P = proj ? proj->P : nullptr;
    key.n_steps = n_steps; key.batch = batch; key.cond_per_row = cond_per_row;
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

int dad_debug_set_tile(int32_t cfg) {
    // cfg >= 100: same, with grid-level split-K disabled (cfg - 100 is the tile, 99 = heuristic)
    g_split_enabled = cfg < 99;
    g_force_tile = cfg >= 99 ? cfg - 100 : cfg;
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
