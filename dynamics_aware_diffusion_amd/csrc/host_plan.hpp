// host_plan.hpp — everything libdad_hip.so decides on the HOST before a kernel is launched:
// validation of the architecture, the launch plan of TemporalUnet.forward, workspace layout,
// weight packing (fp32 and split-f16 images), tile choice, grid-level split-K, the LDS slot
// shifts and the launch geometry of every conv-GEMM.  Plain C++17, no HIP: dad_lib.hip includes
// it for the product, tests/sanitize/host_check.cpp compiles it host-only under
// -fsanitize=address,undefined.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/dad.h"
#include "conv_shapes.hpp"

namespace dadhost {

inline thread_local char g_err[1024] = "";

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

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
    bool bdir = false;       // wide tile: weight fragments go global -> registers
    // identity residual over a channel concat (decoder block whose 2*C_in equals C_out): the two
    // halves are copied side by side into the `res` buffer before this launch
    int cat0 = -1, cat1 = -1, cat_c0 = 0, cat_c1 = 0;
    // The 1x1 residual conv of a ResidualTemporalBlock reads exactly the rows the block's first
    // 5-tap conv stages (temporal_unet.py:117-121).  It exists in the plan as its own launch
    // (rider_of = index of that conv) and, where the kernel variant exists, ALSO as a sixth
    // "tap" inside that conv's weight image (rname/rdst; own accumulator, plain bias epilogue).
    // Which of the two runs is decided per batch (fused_at): the ride needs the whole K in one
    // block, so batches small enough for grid-level split-K keep the separate launch.
    std::string rname;       // weight key prefix of the riding residual conv, or ""
    int rdst = -1;           // buffer the ride writes
    bool ride = false;       // weight image holds the sixth tap (decided at pack time)
    int rider_of = -1;       // this op is the stand-alone form of convs[rider_of]'s ride
    float c1 = 1.0f, c2 = 0.0f;   // x3: output scales 2^-s and 2^-(s+11)
    // training plan only (HostModel::tplan): where the forward keeps what the backward pass needs of a
    // GroupNorm'd conv — its pre-normalisation output (conv + bias) and the (mean, rstd) of every
    // (sample, group) pair, [8][2] floats per sample
    int pre = -1, stats = -1;
    bool net_padded = false; // the model has zero-padded rows or channels: every launch takes the PADDED kernels
    int gn_real = 0;         // > 0: channels per GroupNorm group that exist (the rest of the group is zero padding)
    int lreal = 0;           // > 0: GEMM output rows per sample that exist (dad_model_set_horizon: the rest are zero
                             //      padding: masked in the GroupNorm statistics, stored as zeros); 0: all of them
    int src_len = 0;         // > 0: rows per sample of the EXTERNAL src0 (the trajectory keeps its real horizon)
    int real_in = 0, real_out = 0;   // zero-padded horizon: real positions per sample of the input / output TENSOR (0: no padding)
    // device tensors (owned by the model)
    float* d_w = nullptr;
    float* d_bias = nullptr;
    float* d_gamma = nullptr;
    float* d_beta = nullptr;
    float* d_rbias = nullptr;
    double flops_per_sample = 0;
    int wtaps() const { return taps + (ride ? 1 : 0); }      // tap slots of the packed image
    bool padded() const { return gn_real > 0 || lreal > 0 || src_len > 0; }   // runs the PADDED kernel instantiations
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
// configuration runs 8 waves per block except the last three.
constexpr int kNumTiles = 10;
constexpr TileCfg kTiles[kNumTiles] = {
    {32, 64, 4, 32},    // 0: few output tiles -> deepest split-K
    {64, 64, 2, 32},    // 1: the workhorse at batch 256
    {128, 64, 1, 16},   // 2: GroupNorm groups of 128 channels / plentiful tiles
    {256, 32, 1, 8},    // 3: GroupNorm groups of 256 channels (C = 2048)
    {64, 64, 1, 16},    // 4: plentiful tiles, 4 waves
    {32, 64, 2, 16},    // 5: 4 waves, small LDS: several independent blocks per CU
    {32, 64, 1, 16},    // 6: 2 waves
    {64, 64, 2, 16},    // 7: as 1 with the shallower K chunk (two blocks per CU fit)
    {32, 128, 2, 32},   // 8: layers of 128 positions (horizon 128, QUICKSTART.md:82): one sample per tile
    {64, 128, 1, 16},   // 9: the same with 64-channel tiles (GroupNorm groups of 64 channels)
};

// 1x1 convs have one (tap, group) unit per 8 channels: a deep K chunk keeps enough MFMAs between
// barriers (128 channels; 64 for the 128-row tile, whose stage would not fit LDS twice).
// Split-f16 kernels consume 16 channels per unit: the chunk must give every split-K wave a unit.
// (128-position tiles: a 32-channel chunk — the deep one would not fit LDS twice next to 128 rows.)
constexpr int eff_kc(int cfg_kc, int bm, int taps, int sk = 1, bool x3 = false, bool bd = false, int bn = 64) {
    return bd                          ? 32          // wide tile, direct-B kernel (either arithmetic)
           : (taps == 1 && cfg_kc >= 16) ? (bn >= 128 ? 32 : bm >= 128 ? 64 : 128)
           : (x3 && cfg_kc < 16 * sk)  ? 16 * sk
                                       : cfg_kc;
}

// Host half of a model: what exists before any device allocation.
struct HostModel {
    dad_cfg cfg{};
    std::map<std::string, HostTensor> raw;
    std::map<std::string, std::vector<int64_t>> expected;     // key -> shape
    Plan plan;
    Plan tplan;                                                // the same launches with every tensor in a buffer of
                                                               // its own (nothing is overwritten before the backward
                                                               // pass has read it) + pre-activation / statistics buffers
    int precision = DAD_PREC_FP32;                             // dad_model_set_precision
    // tuning / test hooks (dad_debug_set_tile) — per model, nothing process-wide
    int force_tile = -1;
    bool split_enabled = true;
    bool fuse_residual = true;                                 // 1x1 residual conv rides in conv0
    bool xswz_enabled = true;
    bool xcd_order = true;
    int split_target = 256;
    bool cc_enabled = true;                                    // small batches take the consumer-combine kernels
    int cc_max_rows = 512;                                     //   up to this many batch * horizon rows
    int ccw_max_rows = 128;                                    //   the same for nets whose plan needs conv_ccw.hpp (wide layers)
    int ccw_min_blocks = 256;                                  //   blocks a wide layer keeps when its K slices are fattened
    bool chain_enabled = false;                                // level-0 encoder chain (conv_chain.hpp) for dim <= 128: opt-in
                                                               //   ("chain" option / DAD_CHAIN=1) — measured 73-75 us against 68 us
                                                               //   for the five launches it replaces
    int chain_min_batch = 64;                                  //   from this batch on (one block per sample: below, the
                                                               //   chip is mostly idle and the batch kernels' split-K wins)
    int real_channels[DAD_MAX_LEVELS] = {0};                   // dad_model_set_group_channels: widths before padding (0: as cfg)
    int real_horizon = 0;                                      // dad_model_set_horizon: horizon before padding (0: as cfg)
    int wgrad_blocks = 256;                                    // blocks a weight-gradient launch aims for (tiles x batch splits)
    bool ccw_prefer16 = true;                                  //   two 16-row tiles instead of an LDS-short 32-row one
                                                               //   (measured crossover: batch 16 at H = 32)
    std::map<std::vector<int>, uint64_t> xswz_cache;           // find_xswz memo
    // ---- backward pass (dad_model_set_training): data-gradient launches + the layout of the gradients
    bool training = false;
    struct BwdConv {
        int n = 0;               // data-gradient launches of this forward conv (one per concat source)
        ConvOp op[2];
        int c_lo[2] = {0, 0};    // first input channel of the forward conv each one covers
        int c_n[2] = {0, 0};     // real channels (op.cout is c_n rounded up to 32)
    };
    std::vector<BwdConv> bconvs; // parallel to tplan.convs
    ConvOp bfinal;               // data gradient of final_conv[1] (1x1, transition_dim -> dim)
    struct GradSlot { std::string key; long offset, numel; };
    std::vector<GradSlot> grad_slots;        // flat gradient buffer: reference state_dict keys, torch layouts
    long grad_numel = 0;
    std::map<std::string, long> grad_at;     // key -> offset
    std::map<std::string, int> grad_index;   // key -> index into grad_slots
    int max_cout = 0;                        // widest conv output (per-sample partial sums)
    int max_bwd_m = 0;                       // widest data-gradient launch (zero bias row)
};

inline int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }
inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// ----------------------------------------------------------------------------- validation
inline int check_cfg(const dad_cfg* cfg) {
    if (!cfg) return fail(DAD_E_INVALID, "null argument");
    if (cfg->kernel_size != 3 && cfg->kernel_size != 5 && cfg->kernel_size != 7)
        return fail(DAD_E_INVALID, "kernel_size %d unsupported (3, 5 or 7)", cfg->kernel_size);
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
    return DAD_OK;
}

// ----------------------------------------------------------------------------- planning
struct Allocator {
    std::vector<Buf>& bufs;
    std::vector<bool> in_use;
    bool retain;                     // training plan: a buffer is never handed out twice
    explicit Allocator(std::vector<Buf>& b, bool keep = false) : bufs(b), retain(keep) {}
    int get(long per_sample) {
        int best = -1;
        for (size_t i = 0; i < bufs.size() && !retain; ++i)
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

inline void expect(HostModel* m, const std::string& key, std::vector<int64_t> shape) {
    m->expected[key] = std::move(shape);
}

// Which kernel family a conv belongs to — a function of the architecture and the precision only
// (so workspace sizes do not depend on whether weights have been loaded yet):
//   bdir  wide-group layers (op.kc == 8) use the direct-B kernel in either arithmetic: 16-channel
//         granules, whole 32-channel chunks, the net's k-tap stride-1 convs only (else the LDS-staged wide kernel)
//   ride  the weight image carries the block's 1x1 residual conv as a sixth tap (fp32, LDS-staged)
//   x3    split-f16 operands where the kernels exist for every tile this layer may get: 16-channel
//         granules, and for the strided / transposed convs (no general staging path) whole
//         64-channel chunks
inline void decide_kernel_families(HostModel* m) {
    // a zero-padded net (rows or channels) runs the PADDED instantiations everywhere: fp32, no ride
    bool padded_net = m->real_horizon > 0 && m->real_horizon != m->cfg.horizon;
    for (int i = 0; i < m->cfg.n_levels; ++i) padded_net = padded_net || (m->real_channels[i] > 0 && m->real_channels[i] != m->cfg.channels[i]);
    for (Plan* plan : {&m->plan, &m->tplan})
    for (ConvOp& op : plan->convs) {
        const int cin_all = op.cin0 + op.cin1;
        op.bdir = op.kc == 8 && op.kind == CONV_K5 &&
                  (op.cin0 % 32) == 0 && (cin_all % 32) == 0 && op.cin_pad == cin_all;
        op.ride = !op.rname.empty() && !op.bdir && m->precision == DAD_PREC_FP32 && !padded_net;
        op.x3 = !padded_net && ((op.bdir && m->precision == DAD_PREC_F16X3) ||
                (m->precision == DAD_PREC_F16X3 && op.kc == 16 &&
                 ((op.kind == CONV_K5 && op.taps == 5) || op.kind == CONV_1X1 ||
                  ((op.kind == CONV_DOWN || op.kind == CONV_UP) && (op.cin0 & 63) == 0 && (cin_all & 63) == 0))));
        op.net_padded = padded_net;
    }
}

// Emits the launch plan of TemporalUnet.forward (temporal_unet.py:199-241) including the
// reference's always-upsample decoder and unused level-0 skip.
inline int build_plan_into(HostModel* m, Plan& P, bool retain) {
    const dad_cfg& c = m->cfg;
    P = Plan();
    Allocator A(P.bufs, retain);
    const int k = c.kernel_size;
    const int tdm = c.time_dim;
    int temb_off = 0;

    int level_now = 0;               // level whose width the convs being emitted produce
    const bool rows_padded = m->real_horizon > 0 && m->real_horizon != c.horizon;
    int Lr_now = rows_padded ? m->real_horizon : c.horizon;      // real positions at the level being emitted
    auto conv = [&](const std::string& name, const std::string& norm, ConvKind kind, int src0,
                    int src1, int cin0, int cin1, int cout, int Lin, int dst, int res,
                    int toff) {
        ConvOp op;
        op.name = name; op.norm = norm; op.kind = kind;
        const int real = m->real_channels[level_now];
        if (!norm.empty() && real > 0 && real != cout) op.gn_real = real / 8;
        if (rows_padded) {
            // real rows per sample of the GEMM's output (the transposed conv's GEMM rows are its INPUT positions)
            const int lr_out = kind == CONV_DOWN ? Lr_now / 2 : Lr_now;
            const int lp_out = kind == CONV_DOWN ? Lin / 2 : Lin;
            if (lr_out != lp_out) op.lreal = lr_out;
            if (src0 == -2) op.src_len = Lr_now;
            op.real_in = Lr_now;
            op.real_out = kind == CONV_DOWN ? Lr_now / 2 : kind == CONV_UP ? 2 * Lr_now : Lr_now;
        }
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
            if (retain) { op.pre = A.get((long)cout * op.Lout); op.stats = A.get(16); }
        }
        P.convs.push_back(op);
    };

    // widths before zero-padding (dad_model_set_group_channels): which blocks have a residual conv is a property of
    // the REAL widths (40 / 80 / 120 channels run as 64 / 128 / 128: the 80 -> 120 block keeps its 1x1 conv)
    auto realw = [&](int level) { return m->real_channels[level] > 0 ? m->real_channels[level] : c.channels[level]; };
    bool padded_net = false;
    for (int i = 0; i < c.n_levels; ++i) padded_net = padded_net || realw(i) != c.channels[i];
    const char* plan_error = nullptr;
    auto res_block = [&](const std::string& base, int in0, int in1, int cin0, int cin1, int cout,
                         int L, int rcin, int rcout) -> int {
        const int cin = padded_net ? (rcin == rcout ? cout : cout + 1) : cin0 + cin1;   // only compared with cout below
        const int toff = temb_off;
        temb_off += cout;
        expect(m, base + ".time_mlp.1.weight", {cout, tdm});
        expect(m, base + ".time_mlp.1.bias", {cout});
        const int a0 = A.get((long)cout * L);
        conv(base + ".blocks.0.block.0", base + ".blocks.0.block.1", CONV_K5, in0, in1, cin0, cin1,
             cout, L, a0, -1, toff);
        int res = -1;
        const bool cat_identity = cin == cout && in1 >= 0;   // nn.Identity over torch.cat([x, skip])
        if (cat_identity && padded_net)
            plan_error = "an identity residual over a channel concat (shrinking dim_mults) with zero-padded GroupNorm groups";
        if (cin != cout) {
            res = A.get((long)cout * L);
            const int c0 = (int)P.convs.size() - 1;
            conv(base + ".residual_conv", "", CONV_1X1, in0, in1, cin0, cin1, cout, L, res, -1, -1);
            if (P.convs[c0].kc == 16) {          // LDS-staged weights: a sixth tap can ride
                P.convs[c0].rname = base + ".residual_conv";
                P.convs[c0].rdst = res;
                P.convs.back().rider_of = c0;
            }
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
        level_now = i;
        const std::string b = "downs." + std::to_string(i);
        const int h1 = res_block(b + ".0", x, -1, cx, 0, co, L, i == 0 ? c.transition_dim : realw(i - 1), realw(i));
        if (x >= 0) A.put(x);
        const int h2 = res_block(b + ".1", h1, -1, co, 0, co, L, realw(i), realw(i));
        A.put(h1);
        skips.push_back(h2);
        skip_ch.push_back(co);
        if (i < nl - 1) {
            const int d = A.get((long)co * (L / 2));
            conv(b + ".2.conv", "", CONV_DOWN, h2, -1, co, 0, co, L, d, -1, -1);
            L /= 2;
            Lr_now /= 2;
            x = d;
            if (i == 0) A.put(h2);   // level-0 skip is pushed but never popped (F8)
        } else {
            x = h2;
        }
        cx = co;
    }
    const int cm = c.channels[nl - 1];
    level_now = nl - 1;
    const int m1 = res_block("mid_block1", x, -1, cm, 0, cm, L, realw(nl - 1), realw(nl - 1));
    const int m2 = res_block("mid_block2", m1, -1, cm, 0, cm, L, realw(nl - 1), realw(nl - 1));
    A.put(m1);
    x = m2;
    cx = cm;
    for (int j = 0; j < nl - 1; ++j) {
        const int lvl = nl - 1 - j;                 // level whose skip is popped
        const int skip = skips[lvl];
        const int cs = skip_ch[lvl];
        const int co = c.channels[lvl - 1];
        level_now = lvl - 1;
        const std::string b = "ups." + std::to_string(j);
        const int u1 = res_block(b + ".0", x, skip, cx, cs, co, L, (j == 0 ? realw(nl - 1) : realw(lvl)) + realw(lvl), realw(lvl - 1));
        A.put(x);
        A.put(skip);
        const int u2 = res_block(b + ".1", u1, -1, co, 0, co, L, realw(lvl - 1), realw(lvl - 1));
        A.put(u1);
        const int up = A.get((long)co * (2 * L));
        conv(b + ".2.conv", "", CONV_UP, u2, -1, co, 0, co, L, up, -1, -1);
        A.put(u2);
        L *= 2;
        Lr_now *= 2;
        x = up;
        cx = co;
    }
    if (rows_padded)
        for (const ConvOp& op : P.convs)
            if (op.res == -2) plan_error = "transition_dim == dim (the first block's residual is the trajectory itself) with a zero-padded horizon";
    if (plan_error != nullptr) return fail(DAD_E_INVALID, "%s", plan_error);
    if (cx != c.dim)
        return fail(DAD_E_INVALID, "final_conv expects %d channels but the decoder ends with %d "
                    "(reference requires dim_mults[0] == 1)", c.dim, cx);
    const int f = A.get((long)c.dim * L);
    level_now = 0;
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
// ------------------------------------------------------------------------- backward plan
// The data gradient of every conv is itself a conv on the forward kernels, with its own weight image:
//   Conv1d k (stride 1)          dX[ci,i] = sum_co sum_k' W[co,ci,K-1-k'] dY[co, i - K/2 + k']     same kind, transposed + flipped
//   Downsample1d (k3, s2, p1)    dX[ci, 2j-1+k] += W[co,ci,k] dY[co,j]   = a ConvTranspose1d(k4,s2,p1) whose 4th tap is zero
//   Upsample1d (convT k4,s2,p1)  dX[ci,i] = sum_co sum_kk Wt[ci,co,kk] dY[co, 2i-1+kk]            = a 5-tap stride-2 conv (pad 2) whose first tap is zero
// (/root/reference/m_diffuser/models/temporal_unet.py:35-76; autograd's conv backward, restated.)
inline int round_up(int v, int to) { return (v + to - 1) / to * to; }
inline ConvOp make_bwd_op(const ConvOp& f, const char* tag, ConvKind kind, int taps, int stride, int cin, int c_n,
                          int Lin, int Lout) {
    ConvOp b;
    b.name = f.name + tag;
    b.kind = kind; b.taps = taps; b.stride = stride;
    b.cin0 = cin; b.cin1 = 0;
    b.kc = 16;
    b.cin_pad = round_up(cin, kind == CONV_1X1 ? 128 : 64);
    b.cout = round_up(c_n, 32);
    b.M = kind == CONV_UP ? 2 * b.cout : b.cout;
    b.Lin = Lin; b.Lout = Lout;
    b.src0 = b.src1 = b.dst = b.res = -1;
    b.temb_off = -1;
    b.flops_per_sample = 2.0 * b.cout * cin * (kind == CONV_UP ? 4 : taps) * Lout;
    return b;
}
inline int build_backward_plan(HostModel* m) {
    const std::vector<ConvOp>& convs = m->tplan.convs;
    m->bconvs.assign(convs.size(), HostModel::BwdConv());
    m->grad_slots.clear(); m->grad_at.clear(); m->grad_index.clear(); m->grad_numel = 0;
    m->max_cout = m->cfg.dim; m->max_bwd_m = m->cfg.dim;
    auto slot = [&](const std::string& key, long numel) {
        m->grad_index[key] = (int)m->grad_slots.size();
        m->grad_slots.push_back({key, m->grad_numel, numel});
        m->grad_at[key] = m->grad_numel;
        m->grad_numel += (numel + 3) / 4 * 4;
    };
    for (size_t i = 0; i < convs.size(); ++i) {
        const ConvOp& f = convs[i];
        HostModel::BwdConv& b = m->bconvs[i];
        const int cin = f.cin0 + f.cin1;
        switch (f.kind) {
            case CONV_K5: case CONV_1X1:
                b.n = f.cin1 > 0 ? 2 : 1;
                for (int s = 0; s < b.n; ++s) {
                    b.c_lo[s] = s == 0 ? 0 : f.cin0;
                    b.c_n[s] = s == 0 ? f.cin0 : f.cin1;
                    b.op[s] = make_bwd_op(f, s == 0 ? ".dgrad0" : ".dgrad1", f.kind, f.taps, 1, f.cout, b.c_n[s], f.Lin, f.Lin);
                }
                slot(f.name + ".weight", (long)f.cout * cin * f.taps);
                break;
            case CONV_DOWN:
                b.n = 1; b.c_lo[0] = 0; b.c_n[0] = cin;
                b.op[0] = make_bwd_op(f, ".dgrad0", CONV_UP, 2, 1, f.cout, cin, f.Lout, f.Lout);
                slot(f.name + ".weight", (long)f.cout * cin * 3);
                break;
            case CONV_UP:
                b.n = 1; b.c_lo[0] = 0; b.c_n[0] = cin;
                b.op[0] = make_bwd_op(f, ".dgrad0", CONV_DOWN, 5, 2, f.cout, cin, 2 * f.Lin, f.Lin);
                slot(f.name + ".weight", (long)cin * f.cout * 4);
                break;
        }
        // zero-padded horizon: the data gradient is zero-padded like every activation (its launches store zeros
        // behind the real rows: real GEMM rows = the forward conv's INPUT positions; the down-sampling conv's
        // data gradient is a transposed conv whose GEMM rows are the forward OUTPUT positions)
        for (int s = 0; s < b.n; ++s) {
            const int real_rows = f.kind == CONV_DOWN ? f.real_out : f.real_in;
            const int rows = f.kind == CONV_DOWN ? f.Lout : f.Lin;
            b.op[s].lreal = (real_rows > 0 && real_rows != rows) ? real_rows : 0;
            b.op[s].net_padded = f.net_padded;
        }
        slot(f.name + ".bias", f.cout);
        if (!f.norm.empty()) { slot(f.norm + ".weight", f.cout); slot(f.norm + ".bias", f.cout); }
        m->max_cout = std::max(m->max_cout, f.cout);
        for (int s = 0; s < b.n; ++s) m->max_bwd_m = std::max(m->max_bwd_m, b.op[s].M);
    }
    {   // final_conv[1]: 1x1, dim -> transition_dim; its data gradient is a 1x1 conv transition_dim -> dim
        ConvOp f;
        f.name = "final_conv.1";
        m->bfinal = make_bwd_op(f, ".dgrad0", CONV_1X1, 1, 1, m->cfg.transition_dim, m->cfg.dim, m->cfg.horizon, m->cfg.horizon);
        if (m->real_horizon > 0 && m->real_horizon != m->cfg.horizon) { m->bfinal.lreal = m->real_horizon; m->bfinal.net_padded = true; }
        slot("final_conv.1.weight", (long)m->cfg.transition_dim * m->cfg.dim);
        slot("final_conv.1.bias", m->cfg.transition_dim);
    }
    return DAD_OK;
}
// Why a model cannot be trained on this engine, or nullptr.
inline const char* training_refusal(const HostModel& m) {
    if (m.precision != DAD_PREC_FP32) return "the backward pass exists for the fp32 arithmetic only";
    return nullptr;
}

inline bool tile_valid(const ConvOp& op, int cfg);
// An architecture some layer of which has no conv-GEMM tile is refused when the model is created, not at
// its first launch (validity of a tile does not depend on the batch).
inline int check_tiles(const HostModel& m) {
    for (const ConvOp& op : m.plan.convs) {
        bool any = false;
        for (int cfg = 0; cfg < kNumTiles && !any; ++cfg) any = tile_valid(op, cfg);
        if (!any)
            return fail(DAD_E_INVALID, "no tile configuration for %s (M=%d, C/8=%d, L=%d, %d taps)",
                        op.name.c_str(), op.M, op.cout / 8, op.Lout, op.taps);
    }
    return DAD_OK;
}
inline int build_plan(HostModel* m) {
    m->expected.clear();
    int rc = build_plan_into(m, m->plan, false);
    if (rc == DAD_OK) rc = build_plan_into(m, m->tplan, true);
    if (rc == DAD_OK) decide_kernel_families(m);
    if (rc == DAD_OK) rc = check_tiles(*m);
    if (rc == DAD_OK) rc = build_backward_plan(m);
    return rc;
}

// ------------------------------------------------------------------------------ packing
// Conv1d weight (co, ci, k)  ->  [ci_pad/KC][wtaps][M = co][KC]; tap index `tap_at + t`.
inline void pack_conv_into(std::vector<float>& out, const HostTensor& w, int wtaps, int tap_at, int kc) {
    const int co = (int)w.shape[0], ci = (int)w.shape[1], k = (int)w.shape[2];
    for (int o = 0; o < co; ++o)
        for (int i = 0; i < ci; ++i)
            for (int t = 0; t < k; ++t) {
                const size_t row = ((size_t)(i / kc) * wtaps + tap_at + t) * co + o;
                out[row * kc + (i % kc)] = w.data[((size_t)o * ci + i) * k + t];
            }
}
inline std::vector<float> pack_conv(const HostTensor& w, int cin_pad, int taps, int kc) {
    std::vector<float> out((size_t)cin_pad * taps * (size_t)w.shape[0], 0.0f);
    pack_conv_into(out, w, taps, 0, kc);
    return out;
}

// ConvTranspose1d weight (ci, co, 4), stride 2, pad 1:
//   y[co, 2j]   = sum_ci W[ci,co,3] x[ci,j-1] + W[ci,co,1] x[ci,j]
//   y[co, 2j+1] = sum_ci W[ci,co,2] x[ci,j]   + W[ci,co,0] x[ci,j+1]
// packed as a 2-tap conv with M = 2*co columns: columns [0,co) are the even phase (taps at
// positions j-1, j), columns [co,2co) the odd phase (taps at j, j+1 — the kernel shifts the row
// base by one for tiles of that half).
inline std::vector<float> pack_convT(const HostTensor& w, int cin_pad, int kc) {
    const int ci = (int)w.shape[0], co = (int)w.shape[1];
    const int M = 2 * co;
    std::vector<float> out((size_t)cin_pad * 2 * M, 0.0f);
    auto at = [&](int i, int o, int kk) { return w.data[((size_t)i * co + o) * 4 + kk]; };
    for (int i = 0; i < ci; ++i)
        for (int o = 0; o < co; ++o) {
            auto slot = [&](int tap, int mm) -> float& {
                return out[(((size_t)(i / kc) * 2 + tap) * M + mm) * kc + (i % kc)];
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
inline uint16_t f16_bits(float v) {
    const _Float16 h = (_Float16)v;       // round to nearest even
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
inline int split_f16_image(std::vector<float>& packed) {
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

// What dad_model_finalize uploads for one conv launch.
struct PackedOp {
    std::vector<float> w, bias, rbias;
};
// Decides the layer's kernel family (direct-B, split-f16) and produces its packed image.
inline int pack_op(HostModel* m, ConvOp& op, PackedOp& out) {
    auto need = [&](const std::string& key) -> const HostTensor* {
        auto it = m->raw.find(key);
        return it == m->raw.end() ? nullptr : &it->second;
    };
    const HostTensor* w = need(op.name + ".weight");
    const HostTensor* b = need(op.name + ".bias");
    if (!w || !b) return fail(DAD_E_KEY, "missing key '%s.weight/.bias'", op.name.c_str());
    const int pack_g = op.bdir ? 16 : op.kc;            // flags: decide_kernel_families
    if (op.kind == CONV_UP) {
        out.w = pack_convT(*w, op.cin_pad, pack_g);
    } else {
        const int wt = op.wtaps();
        out.w.assign((size_t)op.cin_pad * wt * (size_t)op.M, 0.0f);
        pack_conv_into(out.w, *w, wt, 0, pack_g);
        if (op.ride) {
            const HostTensor* rw = need(op.rname + ".weight");
            const HostTensor* rb = need(op.rname + ".bias");
            if (!rw || !rb) return fail(DAD_E_KEY, "missing key '%s.weight/.bias'", op.rname.c_str());
            pack_conv_into(out.w, *rw, wt, op.taps, pack_g);
            out.rbias = rb->data;
        }
    }
    op.c1 = 1.0f; op.c2 = 0.0f;
    if (op.x3) {
        const int sh = split_f16_image(out.w);
        op.c1 = std::ldexp(1.0f, -sh);
        op.c2 = std::ldexp(1.0f, -sh - 11);
    }
    out.bias = b->data;
    if (op.kind == CONV_UP) out.bias.insert(out.bias.end(), b->data.begin(), b->data.end());
    return DAD_OK;
}

// Weight image of data-gradient launch `s` of forward conv `f` (see build_backward_plan).
inline int pack_bwd_op(HostModel* m, const ConvOp& f, const HostModel::BwdConv& b, int s, std::vector<float>& out) {
    auto it = m->raw.find(f.name + ".weight");
    if (it == m->raw.end()) return fail(DAD_E_KEY, "missing key '%s.weight'", f.name.c_str());
    const HostTensor& w = it->second;
    const ConvOp& op = b.op[s];
    const int cin = f.cin0 + f.cin1;
    HostTensor t;
    if (f.kind == CONV_K5 || f.kind == CONV_1X1) {
        const int K = f.taps;
        t.shape = {op.cout, f.cout, K};
        t.data.assign((size_t)op.cout * f.cout * K, 0.0f);
        for (int mm = 0; mm < b.c_n[s]; ++mm)
            for (int co = 0; co < f.cout; ++co)
                for (int k = 0; k < K; ++k)
                    t.data[((size_t)mm * f.cout + co) * K + k] = w.data[((size_t)co * cin + b.c_lo[s] + mm) * K + (K - 1 - k)];
        out.assign((size_t)op.cin_pad * K * op.M, 0.0f);
        pack_conv_into(out, t, K, 0, 16);
    } else if (f.kind == CONV_DOWN) {            // -> transposed conv (in = co, out = ci, 4 taps; tap 3 zero)
        t.shape = {f.cout, cin, 4};
        t.data.assign((size_t)f.cout * cin * 4, 0.0f);
        for (int co = 0; co < f.cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int k = 0; k < 3; ++k)
                    t.data[((size_t)co * cin + ci) * 4 + k] = w.data[((size_t)co * cin + ci) * 3 + k];
        out = pack_convT(t, op.cin_pad, 16);
    } else {                                     // CONV_UP -> 5-tap stride-2 conv (out = ci, in = co; tap 0 zero)
        t.shape = {cin, f.cout, 5};
        t.data.assign((size_t)cin * f.cout * 5, 0.0f);
        for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < f.cout; ++co)
                for (int kk = 0; kk < 4; ++kk)
                    t.data[((size_t)ci * f.cout + co) * 5 + kk + 1] = w.data[((size_t)ci * f.cout + co) * 4 + kk];
        out.assign((size_t)op.cin_pad * 5 * op.M, 0.0f);
        pack_conv_into(out, t, 5, 0, 16);
    }
    return DAD_OK;
}
inline int pack_bwd_final(HostModel* m, std::vector<float>& out) {
    auto it = m->raw.find("final_conv.1.weight");
    if (it == m->raw.end()) return fail(DAD_E_KEY, "missing key 'final_conv.1.weight'");
    const HostTensor& w = it->second;            // (td, dim, 1)
    const int td = m->cfg.transition_dim, dim = m->cfg.dim;
    HostTensor t;
    t.shape = {dim, td, 1};
    t.data.assign((size_t)dim * td, 0.0f);
    for (int j = 0; j < td; ++j)
        for (int c = 0; c < dim; ++c) t.data[(size_t)c * td + j] = w.data[(size_t)j * dim + c];
    out.assign((size_t)m->bfinal.cin_pad * m->bfinal.M, 0.0f);
    pack_conv_into(out, t, 1, 0, 16);
    return DAD_OK;
}

// ------------------------------------------------------------------------- tile choice
// Hard constraints: the tile holds whole GroupNorm groups (BM % (C/8) == 0) and whole samples
// (BN % L == 0), BM divides the columns (each phase half for the transposed conv), the K chunk
// matches the packed weights.  Preference: enough blocks to cover the 256 CUs; when tiles are
// scarce, trade tile size for split-K depth.
constexpr bool kPaddedTiles[kNumTiles] = {true, true, true, true, true, false, false, false, true, true};
inline bool tile_valid(const ConvOp& op, int cfg) {
    const TileCfg& t = kTiles[cfg];
    if (op.net_padded && !kPaddedTiles[cfg]) return false;      // (PADDED kernels exist for the heuristic's tiles)
    const int Mrows = op.kind == CONV_UP ? op.M / 2 : op.M;
    const int cpg = op.norm.empty() ? 1 : op.cout / 8;
    if ((t.KC == 8) != (op.kc == 8)) return false;
    if (Mrows % t.BM != 0) return false;
    if (!op.norm.empty() && (t.BM % cpg != 0)) return false;
    if (t.BN % op.Lout != 0) return false;
    const int nthreads = 64 * (t.BM / 32) * (t.BN / 32) * t.SK;
    const int f4pl = t.BM * t.BN / 4 / nthreads;
    if (!op.norm.empty() && op.Lout * cpg / 4 < f4pl) return false;   // >= 1 lane per (group, sample)
    // the stage must fit LDS (the 128-position tiles with the 128-channel chunk of a 1x1 conv do not)
    const int kc = eff_kc(t.KC, t.BM, op.taps, t.SK, op.x3, op.bdir, t.BN);
    if (dad::conv_lds_floats(t.BM, t.BN, kc, op.taps, op.Lin, op.Lout, t.SK, op.bdir, op.taps) * sizeof(float) > dad::kLdsBytes)
        return false;
    return true;
}
inline int choose_tile(const HostModel& m, const ConvOp& op, int batch) {
    auto valid = [&](int cfg) { return tile_valid(op, cfg); };
    auto blocks = [&](int cfg) {
        const TileCfg& t = kTiles[cfg];
        const int spt = t.BN / op.Lout;
        return (long)((batch + spt - 1) / spt) * (op.M / t.BM);
    };
    if (op.kc == 8) return valid(3) ? 3 : -1;
    if (m.force_tile >= 0 && m.force_tile < kNumTiles && valid(m.force_tile)) return m.force_tile;
    if (valid(2) && blocks(2) >= 512) return 2;          // plentiful work: big tile
    if (valid(1) && blocks(1) >= 224) return 1;
    if (valid(0)) return 0;
    if (valid(1)) return 1;
    if (valid(2)) return 2;
    if (valid(4)) return 4;
    if (valid(8)) return 8;              // 128 positions per sample
    if (valid(9)) return 9;
    return -1;
}

// Per-sample slot shifts of the X stage (conv_gemm.hpp, "Activation rows in LDS and bank
// conflicts").  Depth-first over the samples of a block tile: d(s) in [0, 16) such that in every
// 32-row wave tile both 16-lane groups of ds_read_b128 see 16 distinct slots, and no sample is
// pushed onto its neighbour's real rows (d(s) - d(s+1) <= pad * slots-per-row).  Returns 0 (plain
// layout — correct, just slower) when L >= 32, when there is no halo, or when nothing is found.
inline uint64_t find_xswz(HostModel& m, int L, int stride, int pad, int kp4, int BN) {
    if (L >= 32 || pad == 0 || BN / L > 16) return 0;
    const std::vector<int> key{L, stride, pad, kp4, BN};
    auto it = m.xswz_cache.find(key);
    if (it != m.xswz_cache.end()) return it->second;
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
    m.xswz_cache[key] = packed;
    return packed;
}

// Grid-level split-K: when a layer has too few output tiles to cover the chip (small batches;
// the deepest levels of the wide nets), several blocks share a tile and split its K chunks.
struct SplitPlan { int kslices, chunks_per_slice; long slab_floats; };
inline SplitPlan plan_split(const HostModel& m, const ConvOp& op, int cfg, int batch) {
    const TileCfg& t = kTiles[cfg];
    const int spt = t.BN / op.Lout;
    const long tiles = (long)((batch + spt - 1) / spt) * (op.M / t.BM);
    const int kc = eff_kc(t.KC, t.BM, op.taps, t.SK, op.x3, op.bdir, t.BN);
    const int nchunks = (op.cin0 + op.cin1 + kc - 1) / kc;      // chunks holding real channels
    SplitPlan sp{1, nchunks, 0};
    if (!m.split_enabled) return sp;
    if (tiles >= 160 || nchunks < 2 || tiles > kMaxSplitTiles) return sp;
    // blocks = tiles * slices should not spill a few blocks into a second wave of the 256 CUs (24 tiles x
    // 11 slices = 264 blocks took 1.4x the time of 24 x 10): round the slice count DOWN while that still
    // splits, up only for tile counts above half the chip
    int want = tiles * 2 <= m.split_target ? (int)(m.split_target / tiles) : (int)((m.split_target + tiles - 1) / tiles);
    if (want > nchunks) want = nchunks;
    if (want < 2) return sp;
    sp.chunks_per_slice = (nchunks + want - 1) / want;
    sp.kslices = (nchunks + sp.chunks_per_slice - 1) / sp.chunks_per_slice;
    sp.slab_floats = tiles * sp.kslices * (long)t.BN * t.BM;
    return sp;
}

// floats of split-K scratch a batch needs (max over layers)
inline long slab_floats_for(const HostModel& m, int batch) {
    long best = 0;
    for (const ConvOp& op : m.plan.convs) {
        const int cfg = choose_tile(m, op, batch);
        if (cfg < 0) continue;
        best = std::max(best, plan_split(m, op, cfg, batch).slab_floats);
    }
    return best;
}

inline size_t workspace_bytes(const HostModel& m, int batch);

// ---------------------------------------------------------------------- launch geometry
// Everything a conv-GEMM launch needs besides pointers, decided on the host and checked here
// (operand shapes against what the kernel and its grid assume) before anything reaches the GPU.
struct LaunchGeom {
    int cfg = -1;            // index into kTiles
    int kc = 0;              // K chunk of the selected kernel instantiation
    bool ragged = false;     // general staging path (first layer, narrow nets)
    int threads = 0;
    size_t lds_bytes = 0;
    int ntiles_n = 0, mtiles = 0;
    unsigned gx = 1, gy = 1, gz = 1;
    int xcd_gn = 0, xcd_mts = 0, xcd_ntn = 0;
    bool fused = false;      // the residual conv rides in this launch
    bool padded = false;     // PADDED instantiation (zero-padded rows / channels)
    SplitPlan split{1, 0, 0};
    uint64_t xswz = 0;
};

// Does the residual conv ride in `op`'s launch at this batch?  (Needs the whole K in one block
// and the sixth tap's weight rows in LDS.)
inline bool fused_at(const HostModel& m, const ConvOp& op, int batch) {
    if (!op.ride || !m.fuse_residual) return false;
    const int cfg = choose_tile(m, op, batch);
    if (cfg < 0) return false;
    const TileCfg& t = kTiles[cfg];
    if (t.KC < 16 || plan_split(m, op, cfg, batch).kslices != 1) return false;
    const int kc = eff_kc(t.KC, t.BM, op.taps, t.SK, op.x3, op.bdir, t.BN);
    return dad::conv_lds_floats(t.BM, t.BN, kc, op.taps, op.Lin, op.Lout, t.SK, false, op.taps + 1) *
               sizeof(float) <= dad::kLdsBytes;
}

// Which conv-GEMM instantiations exist (the registry of dad_lib.hip, reg_tile, restated on the host so that the
// planner — and the sanitizer harness, which has no device code — refuses a launch no kernel was compiled for;
// dad_debug_kernel_table_consistent() compares the two).
inline bool kernel_registered(int cfg, int taps, int stride, bool x3, bool bdir, bool ragged, bool res, bool padded = false) {
    if (cfg < 0 || cfg >= kNumTiles) return false;
    if (padded && (!kPaddedTiles[cfg] || x3 || res)) return false;
    const bool kc16 = kTiles[cfg].KC >= 16;
    const bool k357 = taps == 3 || taps == 5 || taps == 7;
    if (ragged && !(stride == 1 && (k357 || taps == 1) && !bdir)) return false;
    if (bdir) return !kc16 && !res && !ragged && stride == 1 && k357;
    if (res) return kc16 && !x3 && stride == 1 && k357;
    if (x3) return kc16 && ((taps == 5 && stride == 1) || (taps == 3 && stride == 2) || (taps == 2 && stride == 1) ||
                            (taps == 1 && stride == 1));
    if (stride == 2) return taps == 3 || (taps == 5 && kc16);
    return stride == 1 && (taps == 1 || taps == 2 || k357);
}

inline int plan_launch(HostModel& m, const ConvOp& op, int batch, LaunchGeom& g) {
    if ((long)batch * op.Lout * op.M >= (1L << 31) ||
        (long)batch * op.Lin * (op.cin0 + op.cin1) >= (1L << 31))
        return fail(DAD_E_INVALID, "batch %d too large: a layer's activation tensor exceeds 2^31 elements", batch);
    g.cfg = choose_tile(m, op, batch);
    if (g.cfg < 0)
        return fail(DAD_E_INVALID, "no tile configuration for %s (M=%d, C/8=%d, L=%d)",
                    op.name.c_str(), op.M, op.cout / 8, op.Lout);
    const TileCfg& t = kTiles[g.cfg];
    g.kc = eff_kc(t.KC, t.BM, op.taps, t.SK, op.x3, op.bdir, t.BN);
    const int cin = op.cin0 + op.cin1;
    g.ragged = (op.cin0 & 3) != 0 || (op.cin1 & 3) != 0 || op.cin0 % g.kc != 0 || cin % g.kc != 0;
    if (g.ragged && !(op.stride == 1 && (op.taps & 1) == 1))
        return fail(DAD_E_INVALID, "channel count %d+%d needs the general staging path, which exists "
                    "for stride-1 k-tap and 1x1 convs only", op.cin0, op.cin1);
    if (op.bdir && g.ragged)
        return fail(DAD_E_INVALID, "the direct-B kernel needs whole 32-channel chunks (%d+%d)", op.cin0, op.cin1);
    if (op.bdir && !(t.KC == 8 && op.kind == CONV_K5 && op.stride == 1))
        return fail(DAD_E_INVALID, "no direct-B kernel for tile %d taps=%d stride=%d", g.cfg, op.taps, op.stride);
    if (op.x3 && !op.bdir && t.KC < 16)
        return fail(DAD_E_INVALID, "no split-f16 kernel for tile %d taps=%d stride=%d", g.cfg, op.taps, op.stride);
    g.fused = fused_at(m, op, batch);
    if (g.fused && (op.x3 || op.bdir || op.kind != CONV_K5 || op.stride != 1))
        return fail(DAD_E_INVALID, "no fused-residual kernel for %s on tile %d", op.name.c_str(), g.cfg);
    if (op.cin_pad % g.kc != 0 && !g.ragged)
        return fail(DAD_E_INVALID, "%s: padded channel count %d is not a multiple of the K chunk %d",
                    op.name.c_str(), op.cin_pad, g.kc);
    g.padded = op.net_padded;
    if (!kernel_registered(g.cfg, op.taps, op.stride, op.x3, op.bdir, g.ragged, g.fused, g.padded))
        return fail(DAD_E_INVALID, "no kernel for %s (tile %d taps=%d stride=%d x3=%d bdir=%d ragged=%d res=%d padded=%d)",
                    op.name.c_str(), g.cfg, op.taps, op.stride, (int)op.x3, (int)op.bdir, (int)g.ragged, (int)g.fused, (int)g.padded);
    g.threads = 64 * (t.BM / 32) * (t.BN / 32) * t.SK;
    g.lds_bytes = dad::conv_lds_floats(t.BM, t.BN, g.kc, op.taps, op.Lin, op.Lout, t.SK, op.bdir,
                                       op.taps + (g.fused ? 1 : 0)) * sizeof(float);
    if (g.lds_bytes > dad::kLdsBytes)
        return fail(DAD_E_INVALID, "%s: tile %d needs %zu bytes of LDS", op.name.c_str(), g.cfg, g.lds_bytes);
    const int spt = t.BN / op.Lout;
    g.ntiles_n = (batch + spt - 1) / spt;
    g.mtiles = op.M / t.BM;
    if (g.ntiles_n > 65535) return fail(DAD_E_INVALID, "batch too large for one launch (%d N tiles)", g.ntiles_n);
    g.split = plan_split(m, op, g.cfg, batch);
    if (g.split.kslices > 1 && (long)g.mtiles * g.ntiles_n > kMaxSplitTiles)
        return fail(DAD_E_INVALID, "%s: %ld tiles exceed the split-K ticket table", op.name.c_str(),
                    (long)g.mtiles * g.ntiles_n);
    // XCD-aware tile order when every XCD gets the same whole rectangle of tiles: choose the
    // gm x gn arrangement of the 8 XCDs that minimises  gn * (weight bytes) + gm * (activation bytes)
    const int MT = g.mtiles, NTn = g.ntiles_n;
    g.gx = (unsigned)g.split.kslices; g.gy = (unsigned)MT; g.gz = (unsigned)NTn;
    g.xcd_gn = 0;
    if (m.xcd_order && g.split.kslices == 1 && (MT & (MT - 1)) == 0 && (long)MT * NTn <= 65535 &&
        ((long)MT * NTn) % 8 == 0) {
        const double wbytes = (double)op.M * op.taps * cin;
        const double xbytes = (double)batch * op.Lin * cin;
        double best = -1;
        for (int gm = 1; gm <= 8; gm *= 2) {
            const int gn = 8 / gm;
            if (MT % gm != 0 || NTn % gn != 0) continue;
            const double cost = gn * wbytes + gm * xbytes;
            if (best < 0 || cost < best) {
                best = cost;
                g.xcd_gn = gn; g.xcd_mts = ilog2(MT / gm); g.xcd_ntn = NTn / gn;
            }
        }
        if (g.xcd_gn > 0) { g.gy = (unsigned)(MT * NTn); g.gz = 1; }
    }
    g.xswz = m.xswz_enabled ? find_xswz(m, op.Lout, op.stride, op.taps / 2, (g.kc + 4) / 4, t.BN) : 0;
    return DAD_OK;
}

// ------------------------------------------------------------------ small-batch (CC) plan
// conv_cc.hpp: convs only produce partial sums, consumers finish them.  Decided per batch on the
// host: which launches exist, their K slices, where their partial slabs live, and for every input
// whether it is read finished (external trajectory / already materialised) or in pieces.
constexpr int kCcMaxSlabs = 16;          // slabs a consumer adds (8 per round trip)
constexpr int kCcMaxSlice = 64;          // channels per K slice of conv_cc: weight tile + input slice stay well
                                         // inside LDS and 6 float4 of weights per thread
constexpr int kCcwMaxSlabs = 8;          // conv_ccw (wide layers): slabs per input, one round trip
using dad::kCcwMaxPairs;
constexpr int kCcwMaxPair = 8192;        // elements of one pair (up to 2048 stay in registers between the passes)
struct CcInput {
    int kind = 0;            // 0 none, 1 external trajectory, 2 finished tensor in a plan buffer, 3 in pieces
    int buf = -1;            // kind 2 / 3: the tensor's activation buffer (kind 3: where it is materialised)
    int producer = -1;       // kind 3: conv whose partial slabs these are
};
struct CcOp {
    bool launched = false;   // false: the op does not exist in this form (riding 1x1 conv)
    bool wide = false;       // conv_ccw.hpp: weights streamed through registers, K slices of up to 1024 channels
    int slice_ch = 0, kslices = 0, ntiles = 0;
    int tile_rows = 32;      // GEMM rows per tile: 16 for layers of at most 16 positions (16x16x4 MFMAs)
    long oslab = 0, orslab = -1;       // float offsets into the CC slab region
    int out_rows = 0, out_cols = 0;
    size_t lds_bytes = 0;
    CcInput in0, in1;
    // how this conv's OUTPUT is finished by whoever consumes it
    int res_kind = 0;        // 0 none, 1 external trajectory, 2 finished tensor (buffer res_buf),
                             // 3 ride of conv res_ride, 4 the stand-alone 1x1 conv res_ride, still in pieces
    int res_buf = -1, res_ride = -1;
};
struct CcPlan {
    bool ok = false;
    std::vector<CcOp> ops;
    long slab_floats = 0;
    int final_producer = -1;
    std::string why;         // when !ok: which rule refused the plan (diagnostic)
};
inline CcPlan& refuse(CcPlan& P, const char* why) { P.why = why; return P; }

inline CcPlan cc_plan(const HostModel& m, int batch) {
    CcPlan P;
    const dad_cfg& c = m.cfg;
    const std::vector<ConvOp>& convs = m.plan.convs;
    if (m.precision != DAD_PREC_FP32 || !m.cc_enabled || c.horizon > 128 || c.kernel_size != 5 ||
        (long)batch * c.horizon > m.cc_max_rows)
        return refuse(P, "disabled, split-f16 arithmetic, horizon > 128, kernel_size != 5 or more than cc_max_rows rows");
    // horizons beyond 32 (windowed tiles): measured on the PointMaze net at horizon 64 — 258 / 260 / 272 us per denoise
    // step at batch 1 / 2 / 4 against 282 / 293 / 314 on the batch kernels; at batch 8 the batch kernels win
    if (c.horizon > 32 && (long)batch * c.horizon > 256)
        return refuse(P, "horizon > 32 and more than 256 rows");
    for (const ConvOp& op : convs)
        if (op.gn_real > 0) return refuse(P, "zero-padded GroupNorm groups (dad_model_set_group_channels): batch kernels only");
    if (m.real_horizon > 0 && m.real_horizon != c.horizon) return refuse(P, "zero-padded horizon (dad_model_set_horizon): batch kernels only");
    for (const ConvOp& op : convs)      // weight images in 16-channel granules only
        if ((op.kc != 16 && !op.bdir) || op.cat0 >= 0 || op.x3 || (!op.rname.empty() && !op.ride)) return refuse(P, "a weight image not in 16-channel granules, or an identity residual over a concat");
    P.ops.resize(convs.size());
    std::vector<int> owner(m.plan.bufs.size(), -1);     // buffer -> conv whose output it holds
    std::vector<char> materialised(convs.size(), 0);
    long off = 0;
    for (size_t i = 0; i < convs.size(); ++i) {
        const ConvOp& op = convs[i];
        CcOp& o = P.ops[i];
        if (op.rider_of >= 0) { owner[op.dst] = -2 - op.rider_of; continue; }   // lives in its carrier's launch
        o.launched = true;
        // inputs
        auto input = [&](int buf, CcInput& in) -> bool {
            if (buf == -1) { in.kind = 0; return true; }
            if (buf == -2) { in.kind = 1; return true; }
            const int q = owner[buf];
            if (q < 0) return false;                     // unknown producer (should not happen)
            in.buf = buf;
            if (materialised[q]) { in.kind = 2; return true; }
            in.kind = 3; in.producer = q; materialised[q] = 1;
            return true;
        };
        if (!input(op.src0, o.in0) || !input(op.src1, o.in1)) return refuse(P, "input with no known producer");
        // layers of more than 32 positions (horizon 64 / 128): windowed tiles — 32 rows of one sample per tile,
        // stride-1 5-tap convs of conv_cc only (the pair statistics still span the sample: <= 1024 elements)
        const bool windowed = op.Lout > 32;
        if (op.M % 32 != 0 || (!windowed && 32 % op.Lout != 0) || (windowed && (op.Lout % 32 != 0 || op.kind != CONV_K5)))
            return refuse(P, "output columns not a multiple of 32, a length that does not divide 32, or a long layer that is not a stride-1 conv");
        // K slices: whole GroupNorm groups of the tensor being finished
        const int cin = op.cin0 + op.cin1;
        int need = 32, max_slabs_in = 0;
        long max_pair = 0;
        for (const CcInput* in : {&o.in0, &o.in1})
            if (in->kind == 3) {
                const ConvOp& q = convs[in->producer];
                const CcOp& qo = P.ops[in->producer];
                max_slabs_in = std::max(max_slabs_in, qo.kslices);
                if (qo.res_kind >= 3) max_slabs_in = std::max(max_slabs_in, P.ops[qo.res_ride].kslices);
                if (!q.norm.empty()) {
                    need = std::max(need, q.cout / 8);
                    max_pair = std::max(max_pair, (long)(q.cout / 8) * op.Lin);
                    if (need % (q.cout / 8) != 0) return refuse(P, "inputs whose GroupNorm widths do not nest");          // slices must hold whole groups
                }
            }
        int slice = need;
        while ((cin + slice - 1) / slice > 8 && slice < kCcMaxSlice) slice *= 2;     // 8 slabs: one round trip
        while ((cin + slice - 1) / slice > kCcMaxSlabs) slice *= 2;
        // conv_cc keeps the whole weight slice in LDS and normalises a pair in one wave's registers;
        // anything wider goes to conv_ccw (weights streamed global -> registers)
        o.wide = slice > kCcMaxSlice || max_pair > 1024 || op.bdir || op.kind == CONV_1X1;   // (conv_cc has no 1x1 form)
        if (o.wide && (windowed || op.Lin > 32)) return refuse(P, "a wide layer (conv_ccw) of more than 32 positions");
        if (windowed && max_slabs_in > 8) return refuse(P, "a windowed layer fed more than 8 slabs");
        if (!o.wide) {
            if (slice % 32 != 0) return refuse(P, "K slice not a multiple of 32 channels");
            // 16-row tiles (16x16x4 MFMAs, half the padded rows) as long as the layer still fits one wave
            // of blocks; beyond that the extra N tiles only re-stream the weights
            o.tile_rows = 32;
            if (op.Lout <= 16) {
                const int ks = (cin + slice - 1) / slice;
                const long blocks16 = (long)((batch + 16 / op.Lout - 1) / (16 / op.Lout)) * (op.M / 32) * ks;
                if (blocks16 <= 256) o.tile_rows = 16;
            }
        } else {
            if (op.src0 == -2 || (op.cin0 & 3) || (op.cin1 & 3) || max_slabs_in > kCcwMaxSlabs ||
                max_pair > kCcwMaxPair)
                return refuse(P, "wide layer: ragged channels, more than 8 slabs to add, or a GroupNorm pair above 8192 elements");
            // tile rows: 16 when that needs no more N tiles than 32 would (batch 1 / 2 on short levels),
            // or when a 32-row tile of the narrowest admissible slice does not fit LDS
            auto try_rows = [&](int rows) -> int {
                auto fits = [&](int sl) {
                    return dad::ccw_lds_floats(sl, op.taps, op.Lin, op.Lout, rows) * sizeof(float) <= dad::kLdsBytes;
                };
                const int spt_r = rows / op.Lout;
                const long nt = (batch + spt_r - 1) / spt_r;
                int sl = need;
                while (sl % 32 != 0) sl += need;
                while ((cin + sl - 1) / sl > kCcwMaxSlabs) sl *= 2;
                // fewer, fatter slices while the chip stays covered: every slab is re-read by all the M
                // tiles of its consumer
                while ((long)((cin + 2 * sl - 1) / (2 * sl)) * (op.M / 32) * nt >= m.ccw_min_blocks && 2 * sl <= cin && fits(2 * sl) &&
                       (op.cin1 == 0 || op.cin0 % (2 * sl) == 0))
                    sl *= 2;
                return fits(sl) ? sl : 0;
            };
            o.tile_rows = 32;
            if (op.Lout <= 16 && (batch + 16 / op.Lout - 1) / (16 / op.Lout) == (batch + 32 / op.Lout - 1) / (32 / op.Lout))
                o.tile_rows = 16;
            slice = try_rows(o.tile_rows);
            if (o.tile_rows == 32 && op.Lout <= 16 && m.ccw_prefer16) {
                // LDS-short 32-row tiles end up with twice the K slices (twice the slabs for the consumer
                // to add); two 16-row tiles re-read the weights from L2 instead
                const int s16 = try_rows(16);
                if (slice == 0 || s16 >= 2 * slice) { o.tile_rows = 16; slice = s16; }
            }
            if (slice == 0) return refuse(P, "wide layer: no K slice of at most 8 slabs fits LDS");
            const int spt_w = o.tile_rows / op.Lout;
            int min_cpg = slice;
            for (const CcInput* in : {&o.in0, &o.in1})
                if (in->kind == 3 && !convs[in->producer].norm.empty())
                    min_cpg = std::min(min_cpg, convs[in->producer].cout / 8);
            if ((long)spt_w * (slice / min_cpg) > kCcwMaxPairs) return refuse(P, "wide layer: more than 64 (sample, group) pairs per block");
        }
        if (op.cin1 > 0 && op.cin0 % slice != 0) return refuse(P, "a K slice would straddle the concat");     // a slice may not straddle the concat
        o.slice_ch = slice;
        o.kslices = (cin + slice - 1) / slice;
        if ((long)o.kslices * slice > op.cin_pad) return refuse(P, "weight image too short for whole K slices");    // weight image too short for whole slices
        // a (sample, group) pair of the output is normalised by its consumer: conv_cc / final_cc take at
        // most 1024 elements per pair, conv_ccw 8192 (checked again where the consumer is planned)
        if (!op.norm.empty() && (long)(op.cout / 8) * op.Lout > kCcwMaxPair) return refuse(P, "GroupNorm pair above 8192 elements");
        const int spt = windowed ? 1 : o.tile_rows / op.Lout;
        o.ntiles = windowed ? batch * (op.Lout / 32) : (batch + spt - 1) / spt;
        o.out_rows = op.kind == CONV_UP ? batch * 2 * op.Lout : batch * op.Lout;
        o.out_cols = op.kind == CONV_UP ? op.M / 2 : op.M;
        o.oslab = off;
        off += (long)o.kslices * o.out_rows * o.out_cols;
        if (op.ride) { o.orslab = off; off += (long)o.kslices * o.out_rows * o.out_cols; }
        if (o.wide) {
            o.lds_bytes = dad::ccw_lds_floats(slice, op.taps, op.Lin, op.Lout, o.tile_rows) * sizeof(float);
        } else {
            // the exchange tile needs 2 * 8 * 32 * 36 floats; operands XROWS + weight rows of slice + 4
            const size_t xs = slice + 4;
            const size_t k = (size_t)dad::cc_xrows(op.taps, op.Lin, op.Lout, o.tile_rows) * xs + (size_t)op.wtaps() * 32 * xs;
            const size_t e = (size_t)2 * 8 * o.tile_rows * 36;
            o.lds_bytes = std::max(k, e) * sizeof(float);
        }
        if (o.lds_bytes > dad::kLdsBytes) return refuse(P, "a launch does not fit LDS");
        // how the output gets finished
        if (op.res == -2) o.res_kind = 1;
        else if (op.res >= 0) {
            const int q = owner[op.res];
            if (q <= -2) { o.res_kind = 3; o.res_ride = -2 - q; }
            else if (q >= 0 && materialised[q]) { o.res_kind = 2; o.res_buf = op.res; }
            else if (q >= 0 && convs[q].kind == CONV_1X1 && convs[q].norm.empty() && P.ops[q].launched) {
                o.res_kind = 4; o.res_ride = q;          // the block's own 1x1 residual conv, still in pieces
            } else return refuse(P, "residual tensor neither finished nor a 1x1 conv in pieces");                             // residual not finished yet: not a plan we know
        }
        owner[op.dst] = (int)i;
    }
    P.final_producer = owner[m.plan.final_act];
    if (P.final_producer < 0 || materialised[P.final_producer]) return refuse(P, "final conv input already finished");
    // the streamed-weight form re-reads the weights once per N tile: measured against the batch-256
    // kernels it pays up to 4 plans of 32 positions (HalfCheetah 410 / 627 us per step at batch 1 / 4
    // against 597 / 653; 973 against 683 at batch 6)
    for (const CcOp& o : P.ops)
        if (o.launched && o.wide && (long)batch * c.horizon > m.ccw_max_rows) return refuse(P, "wide layers and more than ccw_max_rows rows");
    {   // final_cc_kernel normalises a pair in one wave's registers
        const ConvOp& f = convs[P.final_producer];
        if ((long)(f.cout / 8) * f.Lout > 1024) return refuse(P, "final conv: GroupNorm pair above 1024 elements");
        // (final_cc_kernel finishes its input with cc_build_input<false>: one round of CC_MAX_SLABS = 8 slabs)
        if (P.ops[P.final_producer].kslices > 8) return refuse(P, "final conv: more than 8 partial slabs");
    }
    P.slab_floats = off;
    P.ok = true;
    return P;
}

// ------------------------------------------------------------------ level-0 chain (conv_chain.hpp)
// downs.0.0 (first conv with the riding 1x1 residual conv, second conv), downs.0.1 (two convs) and the
// down-sampling conv as ONE launch, one block per sample.  Architecture conditions (the batch threshold and
// the per-call conditions — shared timestep, inference plan — are checked where it is launched):
struct ChainPlan { bool ok = false; int last = -1; int C = 0; };
inline ChainPlan chain_plan(const HostModel& m) {
    ChainPlan P;
    const dad_cfg& c = m.cfg;
    const std::vector<ConvOp>& v = m.plan.convs;
    if (m.precision != DAD_PREC_FP32 || c.kernel_size != 5 || c.horizon != 32 || c.n_levels < 2 ||
        (m.real_horizon > 0 && m.real_horizon != c.horizon)) return P;
    const int C = c.channels[0];
    if ((C != 32 && C != 64 && C != 128) || c.transition_dim > 16 || c.transition_dim == C || v.size() < 6) return P;
    // the plan order of build_plan: conv0 (+ride), its stand-alone 1x1 form, conv1, conv0', conv1', down
    if (!(v[0].ride && v[0].kind == CONV_K5 && v[0].src0 == -2 && v[0].cin1 == 0 && v[1].rider_of == 0 &&
          v[2].kind == CONV_K5 && v[2].res == v[1].dst && v[3].kind == CONV_K5 && v[4].kind == CONV_K5 &&
          v[4].res == v[3].src0 && v[5].kind == CONV_DOWN && v[5].src0 == v[4].dst))
        return P;
    for (int i : {0, 2, 3, 4, 5})
        if (v[i].cout != C || v[i].kc != 16 || v[i].x3 || v[i].bdir || v[i].gn_real > 0) return P;
    P.ok = true; P.last = 5; P.C = C;
    return P;
}

// activations, then whichever scratch the batch uses: split-K slabs or the CC partial-sum slabs
inline size_t workspace_bytes(const HostModel& m, int batch) {
    const CcPlan cc = cc_plan(m, batch);
    // (both: a small batch with per-row timesteps still runs the batch-256 kernels)
    const size_t scratch = std::max(cc.ok ? (size_t)cc.slab_floats : 0, (size_t)slab_floats_for(m, batch));
    return ((size_t)m.plan.floats_per_sample * (size_t)batch + scratch) * sizeof(float);
}

// Bytes the parameter arena must hold: packed weights, norms, tables, time-MLP weights, flags.
inline size_t arena_bytes_needed(const HostModel& m) {
    const dad_cfg& c = m.cfg;
    size_t floats = 0, allocs = 0;
    auto add = [&](size_t n) { floats += n + 64; ++allocs; };
    for (const ConvOp& op : m.plan.convs) {
        add((size_t)op.cin_pad * (op.taps + (op.rname.empty() ? 0 : 1)) * op.M);
        add(op.M);
        if (!op.rname.empty()) add(op.M);
        if (!op.norm.empty()) { add(op.cout); add(op.cout); }
        if (op.temb_off >= 0) { add((size_t)op.cout * c.time_dim); add(op.cout); }
    }
    add((size_t)c.transition_dim * c.dim); add(c.transition_dim);
    const size_t T = c.n_timesteps;
    add(T * c.dim); add(T * 4 * c.time_dim); add(T * c.time_dim);
    add(T * std::max(1, m.plan.temb_width));
    add((size_t)4 * c.time_dim * c.dim); add(4 * c.time_dim);
    add((size_t)c.time_dim * 4 * c.time_dim); add(c.time_dim);
    if (c.n_levels >= 2 && c.channels[0] <= 128)               // level-0 chain images (conv_chain.hpp)
        for (int i = 0; i < 6 && i < (int)m.plan.convs.size(); ++i) add((size_t)m.plan.convs[i].cin_pad * 6 * m.plan.convs[i].M);
    if (m.training) {
        for (const HostModel::BwdConv& b : m.bconvs)
            for (int s = 0; s < b.n; ++s) add((size_t)b.op[s].cin_pad * b.op[s].wtaps() * b.op[s].M);
        add((size_t)m.bfinal.cin_pad * m.bfinal.M);
        add((size_t)std::max(m.max_bwd_m, 2 * m.max_cout) + 64);
    }
    return floats * sizeof(float) + allocs * 256 + kMaxSplitTiles * sizeof(unsigned) + (1 << 16);
}

// SinusoidalPosEmb (temporal_unet.py:27-31) for every t in [0, T): fp32, in the reference's
// operation order (frequency = exp(j * -ln(1e4)/(half-1)) in fp32, argument = t * frequency).
inline std::vector<float> sinusoid_table(int T, int dim) {
    std::vector<float> emb((size_t)T * dim);
    const int half = dim / 2;
    const float scale = (float)(-(std::log(10000.0) / (half - 1)));
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < half; ++j) {
            const float f = std::exp((float)j * scale);
            const float arg = (float)t * f;
            emb[(size_t)t * dim + j] = std::sin(arg);
            emb[(size_t)t * dim + half + j] = std::cos(arg);
        }
    return emb;
}

}  // namespace dadhost
