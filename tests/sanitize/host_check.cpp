// host_check.cpp — the host logic of libdad_hip.so (csrc/host_plan.hpp: architecture checks, launch
// plan, workspace layout, weight packing incl. split-f16 images, tile choice, grid split-K, LDS slot
// shifts, launch geometry) compiled WITHOUT HIP under -fsanitize=address,undefined and driven over
// the five architectures of the parity tests plus a deterministic sweep of the fuzz generator's
// space (tests/fuzz_parity.py).  Besides what the sanitizers catch, every launch is checked against
// the invariants the kernels assume.  Built and run by tests/test_host_logic.py (CPU suite); GPU
// sanitizers are not used anywhere.
#include <cinttypes>
#include <cstdio>
#include <random>

#include "../../dynamics_aware_diffusion_amd/csrc/host_plan.hpp"

using namespace dadhost;

static int g_failures = 0;
#define CHECK(cond, ...)                                                       \
    do {                                                                       \
        if (!(cond)) {                                                         \
            ++g_failures;                                                      \
            fprintf(stderr, "CHECK failed %s:%d: %s — ", __FILE__, __LINE__, #cond); \
            fprintf(stderr, __VA_ARGS__);                                      \
            fprintf(stderr, "\n");                                             \
        }                                                                      \
    } while (0)

struct Arch {
    const char* name;
    int td, dim, time_dim, horizon;
    std::vector<int> mults;
    bool pack;      // also load synthetic weights and build the packed images
    int ks = 5;     // TemporalUnet(kernel_size): 3, 5 or 7
    std::vector<int> real = {};   // dad_model_set_group_channels: level widths before zero-padding the groups
    int hreal = 0;                // dad_model_set_horizon: horizon before zero-padding (0: as given)
};

static float synth(uint64_t& state) {       // cheap deterministic values in (-1, 1)
    state = state * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((int64_t)(state >> 33) - (1ll << 30)) / (float)(1ll << 30);
}

// the conflict condition find_xswz promises, re-checked independently for the shifts it returns
static bool xswz_conflict_free(uint64_t packed, int L, int stride, int pad, int kp4, int BN) {
    static const int groups[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const int seg = L * stride + 2 * pad;
    for (int tn = 0; tn < BN / 32; ++tn)
        for (const auto& g : groups) {
            unsigned seen = 0;
            for (int lane : g) {
                const int n = tn * 32 + lane, sm = n / L, l = n % L;
                const int d = (int)((packed >> (4 * sm)) & 15);
                const unsigned bit = 1u << (((sm * seg + l * stride) * kp4 + d) & 15);
                if (seen & bit) return false;
                seen |= bit;
            }
        }
    return true;
}

static void check_arch(const Arch& a, int precision) {
    HostModel m;
    dad_cfg& c = m.cfg;
    c.transition_dim = a.td; c.dim = a.dim; c.time_dim = a.time_dim;
    c.n_levels = (int)a.mults.size();
    for (size_t i = 0; i < a.mults.size(); ++i) c.channels[i] = a.dim * a.mults[i];
    c.kernel_size = a.ks; c.horizon = a.horizon; c.n_timesteps = 20;
    c.predict_epsilon = c.clip_denoised = 1;
    m.precision = precision;               // (build_plan decides the kernel families from it)
    int rc = check_cfg(&c);
    if (rc != DAD_OK) { printf("  %-14s refused: %s\n", a.name, g_err); return; }
    for (size_t i = 0; i < a.real.size(); ++i) m.real_channels[i] = a.real[i];
    m.real_horizon = a.hreal;
    rc = build_plan(&m);
    if (rc != DAD_OK) { printf("  %-14s refused: %s\n", a.name, g_err); return; }
    if (a.hreal > 0) {
        int masked = 0;
        for (const ConvOp& op : m.plan.convs) {
            const int rows = op.kind == CONV_DOWN ? op.Lout : op.Lin;        // GEMM rows per sample
            CHECK(op.lreal >= 0 && op.lreal < rows + 1, "%s: lreal %d of %d", op.name.c_str(), op.lreal, rows);
            if (op.lreal > 0) ++masked;
            if (op.src0 == -2) CHECK(op.src_len == a.hreal, "%s: external rows %d", op.name.c_str(), op.src_len);
            else CHECK(op.src_len == 0, "%s: src_len on an internal tensor", op.name.c_str());
        }
        CHECK(masked == (int)m.plan.convs.size(), "%d of %zu convs know their real length", masked, m.plan.convs.size());
        CHECK(!cc_plan(m, 1).ok, "padded horizon took the small-batch kernels");
    }
    if (!a.real.empty()) {
        int padded_ops = 0;
        for (const ConvOp& op : m.plan.convs) {
            if (op.gn_real > 0) { ++padded_ops; CHECK(!op.norm.empty() && op.gn_real < op.cout / 8, "%s: gn_real %d of %d", op.name.c_str(), op.gn_real, op.cout / 8); }
        }
        CHECK(padded_ops > 0, "no conv knows its real group width");
        // (padded groups train: the training plan's convs know the real group width as well — the GroupNorm
        // backward divides its pair means by it)
        for (size_t i = 0; i < m.tplan.convs.size() && i < m.plan.convs.size(); ++i)
            CHECK(m.tplan.convs[i].gn_real == m.plan.convs[i].gn_real, "%s: training plan lost gn_real", m.plan.convs[i].name.c_str());
        CHECK(!cc_plan(m, 1).ok, "padded groups took the small-batch kernels");
    }
    const Plan& P = m.plan;
    CHECK(P.final_act >= 0 && P.final_act < (int)P.bufs.size(), "final_act %d", P.final_act);

    // buffers are disjoint and inside floats_per_sample
    for (size_t i = 0; i < P.bufs.size(); ++i) {
        CHECK(P.bufs[i].offset % 4 == 0, "buffer %zu offset %ld not float4 aligned", i, P.bufs[i].offset);
        CHECK(P.bufs[i].offset + P.bufs[i].per_sample <= P.floats_per_sample, "buffer %zu overruns", i);
        for (size_t j = i + 1; j < P.bufs.size(); ++j) {
            const bool apart = P.bufs[i].offset + P.bufs[i].per_sample <= P.bufs[j].offset ||
                               P.bufs[j].offset + P.bufs[j].per_sample <= P.bufs[i].offset;
            CHECK(apart, "buffers %zu and %zu overlap", i, j);
        }
    }
    // every op reads / writes buffers big enough for it, and never its own output
    double flops = 0;
    int temb_seen = 0;
    for (const ConvOp& op : P.convs) {
        auto cap = [&](int id) { return id >= 0 ? P.bufs[id].per_sample : 0L; };
        const long out_elems = (long)op.cout * (op.kind == CONV_UP ? 2 * op.Lin : op.Lout);
        CHECK(op.dst >= 0 && cap(op.dst) >= out_elems, "%s: dst too small", op.name.c_str());
        if (op.src0 >= 0) CHECK(cap(op.src0) >= (long)op.cin0 * op.Lin, "%s: src0 too small", op.name.c_str());
        if (op.src1 >= 0) CHECK(cap(op.src1) >= (long)op.cin1 * op.Lin, "%s: src1 too small", op.name.c_str());
        CHECK(op.dst != op.src0 && op.dst != op.src1, "%s: in-place conv", op.name.c_str());
        if (op.res >= 0) CHECK(op.res != op.dst && cap(op.res) >= out_elems, "%s: residual buffer", op.name.c_str());
        if (op.rdst >= 0) CHECK(op.rdst != op.dst && op.rdst != op.src0 && op.rdst != op.src1 &&
                                cap(op.rdst) >= out_elems, "%s: ride buffer", op.name.c_str());
        if (op.rider_of >= 0)
            CHECK(op.rider_of < (int)P.convs.size() && P.convs[op.rider_of].rdst == op.dst &&
                  P.convs[op.rider_of].rname == op.name, "%s: rider link", op.name.c_str());
        CHECK(op.cin_pad >= op.cin0 + op.cin1, "%s: cin_pad", op.name.c_str());
        if (op.temb_off >= 0) {
            CHECK(op.temb_off + op.cout <= P.temb_width, "%s: time table slice", op.name.c_str());
            temb_seen += op.cout;
        }
        flops += op.flops_per_sample;
    }
    CHECK(temb_seen == P.temb_width, "time table width %d vs %d", temb_seen, P.temb_width);
    CHECK(flops > 0, "flops");

    // weights: load synthetic tensors for every expected key, then pack every op
    if (a.pack) {
        uint64_t st = 12345;
        for (const auto& kv : m.expected) {
            HostTensor& t = m.raw[kv.first];
            t.shape = kv.second;
            size_t n = 1;
            for (int64_t d : kv.second) n *= (size_t)d;
            t.data.resize(n);
            for (float& v : t.data) v = synth(st) * 0.05f;
        }
        size_t arena = 0;
        for (ConvOp& op : m.plan.convs) {
            PackedOp po;
            rc = pack_op(&m, op, po);
            CHECK(rc == DAD_OK, "%s: pack_op: %s", op.name.c_str(), g_err);
            CHECK(po.w.size() == (size_t)op.cin_pad * op.wtaps() * op.M, "%s: packed size", op.name.c_str());
            CHECK(po.bias.size() == (size_t)op.M, "%s: bias size", op.name.c_str());
            CHECK(!op.ride || po.rbias.size() == (size_t)op.M, "%s: ride bias size", op.name.c_str());
            if (op.x3) CHECK(op.c1 > 0 && op.c2 > 0 && std::isfinite(op.c1), "%s: split scales", op.name.c_str());
            if (!op.x3) {     // the packed image is a permutation of the weights plus zero padding
                double s_in = 0, s_out = 0;
                for (float v : m.raw[op.name + ".weight"].data) s_in += v;
                if (op.ride) for (float v : m.raw[op.rname + ".weight"].data) s_in += v;
                for (float v : po.w) s_out += v;
                CHECK(std::fabs(s_in - s_out) <= 1e-6 * (1 + std::fabs(s_in)), "%s: packed checksum", op.name.c_str());
            }
            arena += (po.w.size() + po.bias.size() + po.rbias.size()) * sizeof(float) + 3 * 256;
        }
        CHECK(arena <= arena_bytes_needed(m), "arena estimate %zu < %zu", arena_bytes_needed(m), arena);
    }

    // launches: every batch size, every forced tile, with and without split-K / fusion
    const int batches[] = {1, 2, 3, 5, 8, 13, 31, 33, 64, 100, 128, 256, 257, 1024, 2048};
    long launches = 0;
    for (int force = -1; force < kNumTiles; ++force)
        for (int variant = 0; variant < 3; ++variant) {
            m.force_tile = force;
            m.split_enabled = variant != 1;
            m.fuse_residual = variant != 2;
            for (int B : batches) {
                const size_t ws = workspace_bytes(m, B);
                for (const ConvOp& op : m.plan.convs) {
                    LaunchGeom g;
                    rc = plan_launch(m, op, B, g);
                    if (rc != DAD_OK) {
                        // a refusal must be a message, never a crash; the heuristic (force = -1) must
                        // always find a launch for an architecture check_cfg accepted
                        CHECK(force >= 0, "%s B=%d: %s", op.name.c_str(), B, g_err);
                        continue;
                    }
                    ++launches;
                    const TileCfg& t = kTiles[g.cfg];
                    CHECK(tile_valid(op, g.cfg), "%s: invalid tile %d", op.name.c_str(), g.cfg);
                    CHECK(g.threads == 64 * (t.BM / 32) * (t.BN / 32) * t.SK && g.threads <= 1024, "threads %d", g.threads);
                    CHECK(g.lds_bytes <= dad::kLdsBytes, "%s: LDS %zu", op.name.c_str(), g.lds_bytes);
                    CHECK(g.kc % (op.x3 ? 16 : 8) == 0 && (g.kc / (op.x3 ? 16 : 8)) % t.SK == 0, "%s: K chunk %d", op.name.c_str(), g.kc);
                    const long tiles = (long)g.mtiles * g.ntiles_n;
                    CHECK((long)g.gx * g.gy * g.gz == tiles * g.split.kslices, "%s: grid %u %u %u vs %ld tiles x %d",
                          op.name.c_str(), g.gx, g.gy, g.gz, tiles, g.split.kslices);
                    CHECK(g.gy <= 65535 && g.gz <= 65535, "grid dims");
                    CHECK(g.ntiles_n * (t.BN / op.Lout) >= B, "%s: N tiles do not cover the batch", op.name.c_str());
                    if (g.split.kslices > 1) {
                        CHECK(tiles <= kMaxSplitTiles, "ticket table");
                        CHECK(!g.fused, "%s: ride under grid split-K", op.name.c_str());
                        const int nchunks = (op.cin0 + op.cin1 + g.kc - 1) / g.kc;
                        CHECK((g.split.kslices - 1) * g.split.chunks_per_slice < nchunks &&
                              g.split.kslices * g.split.chunks_per_slice >= nchunks, "%s: K slices", op.name.c_str());
                        const size_t slab_end = ((size_t)P.floats_per_sample * B + (size_t)g.split.slab_floats) * sizeof(float);
                        CHECK(slab_end <= ws, "%s B=%d: slab beyond the workspace", op.name.c_str(), B);
                    }
                    if (g.xcd_gn > 0) {
                        const int gm = 8 / g.xcd_gn;
                        CHECK(g.mtiles % gm == 0 && g.ntiles_n % g.xcd_gn == 0 && (1 << g.xcd_mts) == g.mtiles / gm &&
                              g.xcd_ntn == g.ntiles_n / g.xcd_gn, "%s: XCD order", op.name.c_str());
                        // the kernel's decode of a linear block id must be a bijection onto tiles
                        std::vector<char> hit((size_t)tiles, 0);
                        for (long wg = 0; wg < tiles; ++wg) {
                            const int cc = (int)(wg & 7), j = (int)(wg >> 3);
                            const int im = cc / g.xcd_gn, in = cc - im * g.xcd_gn;
                            const int mt = (im << g.xcd_mts) + (j & ((1 << g.xcd_mts) - 1));
                            const int nt = in * g.xcd_ntn + (j >> g.xcd_mts);
                            CHECK(mt < g.mtiles && nt < g.ntiles_n, "%s: XCD decode out of range", op.name.c_str());
                            if (mt < g.mtiles && nt < g.ntiles_n) hit[(size_t)mt * g.ntiles_n + nt]++;
                        }
                        for (char h : hit) CHECK(h == 1, "%s: XCD decode not a bijection", op.name.c_str());
                    }
                    if (g.xswz != 0) {
                        const int pad = op.taps / 2, kp4 = (g.kc + 4) / 4;
                        CHECK(xswz_conflict_free(g.xswz, op.Lout, op.stride, pad, kp4, t.BN), "%s: slot shifts conflict", op.name.c_str());
                        const int S = t.BN / op.Lout;
                        for (int s = 0; s + 1 < S; ++s) {
                            const int d0 = (int)((g.xswz >> (4 * s)) & 15), d1 = (int)((g.xswz >> (4 * (s + 1))) & 15);
                            CHECK(d0 - d1 <= pad * kp4, "%s: shift pushes a sample onto its neighbour", op.name.c_str());
                        }
                        // the stage was sized with kXSwzPad floats of slack for shifts of at most 15 slots
                        CHECK(15 * 4 <= dad::kXSwzPad, "slot shift slack");
                    }
                    if (g.fused) CHECK(op.ride && op.kind == CONV_K5 && (op.taps & 1) && op.stride == 1 && !op.x3 && !op.bdir, "%s: ride", op.name.c_str());
                }
            }
        }
    // small-batch (consumer-combine) plans: slabs disjoint and inside the workspace, inputs finished
    // before they are read finished, slices whole groups, every launch fits LDS
    m.force_tile = -1; m.split_enabled = true; m.fuse_residual = true;
    long cc_plans = 0;
    for (int B : {1, 2, 3, 5, 8, 13, 16, 33}) {
        const CcPlan cc = cc_plan(m, B);
        if (!cc.ok) continue;
        ++cc_plans;
        CHECK(precision == DAD_PREC_FP32 && (long)B * c.horizon <= m.cc_max_rows, "CC plan outside its domain");
        CHECK(workspace_bytes(m, B) >= ((size_t)P.floats_per_sample * B + (size_t)cc.slab_floats) * sizeof(float), "CC workspace");
        std::vector<std::pair<long, long>> spans;
        std::vector<char> finished(P.convs.size(), 0);
        for (size_t i = 0; i < P.convs.size(); ++i) {
            const ConvOp& op = P.convs[i];
            const CcOp& o = cc.ops[i];
            if (!o.launched) { CHECK(op.rider_of >= 0 && P.convs[op.rider_of].ride, "%s not launched", op.name.c_str()); continue; }
            CHECK(o.kslices >= 1 && o.kslices <= (o.wide ? kCcwMaxSlabs : kCcMaxSlabs) && o.slice_ch % 32 == 0 &&
                  (o.wide || o.slice_ch <= kCcMaxSlice),
                  "%s: slices %d x %d", op.name.c_str(), o.kslices, o.slice_ch);
            CHECK((long)o.kslices * o.slice_ch >= op.cin0 + op.cin1 && (long)o.kslices * o.slice_ch <= op.cin_pad,
                  "%s: slices do not cover the input channels", op.name.c_str());
            CHECK(op.cin1 == 0 || op.cin0 % o.slice_ch == 0, "%s: slice straddles the concat", op.name.c_str());
            CHECK(o.lds_bytes <= dad::kLdsBytes, "%s: CC LDS %zu", op.name.c_str(), o.lds_bytes);
            if (o.wide) {
                // conv_ccw.hpp's contract: 16-channel granules, channel counts in float4, LDS sized by its formula
                CHECK((op.kc == 16 || op.bdir) && op.cin0 % 4 == 0 && op.cin1 % 4 == 0 && op.src0 != -2,
                      "%s: wide conv on an input it cannot stage", op.name.c_str());
                CHECK(o.lds_bytes == dad::ccw_lds_floats(o.slice_ch, op.taps, op.Lin, op.Lout, o.tile_rows) * sizeof(float),
                      "%s: wide LDS size", op.name.c_str());
            } else {
                CHECK(op.kind != CONV_1X1, "%s: conv_cc has no 1x1 form", op.name.c_str());
                CHECK((size_t)(o.slice_ch / 16) * op.wtaps() * 128 <= (size_t)6 * 512, "%s: weight staging registers", op.name.c_str());
                CHECK(o.kslices <= 8 || o.slice_ch == kCcMaxSlice, "%s: more than 8 slabs below the widest slice", op.name.c_str());
            }
            if (op.Lout > 32) {           // windowed tiles: 32 rows of one sample each, stride-1 k-tap convs of conv_cc only
                CHECK(!o.wide && op.kind == CONV_K5 && op.stride == 1 && o.tile_rows == 32 && op.Lout % 32 == 0 &&
                      o.ntiles == B * (op.Lout / 32), "%s: windowed tiles", op.name.c_str());
                CHECK(o.lds_bytes >= ((size_t)(32 + 2 * (op.taps / 2)) * (o.slice_ch + 4) + (size_t)op.wtaps() * 32 * (o.slice_ch + 4)) * sizeof(float),
                      "%s: windowed LDS", op.name.c_str());
            } else {
                CHECK((o.tile_rows == 16 || o.tile_rows == 32) && o.tile_rows % op.Lout == 0 &&
                      o.ntiles * (o.tile_rows / op.Lout) >= B, "%s: tiles do not cover the batch", op.name.c_str());
            }
            if (o.wide) CHECK(op.Lin <= 32 && op.Lout <= 32, "%s: conv_ccw beyond 32 positions", op.name.c_str());
            const long n = (long)o.kslices * o.out_rows * o.out_cols;
            spans.push_back({o.oslab, o.oslab + n});
            if (o.orslab >= 0) spans.push_back({o.orslab, o.orslab + n});
            CHECK((o.orslab >= 0) == op.ride, "%s: ride slab", op.name.c_str());
            for (const CcInput* in : {&o.in0, &o.in1}) {
                if (in->kind == 3) {
                    CHECK(in->producer >= 0 && in->producer < (int)i && cc.ops[in->producer].launched && !finished[in->producer],
                          "%s: reads %d in pieces twice", op.name.c_str(), in->producer);
                    finished[in->producer] = 1;
                    const ConvOp& q = P.convs[in->producer];
                    if (!q.norm.empty()) CHECK(o.slice_ch % (q.cout / 8) == 0, "%s: slice splits a GroupNorm group", op.name.c_str());
                    const CcOp& qo = cc.ops[in->producer];
                    if (qo.res_kind == 3) CHECK(cc.ops[qo.res_ride].orslab >= 0 && qo.res_ride < in->producer, "%s: ride source", op.name.c_str());
                    if (qo.res_kind == 4) CHECK(cc.ops[qo.res_ride].launched && P.convs[qo.res_ride].kind == CONV_1X1 && qo.res_ride < in->producer,
                                                "%s: residual conv source", op.name.c_str());
                    const long pair = q.norm.empty() ? 0 : (long)(q.cout / 8) * op.Lin;
                    const int spt = op.Lout > 32 ? 1 : o.tile_rows / op.Lout;
                    if (o.wide) {
                        CHECK(qo.kslices <= kCcwMaxSlabs && (qo.res_kind < 3 || cc.ops[qo.res_ride].kslices <= kCcwMaxSlabs),
                              "%s: wide conv fed more than %d slabs", op.name.c_str(), kCcwMaxSlabs);
                        CHECK(pair <= kCcwMaxPair, "%s: pair of %ld elements", op.name.c_str(), pair);
                        if (!q.norm.empty()) CHECK((long)spt * (o.slice_ch / (q.cout / 8)) <= kCcwMaxPairs, "%s: pairs per block", op.name.c_str());
                    } else {
                        CHECK(pair <= 1024, "%s: conv_cc normalises at most 1024 elements per pair (%ld)", op.name.c_str(), pair);
                    }
                }
            }
        }
        CHECK(cc.final_producer >= 0 && !finished[cc.final_producer], "final conv output already finished");
        if (cc.final_producer >= 0) {
            const ConvOp& f = P.convs[cc.final_producer];
            CHECK((long)(f.cout / 8) * f.Lout <= 1024, "final_cc normalises at most 1024 elements per pair");
        }
        std::sort(spans.begin(), spans.end());
        for (size_t k = 0; k + 1 < spans.size(); ++k) CHECK(spans[k].second <= spans[k + 1].first, "CC slabs overlap");
        if (!spans.empty()) CHECK(spans.front().first >= 0 && spans.back().second <= cc.slab_floats, "CC slabs outside their region");
    }
    // training plan + backward plan (fp32): every tensor of the forward owns its buffer, the data-gradient
    // launches find a tile at every batch, their packed images are permutations of the weights, and the flat
    // gradient buffer holds every conv / GroupNorm / final-conv tensor exactly once
    long bwd_launches = 0;
    if (precision == DAD_PREC_FP32 && training_refusal(m) == nullptr) {
        const Plan& T = m.tplan;
        CHECK(T.convs.size() == P.convs.size() && m.bconvs.size() == T.convs.size(), "training plan size");
        std::vector<int> writers(T.bufs.size(), 0);
        for (const ConvOp& op : T.convs) {
            ++writers[op.dst];
            CHECK(op.norm.empty() == (op.pre < 0) && (op.pre < 0) == (op.stats < 0), "%s: pre / stats buffers", op.name.c_str());
            if (op.pre >= 0) {
                ++writers[op.pre]; ++writers[op.stats];
                CHECK(T.bufs[op.pre].per_sample >= (long)op.cout * op.Lout && T.bufs[op.stats].per_sample >= 16, "%s: pre / stats size", op.name.c_str());
            }
        }
        for (size_t i = 0; i < writers.size(); ++i) CHECK(writers[i] <= 1, "training buffer %zu written by %d launches", i, writers[i]);
        for (size_t i = 0; i < T.bufs.size(); ++i) CHECK(T.bufs[i].offset + T.bufs[i].per_sample <= T.floats_per_sample, "training buffer %zu overruns", i);
        std::map<std::string, int> seen;
        long end = 0;
        for (const auto& gs : m.grad_slots) {
            ++seen[gs.key];
            CHECK(gs.offset >= end && gs.offset % 4 == 0, "gradient slot %s overlaps", gs.key.c_str());
            end = gs.offset + gs.numel;
            auto ex = m.expected.find(gs.key);
            CHECK(ex != m.expected.end(), "gradient slot %s is not a parameter", gs.key.c_str());
            if (ex != m.expected.end()) {
                long n = 1;
                for (int64_t d : ex->second) n *= (long)d;
                CHECK(n == gs.numel, "gradient slot %s: %ld elements, parameter has %ld", gs.key.c_str(), gs.numel, n);
            }
        }
        CHECK(end <= m.grad_numel, "gradient buffer too short");
        for (const auto& kv : m.expected)
            if (kv.first.find("time_mlp.") == std::string::npos) CHECK(seen[kv.first] == 1, "parameter %s has %d gradient slots", kv.first.c_str(), seen[kv.first]);
        for (size_t i = 0; i < T.convs.size(); ++i) {
            const ConvOp& f = T.convs[i];
            const HostModel::BwdConv& b = m.bconvs[i];
            CHECK(b.n == (f.cin1 > 0 ? 2 : 1), "%s: %d data-gradient launches", f.name.c_str(), b.n);
            int covered = 0;
            for (int k = 0; k < b.n; ++k) {
                covered += b.c_n[k];
                CHECK(b.op[k].cout >= b.c_n[k] && b.op[k].cout % 32 == 0 && b.op[k].cin0 == f.cout, "%s: data-gradient shape", f.name.c_str());
                for (int B : {1, 3, 9, 32, 256}) {
                    LaunchGeom g;
                    rc = plan_launch(m, b.op[k], B, g);
                    CHECK(rc == DAD_OK, "%s B=%d: %s", b.op[k].name.c_str(), B, g_err);
                    if (rc == DAD_OK) { ++bwd_launches; CHECK(g.lds_bytes <= dad::kLdsBytes && !g.fused, "%s: geometry", b.op[k].name.c_str()); }
                }
                if (a.pack) {
                    std::vector<float> img;
                    rc = pack_bwd_op(&m, f, b, k, img);
                    CHECK(rc == DAD_OK && img.size() == (size_t)b.op[k].cin_pad * b.op[k].wtaps() * b.op[k].M, "%s: data-gradient image", f.name.c_str());
                    if (b.n == 1) {        // one source: the image is a permutation of the weight tensor plus zeros
                        double s_in = 0, s_out = 0;
                        for (float v : m.raw[f.name + ".weight"].data) s_in += v;
                        for (float v : img) s_out += v;
                        CHECK(std::fabs(s_in - s_out) <= 1e-5 * (1 + std::fabs(s_in)), "%s: data-gradient image checksum", f.name.c_str());
                    }
                }
            }
            CHECK(covered == f.cin0 + f.cin1, "%s: data gradients cover %d of %d input channels", f.name.c_str(), covered, f.cin0 + f.cin1);
        }
        for (int B : {1, 9, 256}) {
            LaunchGeom g;
            rc = plan_launch(m, m.bfinal, B, g);
            CHECK(rc == DAD_OK, "final data gradient B=%d: %s", B, g_err);
        }
        if (a.pack) {
            std::vector<float> img;
            CHECK(pack_bwd_final(&m, img) == DAD_OK && img.size() == (size_t)m.bfinal.cin_pad * m.bfinal.M, "final data-gradient image");
        }
    }
    (void)bwd_launches;
    printf("  %-14s prec=%d: %zu launches in the plan, %zu buffers, %ld floats/sample, %ld launch geometries checked\n",
           a.name, precision, P.convs.size(), P.bufs.size(), P.floats_per_sample, launches);
    (void)cc_plans;
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && std::string(argv[1]) == "--quick";
    std::vector<Arch> archs = {
        {"tiny", 6, 32, 32, 32, {1, 2, 4}, true},
        {"tiny4", 8, 32, 32, 32, {1, 2, 2, 4}, true},
        {"tiny_td64", 6, 32, 64, 32, {1, 2, 4}, true},
        {"pointmaze", 6, 128, 128, 32, {1, 2, 4}, true},
        {"halfcheetah", 23, 256, 256, 32, {1, 4, 8}, !quick},
        {"door", 67, 256, 256, 32, {1, 2, 4, 8}, !quick},
        {"shrink", 6, 32, 32, 32, {1, 4, 2}, true},
        {"shrink2", 5, 32, 32, 32, {1, 4, 2, 1}, true},
        {"single", 3, 32, 32, 32, {1}, true},
        {"h8", 8, 128, 128, 8, {1, 4}, true},
        {"h64", 23, 64, 64, 64, {1, 1, 2}, true},
        {"h128", 6, 128, 128, 128, {1, 2, 4}, true},
        {"h128w", 23, 256, 256, 128, {1, 4, 8}, false},
        {"pointmaze_k3", 6, 128, 128, 32, {1, 2, 4}, true, 3},
        {"tiny_k7", 8, 64, 64, 32, {1, 2}, true, 7},
        {"wide_k3", 23, 256, 256, 32, {1, 4, 8}, !quick, 3},
        {"wide_k7", 9, 256, 256, 16, {1, 8}, false, 7},
        {"h128_k7", 6, 128, 128, 128, {1, 2}, true, 7},
        {"shrink_k7", 5, 32, 32, 32, {1, 4, 2}, true, 7},
        {"d48_padded", 6, 64, 48, 32, {1, 2}, true, 5, {48, 96}},          // --dim 48 as the engine runs it
        {"d40_padded", 5, 64, 40, 32, {1, 2, 2}, true, 5, {40, 80, 120}},   // 40 / 80 / 120 -> 64 / 128 / 128
        {"h24_padded", 6, 32, 32, 32, {1, 2, 4}, true, 5, {}, 24},          // horizon 24 as the engine runs it (32)
        {"h100_padded", 11, 64, 64, 128, {1, 2, 4}, true, 5, {}, 100},      // 100 / 50 / 25 positions in 128 / 64 / 32
        {"h48_d48", 6, 64, 48, 64, {1, 2}, true, 3, {48, 96}, 48},          // both paddings + kernel_size 3
    };
    // the fuzz generator's space (tests/fuzz_parity.py), deterministic sweep
    std::mt19937 rng(7);
    auto pick = [&](std::initializer_list<int> v) { return *(v.begin() + rng() % v.size()); };
    std::vector<std::string> names;
    names.reserve(64);
    for (int it = 0; it < 40; ++it) {
        const int dim = pick({32, 64, 128, 256});
        const int nlev = pick({1, 2, 3, 4});
        std::vector<int> mults{1};
        int mx = 1;
        for (int i = 1; i < nlev; ++i) { mults.push_back(pick({1, 2, 4, 8})); mx = std::max(mx, mults.back()); }
        const int H = pick({8, 16, 32, 64});
        const int td = 2 + (int)(rng() % 23);
        if (mx * dim > 2048) continue;
        names.push_back("fuzz" + std::to_string(it));
        archs.push_back({names.back().c_str(), td, dim, dim, H, mults, mx * dim <= 512, it % 4 == 3 ? pick({3, 7}) : 5});
    }
    for (const Arch& a : archs)
        for (int prec : {DAD_PREC_FP32, DAD_PREC_F16X3}) check_arch(a, prec);

    // refusals are messages
    dad_cfg bad{};
    bad.transition_dim = 6; bad.dim = 48; bad.time_dim = 48; bad.n_levels = 2;
    bad.channels[0] = 48; bad.channels[1] = 96; bad.kernel_size = 5; bad.horizon = 32; bad.n_timesteps = 10;
    CHECK(check_cfg(&bad) == DAD_E_INVALID && g_err[0] != 0, "48 channels accepted");
    bad.dim = 32; bad.channels[0] = 32; bad.channels[1] = 64; bad.horizon = 4;
    CHECK(check_cfg(&bad) == DAD_E_INVALID, "horizon 4 with two levels accepted");
    CHECK(check_cfg(nullptr) == DAD_E_INVALID, "null cfg accepted");

    // split-f16 images: round trip hi + lo * 2^-11 reproduces the scaled weight to 2^-22
    {
        std::vector<float> w(64);
        uint64_t st = 99;
        for (float& v : w) v = synth(st) * 0.3f;
        w[5] = 0.0f; w[6] = 1e-8f;
        std::vector<float> img = w;
        const int s = split_f16_image(img);
        for (size_t g = 0; g < w.size(); g += 16)
            for (int j = 0; j < 16; ++j) {
                uint16_t hb, lb;
                std::memcpy(&hb, (const char*)&img[g] + 2 * j, 2);
                std::memcpy(&lb, (const char*)&img[g + 8] + 2 * j, 2);
                _Float16 h, l;
                std::memcpy(&h, &hb, 2); std::memcpy(&l, &lb, 2);
                const double back = ((double)(float)h + (double)(float)l / 2048.0) * std::ldexp(1.0, -s);
                CHECK(std::fabs(back - w[g + j]) <= std::ldexp(std::fabs((double)w[g + j]), -21) + 1e-12,
                      "split image element %zu: %g vs %g", g + j, back, (double)w[g + j]);
            }
    }
    // sinusoid table: shape and a few exact entries
    {
        const std::vector<float> e = sinusoid_table(10, 32);
        CHECK(e.size() == 320 && e[0] == 0.0f && e[16] == 1.0f, "sinusoid table t=0");
        CHECK(std::fabs(e[32] - std::sin(1.0f)) < 1e-7, "sinusoid table t=1");
    }
    if (g_failures) { fprintf(stderr, "%d host-logic checks FAILED\n", g_failures); return 1; }
    printf("host logic ok\n");
    return 0;
}
