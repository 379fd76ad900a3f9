#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs of profiles/collect.sh into the small files committed under
profiles/:  <tag>_kernel_stats.csv (rocprofv3 --stats table), <tag>_summary.json (per-kernel
averages, HBM traffic per launch from the PMC passes with the gfx950 corrections of
MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request on wide reads, so it is doubled;
both counters are in KiB) and traffic.json (what bench.py reports as roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))


def one(pattern):
    """Newest file matching `<pass>_*/...`: the pass directory must be exactly <pass>_<HHMMSS>, so that
    "trace_*" does not also match "trace_door_b1_*" (newest by the time stamp in the name: file
    mtimes are those of the merge back from the GPU box)."""
    import re
    head = pattern.split("/")[0]
    assert head.endswith("_*"), pattern
    want = re.compile("^" + re.escape(head[:-2]) + r"_(\d{6,10})$")
    hits = []
    for f in glob.glob(os.path.join(src, pattern), recursive=True):
        m = want.match(os.path.relpath(f, src).split(os.sep)[0])
        if m:
            hits.append((m.group(1), f))
    return max(hits)[1] if hits else None


stats = one("trace_*/**/*_kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(here, f"{tag}_kernel_stats.csv"))

summary = {"tag": tag,
           "command": "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-configs",
           "pmc_command": "python3 profiles/pmc_target.py  (one eager sampling loop)",
           "failed_passes": sorted(os.path.basename(p) for p in glob.glob(os.path.join(src, "*.TIMEOUT")))}
trace = one("trace_*/**/*_kernel_trace.csv")
if trace:
    rows = [r for r in csv.DictReader(open(trace)) if "dad::" in r["Kernel_Name"]]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    summary["kernels_us"] = {k: {"calls": n, "total_us": t, "avg_us": t / n} for k, (n, t) in agg.items()}
    conv = [(n, t) for k, (n, t) in agg.items() if "conv_gemm_f32" in k]
    steps = sum(n for k, (n, t) in agg.items() if "final_posterior_kernel" in k)     # one per denoise step
    summary["conv_gemm_all_variants"] = {"calls": sum(n for n, _ in conv),
                                         "total_us": sum(t for _, t in conv),
                                         "avg_us": sum(t for _, t in conv) / max(1, sum(n for n, _ in conv)),
                                         "launches_per_denoise_step": sum(n for n, _ in conv) / max(1, steps),
                                         "us_per_denoise_step": sum(t for _, t in conv) / max(1, steps)}
    summary["all_kernels_us_per_denoise_step"] = sum(t for _, (n, t) in agg.items()) / max(1, steps)


def pmc(dirname, counter):
    f = one(f"{dirname}_*/**/*_counter_collection.csv")
    if not f:
        return None
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if "conv_gemm_f32" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot["conv"] += float(r["Counter_Value"])
            n["conv"] += 1
    return (tot["conv"], n["conv"]) if n["conv"] else None


fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
if fetch and write:
    per_launch = (2.0 * fetch[0] / fetch[1] + write[0] / write[1]) * 1024.0
    summary["hbm_traffic"] = {
        "FETCH_SIZE_KiB_per_launch_raw": fetch[0] / fetch[1],
        "WRITE_SIZE_KiB_per_launch": write[0] / write[1],
        "bytes_per_conv_launch": per_launch,
        "note": "FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request); includes Infinity-Cache hits",
    }
    tpath = os.path.join(here, "traffic.json")
    cur = json.load(open(tpath)) if os.path.exists(tpath) else {}
    cur["pointmaze_b256"] = per_launch
    json.dump(cur, open(tpath, "w"), indent=1)

sq = one("pmc_sq_*/**/*_counter_collection.csv")
if sq:
    tot = collections.defaultdict(float)
    for r in csv.DictReader(open(sq)):
        if "conv_gemm_f32" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
    summary["sq_counters_conv_total"] = dict(tot)
    if tot.get("SQ_WAVE_CYCLES"):
        wc = tot["SQ_WAVE_CYCLES"]
        summary["sq_ratios"] = {k: tot[k] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in tot}
    if tot.get("SQ_BUSY_CU_CYCLES") and tot.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # MFMA utilisation: matrix-pipe busy cycles (summed over the 4 SIMDs of a CU) over the
        # cycles the CU had work — the in-kernel share of time the matrix cores were issuing
        summary["mfma_util"] = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * tot["SQ_BUSY_CU_CYCLES"])
    if tot.get("SQ_LDS_IDX_ACTIVE"):
        summary["lds_bank_conflict_frac"] = tot.get("SQ_LDS_BANK_CONFLICT", 0.0) / tot["SQ_LDS_IDX_ACTIVE"]
# ---- the split-f16 arithmetic (bench.py --precision f16x3): its own stats table and traffic
x3stats = one("x3_trace_*/**/*_kernel_stats.csv")
if x3stats:
    shutil.copy(x3stats, os.path.join(here, f"{tag}_kernel_stats_f16x3.csv"))
x3trace = one("x3_trace_*/**/*_kernel_trace.csv")
if x3trace:
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(x3trace)):
        if "dad::" in r["Kernel_Name"]:
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    conv = [(n, t) for k, (n, t) in agg.items() if "conv_gemm_f32" in k]
    x3 = {"command": "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --precision f16x3",
          "kernels_us": {k: {"calls": n, "total_us": t, "avg_us": t / n} for k, (n, t) in agg.items()},
          "conv_gemm_all_variants": {"calls": sum(n for n, _ in conv), "total_us": sum(t for _, t in conv),
                                     "avg_us": sum(t for _, t in conv) / max(1, sum(n for n, _ in conv))}}
    f3, w3 = pmc("x3_fetch", "FETCH_SIZE"), pmc("x3_write", "WRITE_SIZE")
    if f3 and w3:
        per_launch = (2.0 * f3[0] / f3[1] + w3[0] / w3[1]) * 1024.0
        x3["hbm_traffic"] = {"FETCH_SIZE_KiB_per_launch_raw": f3[0] / f3[1],
                             "WRITE_SIZE_KiB_per_launch": w3[0] / w3[1],
                             "bytes_per_conv_launch": per_launch}
        tpath = os.path.join(here, "traffic.json")
        cur = json.load(open(tpath)) if os.path.exists(tpath) else {}
        cur["pointmaze_b256:f16x3"] = per_launch
        json.dump(cur, open(tpath, "w"), indent=1)
    summary["f16x3"] = x3

# ---- the other BASELINE configurations: per-kernel averages and HBM traffic per denoise step
def pmc_all(dirname, counter):
    f = one(f"{dirname}_*/**/*_counter_collection.csv")
    if not f:
        return None
    tot = 0.0
    for r in csv.DictReader(open(f)):
        if "dad::" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
    return tot


others = {}
for arch, batch, workload in (("halfcheetah", 128, "halfcheetah_b128"), ("door", 128, "door_b128"),
                              ("pointmaze", 1, "pointmaze_b1"), ("halfcheetah", 1, "halfcheetah_b1"),
                              ("door", 1, "door_b1")):
    tag2 = f"{arch}_b{batch}"
    tr = one(f"trace_{tag2}_*/**/*_kernel_trace.csv")
    if not tr:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(tr)):
        if "dad::" in r["Kernel_Name"]:
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    steps = sum(n for k, (n, t) in agg.items() if "final_" in k)
    conv = [(n, t) for k, (n, t) in agg.items() if "conv_" in k]
    entry = {"command": f"python3 profiles/pmc_target.py --arch {arch} --batch {batch} --denoise-steps 20",
             "denoise_steps_traced": steps,
             "conv_launches_per_denoise_step": sum(n for n, _ in conv) / max(1, steps),
             "conv_avg_us": sum(t for _, t in conv) / max(1, sum(n for n, _ in conv)),
             "all_kernels_us_per_denoise_step": sum(t for _, (n, t) in agg.items()) / max(1, steps),
             "kernels_us": {k: {"calls": n, "avg_us": t / n} for k, (n, t) in agg.items()}}
    st = one(f"trace_{tag2}_*/**/*_kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(here, f"{tag}_kernel_stats_{tag2}.csv"))
    fe, wr = pmc_all(f"fetch_{tag2}", "FETCH_SIZE"), pmc_all(f"write_{tag2}", "WRITE_SIZE")
    if fe is not None and wr is not None and steps:
        nconv = max(1, sum(n for n, _ in conv))
        per_step = (2.0 * fe + wr) * 1024.0 / steps
        entry["hbm_bytes_per_denoise_step"] = per_step
        entry["hbm_bytes_per_conv_launch"] = (2.0 * fe + wr) * 1024.0 / nconv
        tpath = os.path.join(here, "traffic.json")
        cur = json.load(open(tpath)) if os.path.exists(tpath) else {}
        cur[workload] = entry["hbm_bytes_per_conv_launch"]
        json.dump(cur, open(tpath, "w"), indent=1)
    others[tag2] = entry
# ---- BASELINE config 3: the projected loop — conv launches + the projection kernels of every step
tr = one("trace_pointmaze_proj_*/**/*_kernel_trace.csv")
if tr:
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(tr)):
        if "dad::" in r["Kernel_Name"] or "copyBuffer" in r["Kernel_Name"]:
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    steps = sum(n for k, (n, t) in agg.items() if "final_" in k)
    conv = [(n, t) for k, (n, t) in agg.items() if "conv_" in k]
    proj = {k: {"calls": n, "avg_us": t / n} for k, (n, t) in agg.items() if "project" in k}
    entry = {"command": "python3 profiles/pmc_target.py --project   (PointMaze B=256, projection after every step, D = 196)",
             "denoise_steps_traced": steps,
             "conv_launches_per_denoise_step": sum(n for n, _ in conv) / max(1, steps),
             "projection_kernels": proj,
             "projection_us_per_denoise_step": sum(v["calls"] * v["avg_us"] for v in proj.values()) / max(1, steps),
             "all_kernels_us_per_denoise_step": sum(t for k, (n, t) in agg.items() if "dad::" in k) / max(1, steps)}
    st = one("trace_pointmaze_proj_*/**/*_kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(here, f"{tag}_kernel_stats_pointmaze_proj.csv"))

    def pmc_kind(dirname, counter, needle):
        f = one(f"{dirname}_*/**/*_counter_collection.csv")
        if not f:
            return None
        tot = 0.0
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"] and r["Counter_Name"] == counter:
                tot += float(r["Counter_Value"])
        return tot
    fe, wr = pmc_kind("fetch_pointmaze_proj", "FETCH_SIZE", "dad::"), pmc_kind("write_pointmaze_proj", "WRITE_SIZE", "dad::")
    fp, wp = pmc_kind("fetch_pointmaze_proj", "FETCH_SIZE", "project"), pmc_kind("write_pointmaze_proj", "WRITE_SIZE", "project")
    if fe is not None and wr is not None and steps:
        nconv = max(1, sum(n for n, _ in conv))
        entry["hbm_bytes_per_denoise_step"] = (2.0 * fe + wr) * 1024.0 / steps
        entry["projection_hbm_bytes_per_denoise_step"] = (2.0 * (fp or 0.0) + (wp or 0.0)) * 1024.0 / steps
        # what bench.py reports as roofline.traffic for config 3: every kernel's bytes of a step (the
        # projection's included), per conv launch
        entry["hbm_bytes_per_conv_launch_incl_projection"] = (2.0 * fe + wr) * 1024.0 / nconv
        tpath = os.path.join(here, "traffic.json")
        cur = json.load(open(tpath)) if os.path.exists(tpath) else {}
        cur["pointmaze_proj_t500_b256"] = entry["hbm_bytes_per_conv_launch_incl_projection"]
        json.dump(cur, open(tpath, "w"), indent=1)
    others["pointmaze_proj_b256"] = entry
st = one("trace_train_*/**/*_kernel_stats.csv")
if st:
    shutil.copy(st, os.path.join(here, f"{tag}_kernel_stats_train_step.csv"))
if others:
    summary["other_configs"] = others

bj = os.path.join(src, "bench.json")
if os.path.exists(bj) and os.path.getsize(bj):
    summary["bench_line"] = json.loads(open(bj).read().strip().splitlines()[-1])
    # the bench of this collection ran before traffic.json was rewritten from its own PMC passes:
    # put the fresh figure in, as every later bench run will report it
    if "hbm_traffic" in summary and summary["bench_line"].get("roofline"):
        summary["bench_line"]["roofline"]["traffic"] = summary["hbm_traffic"]["bytes_per_conv_launch"]
json.dump(summary, open(os.path.join(here, f"{tag}_summary.json"), "w"), indent=1)
print("wrote", f"{tag}_summary.json")
