#!/usr/bin/env python3
"""Per-layer efficiency table of one denoise step from a rocprofv3 --kernel-trace CSV of
profiles/pmc_target.py (batch-256 kernels: dad::conv_gemm_f32<...>):

    python3 profiles/layer_table.py <kernel_trace.csv> <arch> <batch>  >  profiles/rNN_layers_<arch>_b<batch>.md

Launch order = the plan order of csrc/host_plan.hpp build_plan (a residual 1x1 conv rides in its block's
first conv wherever the fused kernel exists: every block below 2048 channels at these batches).
Peak = 157.3 TFLOP/s (fp32 MFMA)."""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamics_aware_diffusion_amd.utils import synth  # noqa: E402

PEAK = 157.3e12


def launches(arch, H=32):
    od, ad, dim, mults, _ = synth.ARCHS[arch]
    td = od + ad
    ch = [dim * m for m in mults]
    out = []            # (name, flops per sample)

    def block(base, cin, cout, L):
        # the 1x1 residual conv is listed as an OPTIONAL launch: whether it rides in conv0 is decided per
        # batch (fused_at) and read off the kernel name of conv0 (template flag RES) in main()
        out.append((f"{base}.conv0 {cin}->{cout} k5 @{L}", 2 * cout * cin * 5 * L))
        if cin != cout:
            out.append((f"?{base}.res1x1 {cin}->{cout} @{L}", 2 * cout * cin * L))
        out.append((f"{base}.conv1 {cout}->{cout} k5 @{L}", 2 * cout * cout * 5 * L))

    L, cx = H, td
    for i, co in enumerate(ch):
        block(f"downs.{i}.0", cx, co, L)
        block(f"downs.{i}.1", co, co, L)
        if i < len(ch) - 1:
            out.append((f"downs.{i}.down {co} k3s2 @{L}->{L // 2}", 2 * co * co * 3 * (L // 2)))
            L //= 2
        cx = co
    block("mid1", cx, cx, L)
    block("mid2", cx, cx, L)
    for j in range(len(ch) - 1):
        lvl = len(ch) - 1 - j
        co = ch[lvl - 1]
        block(f"ups.{j}.0", cx + ch[lvl], co, L)
        block(f"ups.{j}.1", co, co, L)
        out.append((f"ups.{j}.up {co} convT k4s2 @{L}->{2 * L}", 2 * co * co * 4 * L))
        L *= 2
        cx = co
    out.append((f"final.conv0 {cx}->{dim} k5 @{L}", 2 * dim * cx * 5 * L))
    return out


def main():
    path, arch, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    plan = launches(arch)
    rows = [r for r in csv.DictReader(open(path)) if "dad::" in r["Kernel_Name"]]
    raw, cur = [], []
    for r in rows:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if "conv_gemm_f32" in r["Kernel_Name"]:
            cur.append((us, r["Kernel_Name"]))
        elif "final_posterior_kernel" in r["Kernel_Name"]:
            raw.append(cur)
            cur = []

    def targs(kn):
        return [a.strip() for a in kn[kn.index("<") + 1:kn.index(">")].split(",")]

    def match(step):
        """plan entries -> this step's launches; optional residual convs resolved by kernel names"""
        names, i, pending_ride = [], 0, None
        for name, f in plan:
            if name.startswith("?"):
                if pending_ride:                         # rides in the conv0 just seen: fold its FLOPs there
                    names[-1] = (names[-1][0] + " +1x1 ride", names[-1][1] + f)
                    continue
                name = name[1:]
            if i >= len(step):
                return None
            a = targs(step[i][1])
            pending_ride = len(a) >= 10 and a[9] in ("true", "1") and a[4] == "5"
            names.append((name, f))
            i += 1
        return names if i == len(step) else None

    steps, names = [], None
    for st in raw:
        nm = match(st)
        if nm is not None:
            names = nm
            steps.append(st)
    if not steps:
        raise SystemExit("no denoise step matched the plan (%d launches in the first step)" % (len(raw[0]) if raw else 0))
    steps = steps[1:] or steps                     # the first step pays cold caches
    print(f"| layer ({arch}, batch {batch}; {len(steps)} steps averaged) | kernel tile | us | GFLOP | TFLOP/s | % of fp32 MFMA peak |")
    print("|---|---|---|---|---|---|")
    tot_us = tot_fl = 0.0
    groups = collections.OrderedDict()
    for i, (name, f) in enumerate(names):
        us = sum(s[i][0] for s in steps) / len(steps)
        kn = steps[0][i][1]
        tile = kn[kn.index("<") + 1:kn.index(">")] if "<" in kn else kn
        fl = f * batch
        tot_us += us
        tot_fl += fl
        print(f"| {name} | {tile} | {us:.1f} | {fl / 1e9:.2f} | {fl / us / 1e6:.1f} | {100 * fl / (us * 1e-6) / PEAK:.0f} |")
    print(f"| **all {len(names)} conv launches** | | {tot_us:.1f} | {tot_fl / 1e9:.1f} | {tot_fl / tot_us / 1e6:.1f} | {100 * tot_fl / (tot_us * 1e-6) / PEAK:.0f} |")


if __name__ == "__main__":
    main()
