#!/usr/bin/env python3
"""Where one training step's time goes, from a rocprofv3 --kernel-trace CSV of profiles/train_step_timing.py:

    python3 profiles/train_timeline.py <kernel_trace.csv> [--steps N]

The trace holds N + 1 identical steps after the engine's set-up; the LAST step is cut out by looking for the
repeating launch sequence (the first `conv_gemm_f32` after the loss kernels of the previous step).  Printed:
the step's span on the GPU clock, the sum of kernel durations, the idle time between kernels (launch gaps —
host-bound stretches show up here), and per-kernel totals, forward and backward separately (the backward
starts at the first `gn_mish_bwd_kernel` / data-gradient launch after the forward's last conv)."""
import collections
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    if name.startswith("dad::conv_gemm_f32<"):
        a = name[len("dad::conv_gemm_f32<"):-1].split(", ")
        return f"conv_gemm<{a[0]},{a[1]},sk{a[2]},kc{a[3]},t{a[4]},s{a[5]}{',res' if a[-1] == 'true' else ''}>"
    if name.startswith("dad::"):
        return name[5:]
    if name.startswith("Cijk"):
        return "hipblaslt gemm"
    if "at::native" in name:
        m = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)", name)
        return "torch " + (m.group(1) if m else "kernel")
    return name[:48]


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
    # steps end with the backward's last weight-gradient work; a new step starts with the first forward conv that
    # follows a wgrad / col_sums kernel by way of torch kernels only
    starts = []
    seen_bwd = True
    for i, (_, _, n) in enumerate(ev):
        if "conv_wgrad" in n or "gn_mish_bwd" in n:
            seen_bwd = True
        elif "conv_gemm_f32" in n and seen_bwd:
            starts.append(i)
            seen_bwd = False
    if len(starts) < 2:
        sys.exit("fewer than two training steps in the trace")
    # the torch kernels in front of a step's first conv (q_sample, time MLPs) belong to it: walk back to the
    # previous step's last dad:: kernel
    def step_begin(i):
        j = i
        while j > 0 and "dad::" not in ev[j - 1][2]:
            j -= 1
        return j
    b = step_begin(starts[-1])
    step = ev[b:]
    prev_b = step_begin(starts[-2])
    print(f"{len(starts)} steps in the trace; last step = {len(step)} kernels (the one before: {b - prev_b})")
    first_bwd = next(i for i, e in enumerate(step) if "gn_mish_bwd" in e[2] or "conv_wgrad" in e[2] or "row_partial" in e[2])
    # the data-gradient of final_conv[1] and torch's loss backward come first: walk back over non-dad kernels and the
    # 1x1 data-gradient conv in front
    last_fwd_conv = max(i for i, e in enumerate(step[:first_bwd]) if "conv_gemm_f32" in e[2] and ", 1, 1, " not in e[2])
    parts = {"forward": step[:last_fwd_conv + 1], "loss + backward": step[last_fwd_conv + 1:]}
    for tag, evs in parts.items():
        span = (evs[-1][1] - evs[0][0]) / 1e3
        busy = sum(e - s for s, e, _ in evs) / 1e3
        gaps = sum(max(0, evs[i + 1][0] - evs[i][1]) for i in range(len(evs) - 1)) / 1e3
        big = sorted(((evs[i + 1][0] - evs[i][1]) / 1e3, short(evs[i][2]), short(evs[i + 1][2])) for i in range(len(evs) - 1))[-5:]
        print(f"\n== {tag}: {len(evs)} kernels, span {span:.0f} us, kernels {busy:.0f} us, idle between kernels {gaps:.0f} us")
        print("   largest gaps: " + "; ".join(f"{g:.0f} us after {a}" for g, a, _ in reversed(big)))
        tot = collections.defaultdict(lambda: [0, 0.0])
        for s, e, n in evs:
            k = short(n)
            tot[k][0] += 1
            tot[k][1] += (e - s) / 1e3
        print("   | kernel | calls | us total | us each |")
        print("   |---|---|---|---|")
        for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
            print(f"   | {k} | {c} | {t:.1f} | {t / c:.1f} |")


if __name__ == "__main__":
    main()
