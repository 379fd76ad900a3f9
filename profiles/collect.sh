#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash profiles/collect.sh r01'
# Pass 1: kernel trace + stats of the default bench command (timings the JSON line must agree with)
# Pass 2/3: PMC passes, one counter family each (FETCH_SIZE and WRITE_SIZE do not fit one pass,
#           and gpurun refuses --pmc together with runtime traces).  They profile ONE loop (per-launch
#           averages do not need more) under a short timeout: rocprofv3 --pmc FETCH_SIZE has hung
#           intermittently on this pool; ONLY_PMC=1 re-runs just these two passes.
# Every pass prints a progress line first so the run is never silent.
set -uo pipefail
tag="${1:-r01}"
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/profiles_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cmd=(python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-alt)
pmc_cmd=(python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-alt)
if [ "${ONLY_PMC:-0}" != "1" ]; then
  echo "[collect] kernel trace"
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- "${cmd[@]}" > "$out/trace.log" 2>&1
fi
rm -rf "$out/pmc_fetch" "$out/pmc_write"
echo "[collect] pmc FETCH_SIZE"
timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- "${pmc_cmd[@]}" > "$out/pmc_fetch.log" 2>&1
echo "[collect] pmc WRITE_SIZE"
timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- "${pmc_cmd[@]}" > "$out/pmc_write.log" 2>&1
# the opt-in split-f16 arithmetic: its own trace and traffic passes
x3_cmd=(python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-alt --precision f16x3)
x3_pmc=(python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-alt --precision f16x3)
rm -rf "$out/x3_fetch" "$out/x3_write"
if [ "${ONLY_PMC:-0}" != "1" ]; then
  echo "[collect] kernel trace (f16x3)"
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/x3_trace" -- "${x3_cmd[@]}" > "$out/x3_trace.log" 2>&1
fi
echo "[collect] pmc FETCH_SIZE (f16x3)"
timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/x3_fetch" -- "${x3_pmc[@]}" > "$out/x3_fetch.log" 2>&1
echo "[collect] pmc WRITE_SIZE (f16x3)"
timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/x3_write" -- "${x3_pmc[@]}" > "$out/x3_write.log" 2>&1
if [ "${ONLY_PMC:-0}" != "1" ]; then
  echo "[collect] pmc SQ"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$out/pmc_sq" -- "${cmd[@]}" > "$out/pmc_sq.log" 2>&1
  echo "[collect] bench"
  timeout -k 10 400 python3 "$root/bench.py" --steps 5 --warmup 2 > "$out/bench.json" 2> "$out/bench.err"
fi
python3 "$root/profiles/summarize.py" "$out" "$tag"
