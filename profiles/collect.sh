#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash profiles/collect.sh r01'
# Pass 1: kernel trace + stats of the default bench command (timings the JSON line must agree with)
# Pass 2/3: PMC passes, one counter family each (FETCH_SIZE and WRITE_SIZE do not fit one pass,
#           and gpurun refuses --pmc together with runtime traces).
set -uo pipefail
tag="${1:-r01}"
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/profiles_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cmd=(python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline)
echo "[collect] kernel trace"; timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- "${cmd[@]}" > "$out/trace.log" 2>&1
echo "[collect] pmc FETCH_SIZE"; timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- "${cmd[@]}" > "$out/pmc_fetch.log" 2>&1
echo "[collect] pmc WRITE_SIZE"; timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- "${cmd[@]}" > "$out/pmc_write.log" 2>&1
echo "[collect] pmc SQ"; timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$out/pmc_sq" -- "${cmd[@]}" > "$out/pmc_sq.log" 2>&1
echo "[collect] bench"; timeout -k 10 400 python3 "$root/bench.py" --steps 5 --warmup 2 > "$out/bench.json" 2> "$out/bench.err"
python3 "$root/profiles/summarize.py" "$out" "$tag"
