#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash profiles/collect.sh r03'
# Pass 1   kernel trace + stats of the bench command (timings the JSON line must agree with).
# Pass 2-4 counter passes, one counter family each (FETCH_SIZE and WRITE_SIZE do not fit one pass;
#          gpurun refuses --pmc together with runtime traces), on profiles/pmc_target.py: ONE eager
#          sampling loop and nothing else — under --pmc every dispatch is serialised (>100 us each).
# Evidence is never overwritten: every attempt writes <pass>_<HHMMSS>/ and <pass>_<HHMMSS>.log.
# A pass that hits its timeout leaves its partial CSVs and log in place, gets a <pass>_<HHMMSS>.TIMEOUT
# note (log tail + file listing), and ENDS the script: nothing further is started on a GPU a killed
# profiler may have left in an unknown state, and there is no re-run knob.
set -uo pipefail
tag="${1:-r03}"
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/profiles_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
stamp="$(date +%m%d%H%M%S)"

run_pass() {   # name, limit seconds, command...
  local name="$1" limit="$2"; shift 2
  local dir="$out/${name}_$stamp" log="$out/${name}_$stamp.log"
  echo "[collect] $name -> $dir"
  timeout -k 10 "$limit" "$@" > "$log" 2>&1
  local rc=$?
  if [ $rc -ne 0 ]; then
    { echo "rc=$rc after limit ${limit}s: $*"; echo "--- log tail"; tail -40 "$log"; echo "--- files"; find "$dir" -type f -printf '%s %p\n' 2>/dev/null; } > "$out/${name}_$stamp.TIMEOUT"
    echo "[collect] $name FAILED rc=$rc (kept $log and $dir); stopping"
    python3 "$root/profiles/summarize.py" "$out" "$tag" || true
    exit $rc
  fi
}

bench=(python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-configs)
target=(python3 "$root/profiles/pmc_target.py")
run_pass trace 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$stamp" -- "${bench[@]}"
run_pass pmc_fetch 180 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch_$stamp" -- "${target[@]}"
run_pass pmc_write 180 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write_$stamp" -- "${target[@]}"
run_pass pmc_sq 180 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$out/pmc_sq_$stamp" -- "${target[@]}"
# the opt-in split-f16 arithmetic: its own trace and traffic passes
run_pass x3_trace 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/x3_trace_$stamp" -- "${bench[@]}" --precision f16x3
run_pass x3_fetch 180 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/x3_fetch_$stamp" -- "${target[@]}" --precision f16x3
run_pass x3_write 180 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/x3_write_$stamp" -- "${target[@]}" --precision f16x3
# the T=1000 configurations (BASELINE configs 4 and 5) and the batch-1 planning calls (PointMaze and the two
# wide nets): kernel trace + HBM traffic
# of a 20-step loop each (per-launch averages do not need the thousand steps)
for cfg in "halfcheetah 128" "door 128" "pointmaze 1" "halfcheetah 1" "door 1"; do
  set -- $cfg
  run_pass "trace_$1_b$2" 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$1_b$2_$stamp" -- "${target[@]}" --arch "$1" --batch "$2" --denoise-steps 20
  run_pass "fetch_$1_b$2" 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch_$1_b$2_$stamp" -- "${target[@]}" --arch "$1" --batch "$2" --denoise-steps 20
  run_pass "write_$1_b$2" 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write_$1_b$2_$stamp" -- "${target[@]}" --arch "$1" --batch "$2" --denoise-steps 20
done
# BASELINE config 3: the projected loop (projection kernel time and bytes)
run_pass trace_pointmaze_proj 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_pointmaze_proj_$stamp" -- "${target[@]}" --project
run_pass fetch_pointmaze_proj 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch_pointmaze_proj_$stamp" -- "${target[@]}" --project
run_pass write_pointmaze_proj 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write_pointmaze_proj_$stamp" -- "${target[@]}" --project
# one training step (forward + backward) of the headline architecture
run_pass trace_train 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_train_$stamp" -- python3 "$root/profiles/train_step_timing.py" --arch pointmaze --batch 256 --repeats 3
# per-layer efficiency tables (batch-256 kernels)
for cfg in "pointmaze 256" "halfcheetah 128" "door 128"; do
  set -- $cfg
  t=$(ls "$out"/trace_$1_b$2_$stamp/*/*_kernel_trace.csv 2>/dev/null | head -1)
  [ "$1" = pointmaze ] && t=$(ls "$out"/pmc_sq_$stamp/*/*_kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$t" ] && python3 "$root/profiles/layer_table.py" "$t" "$1" "$2" > "$root/gpurun_out/${tag}_layers_$1_b$2.md" 2> "$out/layers_$1_b$2.err" || true
done
echo "[collect] bench"
timeout -k 10 600 python3 "$root/bench.py" --steps 5 --warmup 2 > "$out/bench_$stamp.json" 2> "$out/bench_$stamp.err"
cp "$out/bench_$stamp.json" "$out/bench.json"
python3 "$root/profiles/summarize.py" "$out" "$tag"
