#!/bin/bash
# rocprofv3 passes of the bench command for profiles/: kernel-trace stats of the default bench command, then PMC
# counters in separate passes (never combined with other trace domains) of the same workload's timed steps only.
# Run on the GPU box from the repo root: bash tools/profile_round.sh [tag]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${1:-r01b}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-strict --no-cpu-baseline ${BENCH_EXTRA:-} > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
echo "stats pass done"
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/pmc/p$i" -- python3 $ROOT/bench.py --steps 1 --warmup 1 --lite > "$OUT/pmc_p$i.json" 2> "$OUT/pmc_p$i.err"
  echo "pmc pass $i ($ctr) done"
done
cd "$ROOT"
python3 tools/summarize_prof.py stats "$OUT/stats" "$OUT/kernel_stats.md"
python3 tools/summarize_prof.py pmc "$OUT/pmc" "$OUT/pmc.json"
rm -rf "$OUT/stats" "$OUT/pmc"      # keep only the summaries (the raw csv files are large)
