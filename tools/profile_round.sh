#!/bin/bash
# rocprofv3 passes of a bench command for profiles/: kernel-trace stats, then (PMC=1) hardware counters in SEPARATE passes
# (never combined with other trace domains) of the same workload's timed steps only.  The program comes directly after `--`.
# Run on the GPU box from the repo root:
#   bash tools/profile_round.sh r06_bench                          # default bench command (f32s headline), stats + PMC
#   PMC=0 bash tools/profile_round.sh r06_bench_beam3 --beams 3 --batch 64
#   bash tools/profile_round.sh r06_bench_bf16 --dtype bf16
#   POOLED=1 PMC=0 bash tools/profile_round.sh r06_bench               # + the pooled trace (3 streams, merged passes): stats and timeline
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06_bench}; shift || true
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# one stream: with the default stream pool kernels of different batches co-run and share the GPU, so their durations in the
# trace are not those of the kernel alone (vit_attention_split: 963 us pooled, 228 us alone) - the bench's roofline pass, which
# these averages must agree with, also runs one engine on one stream
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ROOT/bench.py --steps 3 --warmup 1 --streams 1 --no-extra-modes --no-cpu-baseline --no-latency --no-strong "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
echo "stats pass done"
cd "$ROOT"
python3 tools/summarize_prof.py stats "$OUT/stats" "$OUT/kernel_stats.md"
if [ "${PMC:-1}" = "1" ]; then
  cd /tmp
  i=0
  for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/pmc/p$i" -- python3 $ROOT/bench.py --steps 1 --warmup 1 --lite --streams 1 "$@" > "$OUT/pmc_p$i.json" 2> "$OUT/pmc_p$i.err"
    echo "pmc pass $i ($ctr) done"
  done
  cd "$ROOT"
  python3 tools/summarize_prof.py pmc "$OUT/pmc" "$OUT/pmc.json"
fi
if [ "${POOLED:-0}" = "1" ]; then
  # the headline's own execution shape: 3 engines / streams, the pool merging steps into 768-1024-row passes.  Durations in this
  # trace include the wait for CUs held by kernels of the other streams (a kernel's start stamp is its dispatch)
  cd /tmp
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pooled" -- python3 $ROOT/bench.py --steps 20 --warmup 3 --lite "$@" > "$OUT/pooled_bench.json" 2> "$OUT/pooled.err"
  echo "pooled pass done"
  cd "$ROOT"
  python3 tools/summarize_prof.py stats "$OUT/pooled" "$OUT/pooled_kernel_stats.md"
  python3 tools/pool_timeline.py pack "$OUT/pooled" "$OUT/pooled_trace.pkl.gz"
  python3 tools/pool_timeline.py report "$OUT/pooled_trace.pkl.gz" > "$OUT/pool_timeline.txt"
fi
rm -rf "$OUT/stats" "$OUT/pmc" "$OUT/pooled"      # keep only the summaries (the raw csv files are large)
