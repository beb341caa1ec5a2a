#!/bin/bash
# LDS bank-conflict counters of the encoder GEMM and the ViT attention kernel (GPU box; rocprofv3 --pmc in its own pass, the program
# directly after `--`).   bash tools/pmc_lds.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_lds
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/p" -- python3 $ROOT/bench.py --steps 1 --warmup 1 --lite --streams 1 > "$OUT/run.json" 2> "$OUT/run.err"
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in glob.glob(out + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:12]:
    a = v.get("SQ_LDS_IDX_ACTIVE", 0.0); c = v.get("SQ_LDS_BANK_CONFLICT", 0.0)
    print(f"{k:70s} lds_active {a:14.0f} bank_conflict {c:14.0f} ratio {c / a if a else 0:.3f} insts {v.get('SQ_INSTS_LDS', 0):12.0f}")
PY
