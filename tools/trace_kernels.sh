#!/bin/bash
# Per-kernel times of one render_configs.py run: rocprofv3 --kernel-trace --stats, then the top rows of the stats table.
# usage: tools/trace_kernels.sh <outdir under gpurun_out> <render_configs.py arguments ...>
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -- \
  python3 $ROOT/tools/render_configs.py "$@" > "$ROOT/$OUT/trace.log" 2>&1 || { tail -5 "$ROOT/$OUT/trace.log"; exit 1; }
F=$(ls $ROOT/$OUT/trace/*/*kernel_stats.csv | head -1)
cp $F $ROOT/$OUT/kernel_stats.csv
python3 - $F <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-60s calls %6s total_ms %10.3f avg_us %10.1f  %5s%%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
