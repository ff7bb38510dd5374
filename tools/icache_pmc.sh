#!/bin/bash
# Instruction-cache counters of the bench frame (one --pmc pass).  usage: tools/icache_pmc.sh <outdir> [bench args]
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $ROOT/$OUT/icache -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $ROOT/$OUT/icache.log 2>&1 || { tail -5 $ROOT/$OUT/icache.log; exit 1; }
python3 - $ROOT/$OUT/icache <<'PY'
import csv, glob, sys, collections, json
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    name = "k_bounce_loop" if "k_bounce_loop<" in k else ("k_bounce" if "k_bounce<" in k else None)
    if name:
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
for name, c in agg.items():
    c = dict(c)
    if c.get("SQC_ICACHE_REQ"):
        c["icache_miss_rate"] = c.get("SQC_ICACHE_MISSES", 0) / c["SQC_ICACHE_REQ"]
    if c.get("SQ_WAVE_CYCLES"):
        c["ifetch_level_per_wave_cycle"] = c.get("SQ_IFETCH_LEVEL", 0) / c["SQ_WAVE_CYCLES"]
    print(json.dumps({"kernel": name, **c}))
PY
