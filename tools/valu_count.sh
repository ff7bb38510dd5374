#!/bin/bash
# Wave-level VALU instructions of the bench frame's bounce kernels for library variants (cost probes):
# one --pmc pass per variant.  usage: tools/valu_count.sh tag1 tag2 ...   (variants/librt355_<tag>.so)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for T in "$@"; do
  export RT355_LIB=$ROOT/variants/librt355_$T.so
  rm -rf /tmp/valu_$T
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d /tmp/valu_$T -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/valu_$T.log 2>&1 || { tail -3 /tmp/valu_$T.log; exit 1; }
  python3 - "$T" /tmp/valu_$T <<'PY'
import csv, glob, sys, collections
tag, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_bounce<false" in k or "k_bounce_loop<false" in k:
        agg[("loop " if "loop" in k else "bounce ") + r["Counter_Name"]] += float(r["Counter_Value"])
print(tag, {k: "%.4g" % v for k, v in sorted(agg.items())})
PY
done
