#!/usr/bin/env python3
"""Per-bounce kernel times of one frame from a rocprofv3 kernel trace.
usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
                 python3 tools/bounce_times.py DIR <max_depth>
Prints, for the last frame in the trace, each k_bounce launch's duration."""
import csv
import glob
import json
import sys


def main():
    d, depth = sys.argv[1], int(sys.argv[2])
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "k_bounce" in r["Kernel_Name"]:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort()
    last = rows[-depth:]
    t = [(e - s) / 1e6 for s, e in last]
    gaps = [(last[i + 1][0] - last[i][1]) / 1e6 for i in range(len(last) - 1)]
    print(json.dumps({"launches": len(rows), "frame_ms": round((last[-1][1] - last[0][0]) / 1e6, 3),
                      "kernel_ms": [round(x, 3) for x in t], "sum_kernel_ms": round(sum(t), 3),
                      "sum_gap_ms": round(sum(gaps), 3)}))


if __name__ == "__main__":
    main()
