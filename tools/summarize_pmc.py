#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per kernel and derives the ratios quoted in the
design notes.  usage: summarize_pmc.py <dir> [kernel substrings, separated by semicolons] [--current out.json --spp N]

--current writes the file bench.py reads its roofline.traffic and roofline.valu from: the counters
per frame, stamped with a hash of the kernel sources they were measured on (bench.py ignores the
file when the sources have changed since) and with the samples per pixel of the measured frame.

HBM traffic per launch = (FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024) / dispatches:
on gfx950 FETCH_SIZE tallies the 128-byte requests of a 16-B-per-lane streaming read
at 64 bytes, which is the access pattern of the ray queues; WRITE_SIZE is exact for
16-B-per-lane stores."""
import collections
import csv
import glob
import json
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
root = args[0]
match = args[1] if len(args) > 1 else "k_bounce<false;k_bounce_loop<false"  # substrings (kernel names hold commas)
wanted = match.split(";")
traffic_out = sys.argv[sys.argv.index("--current") + 1] if "--current" in sys.argv else None
spp = int(sys.argv[sys.argv.index("--spp") + 1]) if "--spp" in sys.argv else 64
args = [a for a in args if a != traffic_out and a != str(spp)]
# --raw-fetch: FETCH_SIZE as counted (kernels whose reads are mostly 64-byte gathers, which are tallied in full)
fetch_factor = 1 if "--raw-fetch" in sys.argv else 2
tot = collections.defaultdict(float)
disp = collections.defaultdict(set)
for f in sorted(glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    # Kernels without a counting variant (k_shade) also run in the untimed counting frame that comes first:
    # only dispatches from the first one of the first pattern on belong to the measured frame.
    first = min((int(r["Dispatch_Id"]) for r in rows if wanted[0] in r["Kernel_Name"]), default=0)
    for r in rows:
        if not any(w in r["Kernel_Name"] for w in wanted) or int(r["Dispatch_Id"]) < first:
            continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        disp[r["Counter_Name"]].add(r["Dispatch_Id"])
c = tot
out = {"kernel": match, "dispatches": sorted({len(v) for v in disp.values()}),
       "counters": dict(sorted(tot.items()))}
d = {}
if c.get("SQ_WAVE_CYCLES"):
    d["wait_any_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    d["wait_inst_any_frac"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    d["active_inst_any_frac"] = c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    d["active_inst_valu_frac"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
    d["mean_waves_per_busy_cycle"] = c["SQ_WAVE_CYCLES"] / max(c.get("SQ_BUSY_CYCLES", 1), 1)
if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
    # lanes active per VALU instruction cycle: 1.0 = no divergence
    d["valu_active_lane_frac"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
    d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
if c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
    d["l1_miss_to_l2_frac"] = c.get("TCP_TCC_READ_REQ_sum", 0) / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    n = len(disp["FETCH_SIZE"])
    d["fetch_bytes_raw"] = c["FETCH_SIZE"] * 1024
    d["fetch_bytes_corrected_x2"] = c["FETCH_SIZE"] * 2048
    d["write_bytes"] = c["WRITE_SIZE"] * 1024
    d["hbm_bytes_per_launch"] = (c["FETCH_SIZE"] * 1024 * fetch_factor + c["WRITE_SIZE"] * 1024) / max(n, 1)
    if traffic_out:
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        # Vector-instruction issue peak: tools/valu_peak.hip on the same chip sustains 2.5 cycles per
        # wave64 instruction per SIMD for plain fp32 / integer / logic operations with two or more
        # waves per SIMD (profiles/r02_valu_peak.jsonl: 930-970 G wave-instructions/s at the
        # 2.3 GHz the chip holds under that load); v_max, v_mul_lo_u32 take 4.2, v_rcp 8.2.
        cur = {"source_hash": bench.kernel_source_hash(), "spp": spp, "kernel": match, "launches_per_frame": n,
               "fetch_KiB_raw": c["FETCH_SIZE"], "write_KiB": c["WRITE_SIZE"],
               "correction": ("FETCH_SIZE x2 (gfx950: 128-byte requests of 16-B-per-lane streaming reads are tallied "
                              "at 64 bytes; an upper bound here, part of the reads are 16-byte gathers); WRITE_SIZE as is")
               if fetch_factor == 2 else
                             ("FETCH_SIZE as counted (the reads are mostly gathers of 64-byte records, tallied in full; "
                              "the queues' streaming reads, a few per cent of the requests, are under-counted by half: "
                              "a lower bound); WRITE_SIZE as is"),
               "hbm_bytes_per_launch": int(d["hbm_bytes_per_launch"]),
               "source": "profiles/%s (rocprofv3 --pmc passes of bench.py --steps 1 --warmup 0, tools/profile_round.sh)"
                         % os.path.basename(traffic_out)}
        if c.get("SQ_INSTS_VALU"):
            cur["valu"] = {"wave_insts_per_frame": c["SQ_INSTS_VALU"], "salu_insts_per_frame": c.get("SQ_INSTS_SALU"),
                           "peak_ginst_s": 950.0,
                           "peak_basis": "256 CUs x 4 SIMDs x 2.3 GHz / 2.5 cycles per wave64 instruction, measured with "
                                         "tools/valu_peak.hip (profiles/r02_valu_peak.jsonl)",
                           "active_lane_frac": round(d.get("valu_active_lane_frac", 0.0), 4),
                           "wait_any_frac": round(d.get("wait_any_frac", 0.0), 4)}
        if c.get("TCC_EA0_RDREQ_sum"):
            # 64-byte read requests that leave L2 (to the Infinity Cache and HBM): what a traversal of a tree
            # in global memory is limited by
            cur["l2_requests"] = {"ea_read_requests_per_frame": c["TCC_EA0_RDREQ_sum"],
                                  "l2_hit_rate": round(d.get("l2_hit_rate", 0.0), 4)}
        json.dump(cur, open(traffic_out, "w"), indent=1)
out["derived"] = d
print(json.dumps(out, indent=1))
