#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and ms/frame of the Next-Week scene
(test/week.toml of the reference) at 1920x1080, 64 spp, depth 50.

One step = one frame through the HIP wavefront path tracer (rt_render_dev):
generate -> [closest hit -> shade -> compact] x depth -> resolve.  The scene
(BVH + primitive tables + textures) is built on the host and uploaded to HBM
before the timed region; camera rays are generated on the device, so nothing
crosses PCIe inside it.

With N > 1 (launched by torch.distributed.run, one process per GPU) the frame's
32x32 tiles are dealt round-robin to the ranks, every rank renders its tiles
into a zeroed full-size framebuffer and one RCCL reduce(sum) to rank 0 ends the
frame: total work is fixed, so scaling is "strong".

Extra objects on the JSON line:
  roofline     -- the bounce kernels (k_bounce: one launch per bounce while the
                  queue is large; k_bounce_loop: one launch for all later bounces;
                  trees in global memory: k_trace + k_shade per bounce).
                  `bound` names the resource that limits them and `achieved / peak /
                  frac` are of THAT resource:
                    week scene (tree in LDS): "valu" -- wave-level vector instructions
                      per second (count from a rocprofv3 --pmc run of these very kernel
                      sources, time from HIP events recorded on the launch stream inside
                      the timed steps) against the issue rate tools/valu_peak.hip
                      measured on this chip; `frac_vs_guide_peak` is the same against
                      2 cycles per wave64 instruction;
                    --config C5 (tree in global memory; a bounce is k_trace + k_shade):
                      "valu" as well since the trace walks 128-byte four-box records;
                      `l2` holds the 64-byte requests that leave L2 per second against the
                      rate tools/gather_peak.hip measured for dependent random reads of
                      128-byte records from a table of the tree's size.
                  `algorithmic` is the reading the metric's definition asks for:
                  algorithmic bytes of all ray segments of a frame (traversal counters
                  of an untimed counting frame; identical to the oracle's) over the
                  summed duration of the frame's bounce launches, against the 8 TB/s
                  HBM peak.  `traffic` is HBM bytes per launch from the counters.
                  Counter-based fields are null when the counter file was not measured
                  on the kernel sources of this tree (a hash is stored with it).
  cpu_baseline -- the CPU oracle (a port of the reference algorithm, not the
                  product) built -O3 -march=native -ffp-contract=off on the machine
                  and timed on its host cores on a bounded sample, before the GPU
                  phase.
  host         -- CPU model, core count and the threads the CPU leg used.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# name of the scene builder, frame, label of the workload
CONFIGS = {
    "C3": dict(scene="week", width=1920, height=1080, spp=64, depth=50, label="week.toml Next-Week scene"),
    "C5": dict(scene="synthetic", width=3840, height=2160, spp=128, depth=8,
               label="synthetic scene (1 M-triangle terrain + 10 M spheres)"),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
# algorithmic bytes: ray read + hit/next-ray write + 32 B per node visited + record per primitive tested
RAY_BYTES, OUT_BYTES, NODE_BYTES = 32, 32, 32
PRIM_BYTES = [16, 32, 36, 24, 24, 124, 32, 0]  # sphere, moving, triangle, rect, box, gsphere, medium
# Vector-instruction issue: 256 CUs x 4 SIMDs; the guide's 2 cycles per wave64 instruction at 2.4 GHz
GUIDE_ISSUE_PEAK_GINST_S = 256 * 4 * 2.4 / 2.0
# Dependent random reads of 128-byte records from a table of the size of the C5 tree's record tables
# (0.7 GB of two-box and 0.7 GB of four-box records), 6 waves per SIMD: 65 G records/s = 130 G 64-byte
# requests/s (64-byte records: 57 G/s).  tools/gather_peak.hip, profiles/r03_gather_peak.jsonl
GATHER_PEAK_GREQ_S = 130.0


def algorithmic_bytes(stats):
    b = stats.segments * (RAY_BYTES + OUT_BYTES) + stats.nodes_visited * NODE_BYTES
    for k in range(8):
        b += stats.prims_tested[k] * PRIM_BYTES[k]
    return int(b)


def kernel_source_hash():
    """Identifies the kernels a counter file was measured on (there is no git on the GPU box)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "raytracer_amd", "csrc")
    for name in ("rt_kernels.hip", "rt_kernels.hpp", "rt_device.hpp", "rt_api.cpp"):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def host_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}


def native_oracle():
    """Builds the oracle for this machine (-O3 -march=native) into a temporary directory and makes
    tests/orc.py load it; the portable in-tree build is the fallback.  Returns the flags used."""
    import subprocess
    import tempfile
    out = os.path.join(tempfile.mkdtemp(prefix="orc_native_"), "liborc_native.so")
    try:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "NATIVE_OUT=" + out,
                               "liborc_native.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.environ["ORC_LIB"] = out
        return "gcc -O3 -march=native -ffp-contract=off"
    except (OSError, subprocess.CalledProcessError):
        return "gcc -O3 -ffp-contract=off (portable build; native build failed)"


def cpu_baseline(frozen, cam, cfg, threads, target_s=15.0):
    """Oracle (port of the reference path) on the host cores, bounded sample:
    the same frame at fewer samples per pixel."""
    flags = native_oracle()
    from tests import orc
    w, h, depth = cfg["width"], cfg["height"], cfg["depth"]
    o = orc.OracleScene(frozen)
    t0 = time.time()
    _, st = o.render(cam, orc.render_opts(w, h, 1, depth, seed=0), threads=threads)
    t1 = time.time() - t0
    spp = max(1, min(cfg["spp"], int(target_s / max(t1, 1e-3))))
    if spp > 1:
        t0 = time.time()
        _, st = o.render(cam, orc.render_opts(w, h, spp, depth, seed=0), threads=threads)
        t1 = time.time() - t0
    return {"value": round(st.segments / t1 / 1e6, 3), "unit": "Mrays/s", "cores": threads,
            "kind": "port", "build": flags,
            "sample": "%s %dx%d, %d of %d spp, depth %d (%.1f s)" % (cfg["scene"], w, h, spp, cfg["spp"], depth, t1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=0, help="debug only; the headline is 64 (C5: 128)")
    ap.add_argument("--config", default="C3", choices=["C3", "C5"],
                    help="C3: the headline (week.toml 1080p x 64 spp).  C5: the large synthetic scene at "
                         "4K x 128 spp, depth 8 (set-up takes a minute; no CPU leg unless --cpu-threads is given)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--backend", default="nccl", help="rehearsal only: gloo lets several ranks share one GPU")
    ap.add_argument("--one-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--checksum", action="store_true", help="print a checksum of the final frame to stderr")
    ap.add_argument("--no-wait", action="store_true",
                    help="nccl backend only: frames are enqueued without statistics and not waited for, so that "
                         "the frame's reduce follows its kernels on the stream without a host round trip")
    ap.add_argument("--single-rank-collective", action="store_true",
                    help="rehearsal only: a world of one still joins the process group and runs the frame's "
                         "reduce, the barriers and the all-reduces through the backend (RCCL with one rank)")
    args = ap.parse_args()
    if os.environ.get("RT355_BENCH_WATCHDOG"):  # seconds after which every thread's stack is printed and the run ends
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["RT355_BENCH_WATCHDOG"]), exit=True)

    import torch
    import torch.distributed as dist

    from raytracer_amd import abi, lib, scenes
    from raytracer_amd.distributed import env_rank, render_frame

    rank, world, local_rank = env_rank()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "--nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.single_rank_collective
    if use_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    cfg = CONFIGS[args.config]
    WIDTH, HEIGHT, DEPTH = cfg["width"], cfg["height"], cfg["depth"]
    if not args.spp:
        args.spp = cfg["spp"]
    sd, camf = scenes.BUILDERS[cfg["scene"]]()
    frozen = sd.freeze()
    cam = camf(WIDTH, HEIGHT)
    # CPU leg first (rank 0 of a single-GPU run only): the GPU phase then runs in one piece
    cpu = None
    if world == 1 and not args.no_cpu_baseline and (args.config == "C3" or args.cpu_threads):
        threads = args.cpu_threads or int(os.environ.get("OMP_NUM_THREADS", "0")) or len(os.sched_getaffinity(0))
        cpu = cpu_baseline(frozen, cam, cfg, threads)
    scene = lib.Scene(frozen, 4, abi.RT_SPLIT_SAH, device=local_rank)
    scene.upload()
    info = scene.info()
    fb = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def frame(flags=0, want_stats=True):
        def shard(index, count, buf):
            opts = lib.render_opts(WIDTH, HEIGHT, args.spp, DEPTH, seed=0, shard=(index, count), flags=flags)
            return scene.render_dev(cam, opts, buf.data_ptr(), stream, want_stats=want_stats)

        return render_frame(shard, fb, rank, world, dist, single_rank_collective=args.single_rank_collective)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    counted = frame(abi.RT_RENDER_COUNTERS)  # untimed: traversal counters of this rank's shard
    if args.verbose and rank == 0:
        print("bounce_rays", list(counted.bounce_rays[:DEPTH]), file=sys.stderr)
        print("nodes_visited", counted.nodes_visited, "prims_tested", list(counted.prims_tested),
              "segments", counted.segments, file=sys.stderr)
    for _ in range(args.warmup):
        frame()
    fence()
    # Every timed frame carries HIP events around its bounce launches (what the roofline's kernel
    # time is made of) and is waited for before its reduce is enqueued.  --no-wait (ranks joined by
    # RCCL only): a rank enqueues frame and reduce on its stream without a host wait in between --
    # its segment count is that of the counting frame, the scene and the seed being the same -- and
    # one more frame after the timed region takes the kernel times.  It saves one host round trip
    # per frame (tens of microseconds) and has only been run with one rank, no node with several
    # GPUs having been at hand; with gloo, whose reduce stages through the host, two ranks sharing
    # one GPU stalled in it, so there it is refused.
    if args.no_wait and not (use_dist and args.backend == "nccl"):
        sys.exit("--no-wait needs ranks joined by the nccl backend")
    wait_frames = not args.no_wait
    t0 = time.perf_counter()
    segments = 0
    trace_ms = 0.0
    launches = 0
    for _ in range(args.steps):
        if wait_frames:
            st = frame(abi.RT_RENDER_TIMING)
            trace_ms += st.trace_ms
            launches += st.trace_launches
            segments += st.segments
        else:
            frame(want_stats=False)
            segments += counted.segments
    fence()
    elapsed = time.perf_counter() - t0
    timed_frames = args.steps
    if not wait_frames:
        st = frame(abi.RT_RENDER_TIMING)
        trace_ms, launches, timed_frames = st.trace_ms, st.trace_launches, 1
        fence()

    tot = torch.tensor([elapsed, float(segments)], dtype=torch.float64, device="cuda")
    if use_dist:
        t_max = tot[:1].clone()
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        seg_sum = tot[1:].clone()
        dist.all_reduce(seg_sum, op=dist.ReduceOp.SUM)
        elapsed, segments = float(t_max.item()), float(seg_sum.item())

    if rank == 0 and args.checksum:
        import hashlib
        print("frame sha256", hashlib.sha256(fb.cpu().numpy().tobytes()).hexdigest(), file=sys.stderr)
    if rank == 0:
        alg_bytes = algorithmic_bytes(counted)  # per frame, this rank's shard
        frame_trace_s = trace_ms / max(timed_frames, 1) * 1e-3  # bounce launches of one frame
        per_launch_ms = trace_ms / max(launches, 1)
        launches_per_frame = launches / max(timed_frames, 1)
        alg_gbs = alg_bytes / frame_trace_s / 1e9 if trace_ms > 0 else 0.0
        lds_tree = args.config == "C3"
        roofline = {
            "kernel": "k_bounce + k_bounce_loop" if lds_tree else "k_trace + k_shade (one pair per bounce)",
            "bound": "valu",
            "achieved": None, "peak": None, "unit": "Ginst/s", "frac": None,
            "traffic": None,
            "avg_launch_ms": round(per_launch_ms, 4), "launches_per_frame": launches_per_frame,
            # the reading the metric's definition asks for
            "algorithmic": {"achieved": round(alg_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(alg_gbs / HBM_PEAK_GBS, 4),
                            "bytes_per_launch": int(alg_bytes / max(launches_per_frame, 1)),
                            "bytes_per_segment": round(alg_bytes / max(counted.segments, 1), 1)},
            "note": ("the interior records of this tree are staged in LDS by every workgroup, so the algorithmic "
                     "bytes (32 B per node visited + ray, result and primitive records) never reach HBM; the "
                     "counters show vector-instruction issue as the limit, and achieved / peak / frac are wave-level "
                     "vector instructions per second against the rate measured on this chip (tools/valu_peak.hip)")
            if lds_tree else
                    ("a bounce is a trace launch (k_trace) and a shade launch (k_shade); avg_launch_ms is the pair's.  "
                     "The tree is far larger than the caches.  From the third bounce on the trace walks 128-byte "
                     "records with four boxes, which the memory system delivers as fast as 64-byte ones, and the "
                     "counters then show vector-instruction issue ahead of memory requests as the resource nearest "
                     "its limit: achieved / peak / frac are wave-level vector instructions per second against the "
                     "rate measured on this chip (tools/valu_peak.hip); `l2` has the 64-byte requests that leave L2 "
                     "per second against the rate measured for dependent random reads of 128-byte records "
                     "(tools/gather_peak.hip).  `algorithmic` counts every node visit of the reference's walk"),
        }
        out = {
            "metric": "Mrays/s", "value": round(segments / elapsed / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d %d spp depth %d, %d primitives, %d BVH nodes (SAH)" % (
                           cfg["label"], WIDTH, HEIGHT, args.spp, DEPTH, info.n_prims, info.n_nodes),
                       "sharding": "32x32 tiles round-robin over ranks, one reduce(sum) per frame",
                       "ray_segments_per_frame": int(segments / args.steps)},
            "roofline": roofline,
            "host": host_info(),
        }
        # hardware counters of a rocprofv3 run on these very kernel sources (tools/profile_round.sh)
        pmc_file = os.path.join(ROOT, "profiles", "pmc_current.json" if lds_tree else "pmc_current_c5.json")
        if world == 1 and os.path.exists(pmc_file) and trace_ms > 0:
            with open(pmc_file) as f:
                pmc = json.load(f)
            if pmc.get("source_hash") == kernel_source_hash() and pmc.get("spp") == args.spp:
                roofline["traffic"] = pmc.get("hbm_bytes_per_launch")
                v = pmc.get("valu")
                if v:
                    seg = float(pmc.get("segments_per_frame", segments / args.steps))
                    rate = v["wave_insts_per_frame"] / frame_trace_s / 1e9
                    roofline["valu"] = {
                        "wave_insts_per_segment": round(v["wave_insts_per_frame"] / seg, 2),
                        "achieved_ginst_s": round(rate, 1), "peak_ginst_s": v["peak_ginst_s"],
                        "frac": round(rate / v["peak_ginst_s"], 4),
                        "frac_vs_guide_peak": round(rate / GUIDE_ISSUE_PEAK_GINST_S, 4),
                        "guide_peak_ginst_s": round(GUIDE_ISSUE_PEAK_GINST_S, 1),
                        "active_lane_frac": v.get("active_lane_frac"), "peak_basis": v.get("peak_basis"),
                        "source": pmc.get("source", "profiles/" + os.path.basename(pmc_file))}
                    roofline.update(achieved=round(rate, 1), peak=v["peak_ginst_s"],
                                    frac=round(rate / v["peak_ginst_s"], 4))
                g = pmc.get("l2_requests")
                if g and not lds_tree:
                    rate = g["ea_read_requests_per_frame"] / frame_trace_s / 1e9
                    roofline["l2"] = {"hit_rate": g.get("l2_hit_rate"),
                                      "requests_beyond_l2_per_segment": round(g["ea_read_requests_per_frame"] / max(counted.segments, 1), 2),
                                      "achieved_grequests_s": round(rate, 2), "peak_grequests_s": GATHER_PEAK_GREQ_S,
                                      "frac": round(rate / GATHER_PEAK_GREQ_S, 4),
                                      "peak_basis": "dependent random reads of 128-byte records (two 64-byte requests each) from "
                                                    "a 1.4 GB table, 6 waves per SIMD (tools/gather_peak.hip, "
                                                    "profiles/r03_gather_peak.jsonl)"}
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["host"]["cpu_threads_used"] = cpu["cores"]
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
