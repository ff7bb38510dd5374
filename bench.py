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
                  queue is large; k_bounce_loop: one launch for all later bounces):
                  algorithmic bytes of all ray segments of a frame (traversal
                  counters of an untimed counting frame; identical to the oracle's)
                  over the summed duration of the frame's bounce launches, measured
                  with HIP events recorded on the launch stream inside the timed
                  steps.  On this scene the tree is read from LDS, so those bytes never
                  reach HBM and the kernel's real limit is vector-instruction issue:
                  `bound` says so, and `traffic` (HBM bytes per launch) and `valu`
                  (instructions per segment, issue rate against the chip's, active
                  lanes) carry the hardware counters of a separate rocprofv3 run --
                  only when that run was made with the kernel sources of this tree
                  (a hash of them is stored with the counters), null otherwise.
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

WIDTH, HEIGHT, SPP, DEPTH = 1920, 1080, 64, 50
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
# algorithmic bytes: ray read + hit/next-ray write + 32 B per node visited + record per primitive tested
RAY_BYTES, OUT_BYTES, NODE_BYTES = 32, 32, 32
PRIM_BYTES = [16, 32, 36, 24, 24, 124, 32, 0]  # sphere, moving, triangle, rect, box, gsphere, medium


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


def cpu_baseline(frozen, cam, threads, target_s=15.0):
    """Oracle (port of the reference path) on the host cores, bounded sample:
    the same 1920x1080 frame at fewer samples per pixel."""
    flags = native_oracle()
    from tests import orc
    o = orc.OracleScene(frozen)
    t0 = time.time()
    _, st = o.render(cam, orc.render_opts(WIDTH, HEIGHT, 1, DEPTH, seed=0), threads=threads)
    t1 = time.time() - t0
    spp = max(1, min(SPP, int(target_s / max(t1, 1e-3))))
    t0 = time.time()
    _, st = o.render(cam, orc.render_opts(WIDTH, HEIGHT, spp, DEPTH, seed=0), threads=threads)
    dt = time.time() - t0
    return {"value": round(st.segments / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads,
            "kind": "port", "build": flags,
            "sample": "week scene %dx%d, %d of %d spp, depth %d (%.1f s)" % (WIDTH, HEIGHT, spp, SPP, DEPTH, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP, help="debug only; the headline is 64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--backend", default="nccl", help="rehearsal only: gloo lets several ranks share one GPU")
    ap.add_argument("--one-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--checksum", action="store_true", help="print a checksum of the final frame to stderr")
    ap.add_argument("--single-rank-collective", action="store_true",
                    help="rehearsal only: a world of one still joins the process group and runs the frame's "
                         "reduce, the barriers and the all-reduces through the backend (RCCL with one rank)")
    args = ap.parse_args()

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

    sd, camf = scenes.next_week()
    frozen = sd.freeze()
    cam = camf(WIDTH, HEIGHT)
    # CPU leg first (rank 0 of a single-GPU run only): the GPU phase then runs in one piece
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        threads = args.cpu_threads or int(os.environ.get("OMP_NUM_THREADS", "0")) or len(os.sched_getaffinity(0))
        cpu = cpu_baseline(frozen, cam, threads)
    scene = lib.Scene(frozen, 4, abi.RT_SPLIT_SAH, device=local_rank)
    scene.upload()
    info = scene.info()
    fb = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def frame(flags=0):
        def shard(index, count, buf):
            opts = lib.render_opts(WIDTH, HEIGHT, args.spp, DEPTH, seed=0, shard=(index, count), flags=flags)
            return scene.render_dev(cam, opts, buf.data_ptr(), stream)

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
    t0 = time.perf_counter()
    segments = 0
    trace_ms = 0.0
    launches = 0
    for _ in range(args.steps):
        st = frame(abi.RT_RENDER_TIMING)
        segments += st.segments
        trace_ms += st.trace_ms
        launches += st.trace_launches
    fence()
    elapsed = time.perf_counter() - t0

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
        per_launch_ms = trace_ms / max(launches, 1)
        launches_per_frame = launches / max(args.steps, 1)
        achieved = alg_bytes / (trace_ms / args.steps * 1e-3) / 1e9 if trace_ms > 0 else 0.0
        out = {
            "metric": "Mrays/s", "value": round(segments / elapsed / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "week.toml Next-Week scene %dx%d %d spp depth %d, %d primitives, "
                                   "%d BVH nodes (SAH)" % (WIDTH, HEIGHT, args.spp, DEPTH, info.n_prims,
                                                           info.n_nodes),
                       "sharding": "32x32 tiles round-robin over ranks, one reduce(sum) per frame",
                       "ray_segments_per_frame": int(segments / args.steps)},
            "roofline": {"kernel": "k_bounce + k_bounce_loop",
                         # the interior records of this tree are staged in LDS by every workgroup: the
                         # algorithmic bytes are served by LDS, and the counters show vector-instruction
                         # issue, not memory, as the limit (valu, below)
                         "bound": "valu", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": None,
                         "algorithmic_bytes_per_launch": int(alg_bytes / max(launches_per_frame, 1)),
                         "avg_launch_ms": round(per_launch_ms, 4),
                         "launches_per_frame": launches_per_frame,
                         "valu": None,
                         "note": "achieved/peak/frac: algorithmic bytes (32 B per node visited + ray, result and "
                                 "primitive records) over the HBM peak, as the metric is defined; the tree is "
                                 "LDS resident, so this is not HBM traffic (see traffic) and the limit is "
                                 "instruction issue (see valu)"},
            "host": host_info(),
        }
        # hardware counters of a rocprofv3 run on these very kernel sources (tools/profile_round.sh)
        pmc_file = os.path.join(ROOT, "profiles", "pmc_current.json")
        if world == 1 and os.path.exists(pmc_file):
            with open(pmc_file) as f:
                pmc = json.load(f)
            if pmc.get("source_hash") == kernel_source_hash() and pmc.get("spp") == args.spp:
                out["roofline"]["traffic"] = pmc.get("hbm_bytes_per_launch")
                v = pmc.get("valu")
                if v:
                    seg = float(pmc.get("segments_per_frame", segments / args.steps))
                    rate = v["wave_insts_per_frame"] / (trace_ms / args.steps * 1e-3)
                    out["roofline"]["valu"] = {
                        "wave_insts_per_segment": round(v["wave_insts_per_frame"] / seg, 2),
                        "achieved_ginst_s": round(rate / 1e9, 1), "peak_ginst_s": v["peak_ginst_s"],
                        "frac": round(rate / 1e9 / v["peak_ginst_s"], 4),
                        "active_lane_frac": v.get("active_lane_frac"), "peak_basis": v.get("peak_basis"),
                        "source": pmc.get("source", "profiles/pmc_current.json")}
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["host"]["cpu_threads_used"] = cpu["cores"]
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
