#!/usr/bin/env python3
"""bench.py -- Mpaths/s of the MI355X-native GuidedVolPathVSPG hot path on the 1920x1080
homogeneous-fog scene (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step is one WAVE: one pass of the hot path (EvaluatePixelSample -> Li -> SampleDistance ->
film.AddSample, then PostProcessWave) over one 1-spp batch of 1920x1080 = 2,073,600 camera
paths per GPU.  With N GPUs every rank renders its own sample indices of the same frame
(weak scaling: per-GPU work fixed), and the float film tiles are all-reduced over RCCL at frame
end, inside the timed region.  Inputs (scene, film, VSP buffer) are resident in HBM before the
timed region starts.

The JSON line carries
  roofline     : dominant kernel (k_render_wave_wg) -- algorithmic bytes per launch (SURVEY.md 8d:
                 256 B per path segment + 76 B per path) / mean launch duration measured with
                 HIP events on the launch stream, against the 8 TB/s HBM peak.
  cpu_baseline : the CPU oracle (a port of the reference path; the reference itself cannot be
                 built) timed on this box's host cores on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
B_SEGMENT = 256                # 2 x 128 B SoA path state per segment
B_PATH_FIXED = 32 + 4 + 40     # film RMW + primary-VSP read + ISG sample write
PMC_PROFILE = "r01e_pmc_k_render_wave.json"  # scripts/gpu_profile.sh + scripts/summarize_profile.py


class DevArray:
    """Exposes a raw device pointer to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, n_floats):
        self.__cuda_array_interface__ = {"shape": (n_floats,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def host_threads():
    """CPU threads this process may actually use: cgroup quota, else affinity (the GPU box exposes
    every core of the host in os.cpu_count() but grants a 1-GPU job a 16-core share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("VSPG_CPU_THREADS", "16"))))


def cpu_baseline(pkg, scene, prm, W, H, budget_s=15.0):
    """Oracle timed on the host cores over a bounded sample of the same workload."""
    import oracle_lib

    cpu = oracle_lib.OracleRenderer(scene, prm, W, H)
    cores = host_threads()
    t0 = time.perf_counter()
    cpu.render_wave(0, 1, cores)
    cpu.post_process_wave()
    t1 = time.perf_counter() - t0
    waves = 1
    extra = int(max(0.0, min(budget_s, 30.0) - t1) / max(t1, 1e-3))
    if extra > 0:
        t0 = time.perf_counter()
        for w in range(1, 1 + extra):
            cpu.render_wave(w, w + 1, cores)
            cpu.post_process_wave()
        t1 += time.perf_counter() - t0
        waves += extra
    paths = cpu.counters()["paths"]
    cpu.close()
    return {"value": paths / t1 / 1e6, "unit": "Mpaths/s", "cores": cores, "kind": "port",
            "sample": "%d full-frame 1-spp waves of %dx%d (%d paths) in %.1f s, OpenMP over 16x16 tiles" % (waves, W, H, paths, t1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--xres", type=int, default=1920)
    ap.add_argument("--yres", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["fog", "cloud", "cloud-nvdb"], default="fog",
                    help="fog = BASELINE.json's metric workload (default); cloud = procedural heterogeneous GridMedium (configs 3-4 stand-in); "
                         "cloud-nvdb = the same grid with NanoVDBMedium semantics (64^3 majorants)")
    ap.add_argument("--diag-maxdepth", type=int, default=None,
                    help="DIAGNOSTIC ONLY (not the benchmark config): override maxdepth to time parts of the path")
    ap.add_argument("--diag-guiding", action="store_true",
                    help="DIAGNOSTIC ONLY: the reference's default guiding options (field trained during the warm-up waves)")
    ap.add_argument("--grid", type=int, default=256, help="voxels per axis of the cloud workload's density grid")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    pkg = load_package()
    pkg.load()
    import importlib.util
    spec = importlib.util.spec_from_file_location("vspg_sharding", os.path.join(ROOT, "vspg-pbrt-v4_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    W, H = args.xres, args.yres
    scene = (pkg.fog_box_scene(W, H) if args.workload == "fog" else pkg.cloud_box_scene(W, H, args.grid) if args.workload == "cloud"
             else pkg.nanovdb_box_scene(W, H, args.grid))
    prm = pkg.app_f_params()
    if args.diag_maxdepth is not None:
        prm.maxdepth = args.diag_maxdepth
    if args.diag_guiding:
        prm = pkg.default_params()
        prm.guide_num_training_waves = max(1, args.warmup)
    r = pkg.Renderer(scene, prm, W, H, spp=args.steps * world, seed=0, shard_index=rank, shard_count=world,
                     device=local_rank)
    fptr, fn = r.film_ptr()
    film = torch.as_tensor(DevArray(fptr, fn), device=torch.device("cuda", local_rank))
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        # global waves [i*world, (i+1)*world): this rank runs exactly the one with w % world == rank
        w0, w1 = sh.step_wave_range(i, world)
        r.render_wave(w0, w1, stream)
        r.post_process_wave(stream)

    for i in range(args.warmup):
        step(i)
    # untimed: first use of the communicator at the film's size (RCCL sets up channels / buffers lazily)
    sh.frame_end_allreduce(dist, film, world)
    torch.cuda.synchronize()
    r.film_clear(stream)
    r.reset_counters(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        w0, w1 = sh.step_wave_range(args.warmup + i, world)
        r.render_wave(w0, w1, stream)
        ev[i][1].record()
        r.post_process_wave(stream)
    sh.frame_end_allreduce(dist, film, world)  # frame-end film all-reduce over RCCL / xGMI
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = sh.max_over_ranks(dist, elapsed, world, "cuda")

    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)
    cnt = r.counters()
    paths_rank = cnt["paths"]
    segs_rank = cnt["segments"]
    paths_total, segs_total = sh.sum_over_ranks(dist, [paths_rank, segs_rank], world, "cuda")

    if rank == 0:
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes of this same
        # command (profiles/<PMC_PROFILE>; FETCH_SIZE and WRITE_SIZE are separate passes, KiB per
        # launch).  gfx950 correction of the microarch guide: FETCH_SIZE tallies 128-B read requests
        # at 64 B, so it is doubled (the kernel's reads are 16-B-per-lane film / ISG records, 4-B
        # spill reloads and scalar loads; for the narrow ones the factor is an upper bound);
        # WRITE_SIZE is exact for 16-B-per-lane stores (film / ISG records) and dword stores (spills).
        traffic_gbs, traffic_bytes = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_PROFILE)))
            if W == 1920 and H == 1080 and args.workload == "fog":
                traffic_bytes = (2.0 * pmc["FETCH_SIZE"]["mean_per_launch"] + pmc["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
        except Exception:
            pass
        kbar = segs_rank / max(1, paths_rank)
        # heterogeneous media add 36 B per density query (8 voxels + 1 majorant)
        dq_rank = cnt["density_queries"] if args.workload != "fog" else 0
        bytes_per_launch = (segs_rank * B_SEGMENT + paths_rank * B_PATH_FIXED + dq_rank * 36) / max(1, args.steps)
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        out = {
            "metric": "Mpaths/sec on 1920x1080 homogeneous fog; relMSE vs CPU ref at equal spp" if args.workload == "fog"
                      else "Mpaths/sec on a procedural %d^3 cloud grid (not BASELINE.json's metric workload)" % args.grid,
            "value": paths_total / elapsed / 1e6,
            "unit": "Mpaths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(1, args.steps) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("fog-box %dx%d" % (W, H) if args.workload == "fog" else
                                    "cloud-box %dx%d, %s %d^3 value noise, sigma_t 8, albedo 0.99, g 0.877, resampling" % (
                                        W, H, "GridMedium" if args.workload == "cloud" else "NanoVDBMedium (dense copy, 64^3 majorants)", args.grid)) +
                                   ", guidedvolpathvspg vspguiding=true (primary-ray VSP), 1 spp per step per GPU, "
                                   "independent sampler seed 0, maxdepth %d%s" % (prm.maxdepth, "" if args.diag_maxdepth is None and not args.diag_guiding else " (DIAGNOSTIC override)"),
                       "paths_per_step_per_gpu": W * H, "mean_segments_per_path": kbar,
                       "parallelism": "sample-index sharding x%d, film all-reduce at frame end" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (traffic_bytes / (kern_ms * 1e-3) / 1e9) if traffic_bytes and kern_ms > 0 else None,
                         "traffic_bytes_per_launch": traffic_bytes,
                         "kernel": "k_render_wave_wg" if args.workload == "fog" and not args.diag_guiding and os.environ.get("VSPG_KERNEL") != "lane"
                                   else "k_render_wave", "kernel_ms": kern_ms, "density_queries_per_path": dq_rank / max(1, paths_rank),
                         "algorithmic_bytes_per_launch": bytes_per_launch},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pkg, scene, prm, W, H)
        print(json.dumps(out))
    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
