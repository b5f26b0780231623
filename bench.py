#!/usr/bin/env python3
"""bench.py -- Mpaths/s of the MI355X-native GuidedVolPathVSPG hot path on the 1920x1080
homogeneous-fog scene (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no torch.distributed environment this process starts its own ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`) as a CHILD process -- before
anything here has touched the GPU -- and relays rank 0's JSON line; started under
torch.distributed.run it is one of the ranks.

A step is one WAVE: one pass of the hot path (EvaluatePixelSample -> Li -> SampleDistance ->
film.AddSample, then PostProcessWave) over one 1-spp batch of 1920x1080 = 2,073,600 camera
paths per GPU.  With N GPUs every rank renders its own sample indices of the same frame
(weak scaling: per-GPU work fixed); the image-space VSP statistics are all-reduced at the waves
where the buffer updates (1, 2, 4, ... global waves) and the float film tiles at frame end, over
RCCL, inside the timed region.  Inputs (scene, film, VSP buffer) are resident in HBM before the
timed region starts.

The JSON line carries
  roofline     : dominant kernel -- algorithmic bytes per launch (256 B per path
                 segment + 76 B per path [+ 36 B per density query]) / mean launch duration measured
                 with HIP events on the launch stream, against the 8 TB/s HBM peak.
  cpu_baseline : the CPU oracle (a port of the reference path; the reference itself cannot be
                 built) timed on this box's host cores on a bounded sample.
  relmse       : the metric's second half -- the same waves rendered by the GPU path and by the
                 CPU oracle at equal spp and seeds: relMSE, bit-identical pixel fraction, max abs
                 difference, and the relMSE of each against a 16x-spp render (noise floor).
  generic_instantiation : the same workload through the kernel instantiation a chromatic medium
                 takes (no grey-spectrum / zero-null-coefficient specialisation), untimed for `value`.
  reference_defaults : the same scene with the reference's DEFAULT integrator options (surface RIS + volume MIS
                 guiding, primary + secondary VSP: the cache query in the loop) -- trained-wave Mpaths/s, training-wave
                 ms and the effective rate of a 256-spp frame whose first 128 waves train; untimed for `value`.
  roofline.issue_bound / roofline.traffic : counters of THIS run -- bench.py profiles itself first (three short
                 `rocprofv3 --pmc` child runs of the same workload, before this process touches the GPU): vector
                 instructions per launch, lane utilisation, the issue fraction they price to, HBM bytes per launch.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
B_SEGMENT = 256                # 2 x 128 B SoA path state per segment
B_PATH_FIXED = 32 + 4 + 40     # film RMW + primary-VSP read + ISG sample write
B_DENSITY_QUERY = 36           # heterogeneous media: 8 voxels x 4 B + 4 B majorant per density query
RELMSE_EPS = 1e-4              # mean (I_gpu - I_cpu)^2 / (I_cpu^2 + eps)


def csrc_hash():
    """Identity of the kernel sources a PMC summary was taken on (profiles/*_pmc_*.json carry it)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "vspg-pbrt-v4_amd", "csrc")
    for n in sorted(os.listdir(d)):
        if n.endswith((".h", ".hip")) or n == "Makefile":
            h.update(n.encode())
            h.update(open(os.path.join(d, n), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic_bytes(workload, W, H):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary of
    this same command taken on THESE kernel sources (profiles/*_pmc_<workload>.json, written by
    scripts/summarize_profile.py with the source hash); None when the sources have changed since."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for n in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not (n.endswith(".json") and "_pmc_" in n):
            continue
        try:
            pmc = json.load(open(os.path.join(pdir, n)))
        except Exception:
            continue
        if pmc.get("_csrc_hash") != csrc_hash() or pmc.get("_workload", "fog") != workload or pmc.get("_res", [1920, 1080]) != [W, H]:
            continue
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            # gfx950 correction of the microarch guide: FETCH_SIZE tallies 128-B read requests at 64 B -> x2;
            # WRITE_SIZE is exact for 16-B-per-lane and dword stores.  Both KiB per launch.
            best = (2.0 * pmc["FETCH_SIZE"]["mean_per_launch"] + pmc["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
    return best


PMC_GROUPS = (["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVES"], ["FETCH_SIZE"], ["WRITE_SIZE"])
# scripts/microbench/issue.hip on MI355X (profiles/r03_microbench_issue.txt): SIMD cycles one wave-instruction of a path-kernel-like
# mix costs with four waves per SIMD issuing (the hardware floor is 2: MI355X_MICROARCH.md, wave scheduling), and the clock the chip held
ISSUE_CYCLES_PER_INST = 2.47
ISSUE_CLOCK_GHZ = 2.38


def live_pmc(args):
    """Counters of this run's workload: rocprofv3 --pmc child runs of `bench.py --pmc-child` (own passes per counter group:
    FETCH_SIZE and WRITE_SIZE do not share one, MI355X_MICROARCH.md), started BEFORE this process touches the GPU.  Returns
    {kernel name: {counter: mean per launch}} over the workload's path kernels, or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile

    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not found"
    # Never profile from inside a profiled run: the child would inherit the outer profiler's preloaded tool library, which
    # initialises the GPU in rocprofv3's own `env python3` launcher hop before that hop execs -- the exec of a GPU-initialised
    # process this pool forbids.  (Scripts that run bench.py under rocprofv3 also pass --no-pmc; this is the belt to that brace.)
    outer = [k for k in os.environ if k.startswith(("ROCP", "ROCPROF"))]
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "HSA_TOOLS_LIB")) or outer:
        return None, "already under a profiler (%s): no self-profiling" % ", ".join(sorted(set(outer + [k for k in ("LD_PRELOAD", "HSA_TOOLS_LIB") if k in os.environ])))
    child_env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "HSA_TOOLS_LIB") and not k.startswith(("ROCP", "ROCPROF"))}
    child_env["TMPDIR"] = "/tmp"
    agg = {}
    for group in PMC_GROUPS:
        d = tempfile.mkdtemp(prefix="vspg_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc"] + group + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
               "--workload", args.workload, "--xres", str(args.xres), "--yres", str(args.yres), "--grid", str(args.grid), "--cloud-shape", args.cloud_shape, "--steps", "3", "--warmup", "2",
               "--train-waves", str(min(args.train_waves, 8))]
        try:
            res = subprocess.run(cmd, cwd="/tmp", env=child_env, timeout=240, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        except Exception as e:  # noqa: BLE001
            shutil.rmtree(d, ignore_errors=True)
            return None, "rocprofv3 child: %s" % e
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if res.returncode != 0 or not files:
            shutil.rmtree(d, ignore_errors=True)
            return None, "rocprofv3 child rc=%d: %s" % (res.returncode, res.stdout[-300:])
        for f in files:
            rows = list(csv.DictReader(open(f)))
            # a guided child trains its field first: only the launches BEHIND the last training kernel (k_propagate, k_train_*,
            # in dispatch order) belong to the trained waves the line reports -- 2 warm-up + 3 timed ones, like the unguided child
            last_train = max([int(r_["Dispatch_Id"]) for r_ in rows if "k_propagate" in r_["Kernel_Name"] or "k_train_" in r_["Kernel_Name"]], default=-1)
            for row in rows:
                k = row["Kernel_Name"]
                if int(row["Dispatch_Id"]) <= last_train or ("k_render_wave" not in k and "k_wf_" not in k):
                    continue
                agg.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        shutil.rmtree(d, ignore_errors=True)
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in cs.items()} for k, cs in agg.items()}, None


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


def cpu_baseline(pkg, scene, prm, W, H, budget_s=15.0, min_waves=4):
    """Oracle timed on the host cores over a bounded sample of the same workload.  Returns the
    baseline record, the oracle's film after those waves and the wave count (for the relMSE leg)."""
    import oracle_lib

    cpu = oracle_lib.OracleRenderer(scene, prm, W, H)
    cores = host_threads()
    t0 = time.perf_counter()
    cpu.render_wave(0, 1, cores)
    cpu.post_process_wave()
    t1 = time.perf_counter() - t0
    waves = 1
    extra = max(min_waves - 1, int(max(0.0, min(budget_s, 30.0) - t1) / max(t1, 1e-3)))
    extra = min(extra, 63)
    if extra > 0:
        t0 = time.perf_counter()
        for w in range(1, 1 + extra):
            cpu.render_wave(w, w + 1, cores)
            cpu.post_process_wave()
        t1 += time.perf_counter() - t0
        waves += extra
    paths = cpu.counters()["paths"]
    film = cpu.film()
    cpu.close()
    rec = {"value": paths / t1 / 1e6, "unit": "Mpaths/s", "cores": cores, "kind": "port",
           "sample": "%d full-frame 1-spp waves of %dx%d (%d paths) in %.1f s, OpenMP over 16x16 tiles" % (waves, W, H, paths, t1)}
    return rec, film, waves


def film_image(f):
    import numpy as np
    w = np.maximum(f[..., 3:4], 1e-30)
    return (f[..., :3] / w).astype(np.float64)


def relmse_leg(pkg, scene, prm, W, H, cpu_film, waves, device, torch):
    """relMSE vs the CPU oracle at equal spp on identical seeds (untimed), plus the relMSE of both
    against a 16x-spp render of an independent seed (the noise floor that separates bias from noise)."""
    import numpy as np

    g = pkg.Renderer(scene, prm, W, H, spp=waves, seed=0, device=device)
    for w in range(waves):
        g.render_wave(w, w + 1)
        g.post_process_wave()
    gf = g.film()
    g.close()
    ig, ic = film_image(gf), film_image(cpu_film)
    rel = (ig - ic) ** 2 / (ic ** 2 + RELMSE_EPS)
    same = np.all(gf == cpu_film, axis=-1)
    # the film sums float on the GPU and double in the oracle (the reference's RGBFilm, film.h:316-317): individual PATHS are the
    # sharper statement -- 20 000 (pixel, sample) pairs replayed on both sides, radiance compared bit for bit
    import oracle_lib
    rng = np.random.default_rng(1)
    pix = np.stack([rng.integers(0, W, 20000), rng.integers(0, H, 20000)], axis=1).astype(np.int32)
    si = rng.integers(0, 4096, 20000).astype(np.int32)
    g2 = pkg.Renderer(scene, prm, W, H, device=device)
    Lg, _ = g2.trace_paths(pix, si)
    g2.close()
    c2 = oracle_lib.OracleRenderer(scene, prm, W, H)
    Lc, _ = c2.trace_paths(pix, si)
    c2.close()
    path_same = float(np.mean(np.all(Lg.view(np.uint32) == Lc.view(np.uint32), axis=1)))
    t = pkg.Renderer(scene, prm, W, H, spp=16 * waves, seed=7919, device=device)
    for w in range(16 * waves):
        t.render_wave(w, w + 1)
        t.post_process_wave()
    it = film_image(t.film())
    t.close()
    return {"value": float(rel.mean()), "spp": waves, "bit_identical_pixel_frac": float(same.mean()),
            "bit_identical_path_frac": path_same, "paths_compared": 20000,
            "max_abs_diff": float(np.abs(ig - ic).max()), "eps": RELMSE_EPS,
            "vs": "CPU oracle (port of the reference path), same seeds, float film on both sides",
            "vs_16x_truth": {"gpu": float(((ig - it) ** 2 / (it ** 2 + RELMSE_EPS)).mean()),
                             "cpu": float(((ic - it) ** 2 / (it ** 2 + RELMSE_EPS)).mean()),
                             "truth": "GPU path, %d spp, independent seed" % (16 * waves)}}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--xres", type=int, default=1920)
    ap.add_argument("--yres", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the cpu_baseline and relmse legs")
    ap.add_argument("--no-generic", action="store_true", help="skip the generic-instantiation leg")
    ap.add_argument("--workload", choices=["fog", "fog-guided", "cloud", "cloud-nvdb", "cloud-guided", "cloud-nvdb-guided", "cloud-scene", "cloud-scene-nvdb",
                                           "cloud-scene-guided", "cloud-scene-nvdb-guided"], default="fog",
                    help="fog = BASELINE.json's metric workload (default); fog-guided = the same scene with the reference's DEFAULT "
                         "integrator options (directional guiding + secondary-ray VSP: cache query in the loop; the field trains "
                         "in-loop during untimed waves, reported separately); cloud = procedural heterogeneous GridMedium "
                         "(configs 3-4 stand-in); cloud-nvdb = the same grid with NanoVDBMedium semantics (64^3 majorants); "
                         "cloud-scene* = the same cloud in the SHAPE of the reference's cloud scenes: camera in vacuum, the medium behind an "
                         "interface-material bounding sphere (MediumInterface + Material \"interface\"), ground, sun + sky")
    ap.add_argument("--diag-maxdepth", type=int, default=None,
                    help="DIAGNOSTIC ONLY (not the benchmark config): override maxdepth to time parts of the path")
    ap.add_argument("--train-waves", type=int, default=16, help="fog-guided: in-loop training waves before the timed region")
    ap.add_argument("--grid", type=int, default=256, help="voxels per axis of the cloud workload's density grid")
    ap.add_argument("--cloud-shape", choices=["noise", "blob"], default="noise",
                    help="cloud workloads: value noise filling the medium's bounds (default) or the same noise inside a ball with empty space around it")
    ap.add_argument("--no-pmc", action="store_true", help="skip the self-profiling child runs (roofline.issue_bound / live traffic)")
    ap.add_argument("--no-fast-arith", action="store_true", help="skip the tolerance-mode leg (fast_arith)")
    ap.add_argument("--no-reference-defaults", action="store_true", help="skip the reference-default-options leg of the default line")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the run rocprofv3 watches: waves only, no output
    return ap.parse_args()


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # not under torch.distributed.run: start the ranks as a child process.  Nothing in this process has touched
        # the GPU yet (no torch import, no HIP call), and it never will: it only relays the child's exit code.
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    # self-profiling (rank 0 of a 1-GPU run): child processes, before anything here has touched the GPU
    pmc, pmc_note = None, "skipped"
    if not args.pmc_child and not args.no_pmc and args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.diag_maxdepth is None:
        pmc, pmc_note = live_pmc(args)

    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    # VSPG_BENCH_REHEARSE=1 (1-GPU boxes only; never a benchmark number): every rank on device 0, collectives over gloo -- walks
    # the whole N-rank code path (self-launch, sharded steps with the statistics exchange, film all-reduce) where one card exists
    rehearse = world > 1 and os.environ.get("VSPG_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    pkg = load_package()
    pkg.load()
    import importlib.util
    spec = importlib.util.spec_from_file_location("vspg_sharding", os.path.join(ROOT, "vspg-pbrt-v4_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    W, H = args.xres, args.yres
    fog = args.workload in ("fog", "fog-guided")
    guided = args.workload.endswith("-guided")
    bounded = args.workload.startswith("cloud-scene")
    scene = (pkg.fog_box_scene(W, H) if fog else pkg.cloud_scene(W, H, args.grid, shape=args.cloud_shape, nvdb="nvdb" in args.workload) if bounded
             else pkg.cloud_box_scene(W, H, args.grid, shape=args.cloud_shape) if args.workload in ("cloud", "cloud-guided")
             else pkg.nanovdb_box_scene(W, H, args.grid, shape=args.cloud_shape))
    prm = pkg.app_f_params()
    if args.diag_maxdepth is not None:
        prm.maxdepth = args.diag_maxdepth
    if guided:
        prm = pkg.default_params()  # GuidedVolPathVSPGIntegrator::Create defaults (:1263-1319)
        prm.guide_num_training_waves = max(1, args.train_waves)
    total_waves = (args.warmup + args.steps + (args.train_waves if guided else 0)) * world
    r = pkg.Renderer(scene, prm, W, H, spp=total_waves, seed=0, shard_index=rank, shard_count=world, device=local_rank)
    fptr, fn = r.film_ptr()
    film = torch.as_tensor(DevArray(fptr, fn), device=torch.device("cuda", local_rank))
    sync = sh.ShardSync(dist, r, world, torch, device=torch.device("cuda", local_rank))
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        # global waves [i*world, (i+1)*world): this rank runs exactly the one with w % world == rank
        w0, w1 = sh.step_wave_range(i, world)
        r.render_wave(w0, w1, stream)
        sync.post_process_step(stream)

    train_ms = None
    step0 = 0
    if guided:  # in-loop training (guideTraining, :109, :230-248): timed on its own, never part of `value`
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.train_waves):
            step(i)
        torch.cuda.synchronize()
        train_ms = (time.perf_counter() - t0) / max(1, args.train_waves) * 1e3
        step0 = args.train_waves
    for i in range(args.warmup):
        step(step0 + i)
    if args.pmc_child:  # the waves rocprofv3 counts: the timed loop's launches, nothing else
        for i in range(args.steps):
            step(step0 + args.warmup + i)
        torch.cuda.synchronize()
        r.close()
        return
    ranks_seen = None
    if world > 1:  # launch check: every rank of the communicator answers (an all-reduce of ones)
        ones = torch.ones(1, dtype=torch.int32, device="cpu" if rehearse else torch.device("cuda", local_rank))
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(ones.item())
    # untimed: first use of the communicator at the film's size (RCCL sets up channels / buffers lazily)
    sh.frame_end_allreduce(dist, film, world, r, stream, torch=torch, device=torch.device("cuda", local_rank))
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
        w0, w1 = sh.step_wave_range(step0 + args.warmup + i, world)
        r.render_wave(w0, w1, stream)
        ev[i][1].record()
        sync.post_process_step(stream)
    # frame end: the last wave's parked samples enter the film (vspg_flush: k_film_resolve on the render stream), then the film
    # all-reduce over RCCL / xGMI -- both inside the timed region
    sh.frame_end_allreduce(dist, film, world, r, stream, torch=torch, device=torch.device("cuda", local_rank))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = sh.max_over_ranks(dist, elapsed, world, "cuda")

    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)
    # every pixel of the (summed) film holds exactly the frame's samples: `steps` per rank, every rank's last wave included
    fw = film.view(-1, 4)[:, 3]
    film_weight_ok = bool(((fw == float(args.steps * world)).all()).item())
    cnt = r.counters()
    paths_rank = cnt["paths"]
    segs_rank = cnt["segments"]
    paths_total, segs_total = sh.sum_over_ranks(dist, [paths_rank, segs_rank], world, "cuda")
    kernel_name = r.kernel_name()
    vsp_trained = r.vsp_buffer(stream)[0].copy() if rank == 0 else None   # (the fast_arith leg loads it: no feedback from film statistics)
    r.close()

    if rank == 0:
        kbar = segs_rank / max(1, paths_rank)
        dq_rank = cnt["density_queries"] if not fog else 0
        bytes_per_launch = (segs_rank * B_SEGMENT + paths_rank * B_PATH_FIXED + dq_rank * B_DENSITY_QUERY) / max(1, args.steps)
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        diag = args.diag_maxdepth is not None
        pipeline = not fog  # heterogeneous media: the four-kernel wavefront pipeline, kern_ms spans all its launches of a wave
        # ---- this run's counters (live_pmc): per WAVE over the workload's path kernels
        traffic_bytes, traffic_src, issue = None, None, None
        if pmc:
            def per_wave(counter):
                tot, seen = 0.0, False
                for cs in pmc.values():
                    if counter in cs:
                        mean, n = cs[counter]
                        tot += mean * n / 5.0   # the child ran 2 warm-up + 3 timed waves; a pipeline kernel launches several times per wave
                        seen = True
                return tot if seen else None
            fs, ws = per_wave("FETCH_SIZE"), per_wave("WRITE_SIZE")
            if fs is not None and ws is not None:
                # gfx950 correction of the microarch guide: FETCH_SIZE tallies 128-B read requests at 64 B -> x2; both in KiB
                traffic_bytes = (2.0 * fs + ws) * 1024.0
                traffic_src = "this run: rocprofv3 --pmc child passes (FETCH_SIZE x2 + WRITE_SIZE), per wave over the path kernels"
            insts, act, thr = per_wave("SQ_INSTS_VALU"), per_wave("SQ_ACTIVE_INST_VALU"), per_wave("SQ_THREAD_CYCLES_VALU")
            if insts and act and thr and kern_ms > 0:
                n_simd = torch.cuda.get_device_properties(local_rank).multi_processor_count * 4
                cyc = kern_ms * 1e-3 * ISSUE_CLOCK_GHZ * 1e9
                issue = {"valu_insts_per_launch": insts, "lane_util": thr / (64.0 * act),
                         "est_issue_frac": insts / n_simd * ISSUE_CYCLES_PER_INST / cyc,
                         "est_issue_frac_at_2_cycle_floor": insts / n_simd * 2.0 / cyc,
                         "cycles_per_wave_inst": ISSUE_CYCLES_PER_INST, "clock_ghz": ISSUE_CLOCK_GHZ, "simds": n_simd,
                         "note": "vector wave-instructions of one wave's launches (SQ_INSTS_VALU) / SIMDs x the measured price of a path-kernel-like "
                                 "mix at four waves per SIMD (profiles/r03_microbench_issue.txt) / kernel cycles: the fraction of the time the vector "
                                 "pipes issue.  lane_util = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU).  This, not HBM, is what bounds the kernel"}
        if traffic_bytes is None:
            traffic_bytes = pmc_traffic_bytes(args.workload, W, H)
            traffic_src = "committed profile taken on these kernel sources (profiles/*_pmc_*.json)" if traffic_bytes else None
        if args.workload == "fog":
            metric = "Mpaths/sec on 1920x1080 homogeneous fog; relMSE vs CPU ref at equal spp"
            wl = "fog-box %dx%d, guidedvolpathvspg vspguiding=true (primary-ray VSP; App. F options)" % (W, H)
        elif guided and fog:
            metric = "Mpaths/sec on 1920x1080 homogeneous fog, reference-default integrator options (not BASELINE.json's metric configuration)"
            wl = ("fog-box %dx%d, guidedvolpathvspg with the reference's default options (surface RIS + volume MIS guiding, "
                  "primary + secondary VSP), field trained in-loop for %d waves before the timed region" % (W, H, args.train_waves))
        else:
            metric = "Mpaths/sec on a procedural %d^3 cloud grid (not BASELINE.json's metric workload)" % args.grid
            wl = "%s %dx%d, %s %d^3 value noise%s, sigma_t 8, albedo 0.99, g 0.877, resampling" % (
                "cloud scene (camera in vacuum, medium behind an interface-material sphere, ground, sun + sky)" if bounded else "cloud-box",
                W, H, "GridMedium" if "nvdb" not in args.workload else "NanoVDBMedium (brick layout, 64^3 majorants)", args.grid,
                " inside a ball (80 % empty voxels)" if args.cloud_shape == "blob" else "")
            if guided:  # config 5's shape: secondary-ray VSP + cache train + query on a heterogeneous medium
                wl += ("; the reference's default options (surface RIS + volume MIS guiding, primary + secondary VSP), field trained "
                       "in-loop for %d waves before the timed region" % args.train_waves)
        if pipeline:
            note = ("achieved = ALGORITHMIC bytes (SURVEY 8d: 256 B per segment + 76 B per path + 36 B per density query) / the time of one "
                    "wave's launches of the wavefront pipeline (k_wf_start, then k_wf_dist_walk, k_wf_vertex, k_wf_shadow_walk -- or k_wf_walk for both "
                    "walks, k_wf_vertex -- x (maxdepth + 1) iterations; the path records live in HBM): `kernel` names the pipeline by its walk "
                    "kernel, `kernel_ms` is the whole wave")
        else:
            note = ("achieved = ALGORITHMIC bytes (SURVEY 8d) / kernel time: the path state lives in LDS, so this is a notional rate -- the "
                    "kernel is bound by vector issue and latency (issue_bound), not by HBM")
        out = {
            "metric": metric,
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
            "config": {"workload": wl + ", 1 spp per step per GPU, independent sampler seed 0, maxdepth %d%s" % (
                           prm.maxdepth, " (DIAGNOSTIC override)" if diag else ""),
                       "paths_per_step_per_gpu": W * H, "mean_segments_per_path": kbar,
                       "parallelism": "sample-index sharding x%d, VSP statistics all-reduced at buffer updates, film all-reduce at frame end" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (traffic_bytes / (kern_ms * 1e-3) / 1e9) if traffic_bytes and kern_ms > 0 else None,
                         "traffic_bytes_per_launch": traffic_bytes, "traffic_source": traffic_src,
                         "issue_bound": issue, "pmc_note": pmc_note,
                         "kernel": ("wavefront pipeline: " if pipeline else "") + kernel_name, "kernel_ms": kern_ms,
                         "density_queries_per_path": dq_rank / max(1, paths_rank),
                         # the NEE shadow rays' ratio tracking fetches densities too (8 voxels + a majorant per tentative collision); SURVEY 8d's
                         # model counts like the reference's densityQueryCount (:692, :886), which leaves them out: reported beside it, so that
                         # traffic / algorithmic bytes can be split into work the model omits and waste
                         "shadow_density_queries_per_path": cnt.get("shadow_density_queries", 0) / max(1, paths_rank) if not fog else 0.0,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "algorithmic_bytes_incl_shadow_queries_per_launch": bytes_per_launch + (cnt.get("shadow_density_queries", 0) * B_DENSITY_QUERY / max(1, args.steps) if not fog else 0.0),
                         "note": note},
        }
        out["film_weight_ok"] = film_weight_ok
        if ranks_seen is not None:
            out["rccl_ranks_seen"] = ranks_seen
        if train_ms is not None:
            out["training"] = {"waves": args.train_waves, "ms_per_wave": train_ms,
                               "note": "render with segment recording + PropagateSamples + Field::Update, host-timed, untimed for value"}
        if args.workload == "fog" and not args.no_generic and world == 1 and not diag:
            # the instantiation a chromatic medium takes: every specialisation off, same scene, same results
            for k in ("VSPG_NO_GREY", "VSPG_NO_GREY_KD", "VSPG_NO_NULLZERO"):
                os.environ[k] = "1"
            g = pkg.Renderer(scene, prm, W, H, spp=args.warmup + 16, seed=0, device=local_rank)
            for k in ("VSPG_NO_GREY", "VSPG_NO_GREY_KD", "VSPG_NO_NULLZERO"):
                del os.environ[k]
            for i in range(args.warmup):
                g.render_wave(i, i + 1, stream)
                g.post_process_wave(stream)
            g.reset_counters(stream)
            gev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(16)]
            torch.cuda.synchronize()
            tg = time.perf_counter()
            for i in range(16):
                gev[i][0].record()
                g.render_wave(args.warmup + i, args.warmup + i + 1, stream)
                gev[i][1].record()
                g.post_process_wave(stream)
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg
            out["generic_instantiation"] = {"value": g.counters()["paths"] / tg / 1e6, "unit": "Mpaths/s", "steps": 16,
                                            "kernel_ms": sum(a.elapsed_time(b) for a, b in gev) / 16, "kernel": g.kernel_name(),
                                            "note": "same scene through the instantiation a chromatic medium / coloured walls take"}
            g.close()
        if args.workload in ("fog", "cloud", "cloud-scene") and not args.no_fast_arith and world == 1 and not diag:
            # the tolerance-mode instantiations (csrc/vspg_arith.h): the same waves through vspg_renderer_set_arithmetic.  Untimed for
            # `value` (that stays the bit-exact kernel).  relMSE against the EXACT film at equal spp with the same loaded VSP buffer (what
            # each mode promises is in tests/test_fast_arith.py); flipped paths: replayed (pixel, sample) pairs whose radiance leaves
            # the exact replay's by more than 1e-5 relative.
            import numpy as np
            n_meas = 16 if fog else 6
            vsp_fixed = vsp_trained
            rng = np.random.default_rng(11)
            pix = np.stack([rng.integers(0, W, 20000), rng.integers(0, H, 20000)], axis=1).astype(np.int32)
            si = rng.integers(0, 4096, 20000).astype(np.int32)
            films, fa = {}, {}
            Lx = None
            for mode, label in ((pkg.ARITH_EXACT, "exact"), (pkg.ARITH_FAST_WEIGHTS, "fast_weights"), (pkg.ARITH_FAST, "fast")):
                g = pkg.Renderer(scene, prm, W, H, spp=2 + n_meas, seed=0, device=local_rank)
                g.load_vsp_buffer(vsp_fixed, stream)
                g.set_arithmetic(mode)
                for i in range(2):
                    g.render_wave(i, i + 1, stream)
                g.reset_counters(stream)
                gev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_meas)]
                torch.cuda.synchronize()
                tg = time.perf_counter()
                for i in range(n_meas):
                    gev[i][0].record()
                    g.render_wave(2 + i, 3 + i, stream)
                    gev[i][1].record()
                g.flush(stream)
                torch.cuda.synchronize()
                tg = time.perf_counter() - tg
                paths_m = g.counters()["paths"]
                films[label] = film_image(g.film())
                Lg, sg = g.trace_paths(pix, si)
                if mode == pkg.ARITH_EXACT:
                    Lx, sx = Lg, sg
                fa[label] = {"value": paths_m / tg / 1e6, "unit": "Mpaths/s", "steps": n_meas, "ms_per_step": tg / n_meas * 1e3,
                             "kernel_ms": sum(a.elapsed_time(b) for a, b in gev) / n_meas, "kernel": g.kernel_name()}
                if mode != pkg.ARITH_EXACT:
                    ix = films["exact"]
                    fa[label]["relmse_vs_exact"] = float(((films[label] - ix) ** 2 / (ix ** 2 + RELMSE_EPS)).mean())
                    fa[label]["flipped_path_frac"] = float(1.0 - np.mean(np.all(np.abs(Lg - Lx) <= 1e-5 * (np.abs(Lx) + 1e-3), axis=1)))
                    fa[label]["same_segment_count_frac"] = float(np.mean(sg == sx))
                g.close()
            fa["note"] = ("csrc/vspg_arith.h: fast_weights = contribution-only quotients through v_rcp_f32 (trajectories stay exact); fast = every division / sqrt "
                          "at 2.5 ulp + native log / sin / cos (equal in distribution only: the reference seeds shadow-ray RNGs from position bits, :1193). "
                          "%d spp, VSP buffer loaded (no feedback); kernel_ms = HIP-event time of vspg_render_wave (the whole pipeline pass for the cloud workloads)" % (2 + n_meas))
            out["fast_arith"] = fa
        if args.workload == "fog" and not args.no_reference_defaults and world == 1 and not diag:
            # the configuration a `guidedvolpathvspg` user gets by default (:1263-1319): directional guiding + secondary-ray VSP,
            # i.e. the cache query in the loop.  The field trains in-loop for the first waves, like the reference's first 128.
            dprm = pkg.default_params()
            n_train, n_meas = 32, 16
            dprm.guide_num_training_waves = n_train
            g = pkg.Renderer(scene, dprm, W, H, spp=n_train + 4 + n_meas, seed=0, device=local_rank)
            train_kernel = g.kernel_name()
            tms = []
            for i in range(n_train):
                torch.cuda.synchronize()
                tt = time.perf_counter()
                g.render_wave(i, i + 1, stream)
                g.post_process_wave(stream)
                torch.cuda.synchronize()
                tms.append((time.perf_counter() - tt) * 1e3)
            for i in range(4):
                g.render_wave(n_train + i, n_train + i + 1, stream)
                g.post_process_wave(stream)
            g.reset_counters(stream)
            gev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_meas)]
            torch.cuda.synchronize()
            tg = time.perf_counter()
            for i in range(n_meas):
                gev[i][0].record()
                g.render_wave(n_train + 4 + i, n_train + 5 + i, stream)
                gev[i][1].record()
                g.post_process_wave(stream)
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg
            gcnt = g.counters()
            trained_ms = tg / n_meas * 1e3
            train_ms_late = sum(tms[n_train // 2:]) / (n_train - n_train // 2)
            out["reference_defaults"] = {
                "value": gcnt["paths"] / tg / 1e6, "unit": "Mpaths/s (trained waves)", "steps": n_meas, "ms_per_trained_wave": trained_ms,
                "kernel_ms": sum(a.elapsed_time(b) for a, b in gev) / n_meas, "kernel": g.kernel_name(),
                "training": {"waves": n_train, "ms_per_wave": train_ms_late, "kernel": train_kernel,
                             "note": "render with segment recording + PropagateSamples + Field::Update, host-timed; mean of the last %d" % (n_train - n_train // 2)},
                "effective_256spp": {"value": W * H * 256 / ((128 * train_ms_late + 128 * trained_ms) * 1e-3) / 1e6, "unit": "Mpaths/s",
                                     "note": "a 256-spp frame with guidenumtrainingwaves 128 (the reference's default): 128 training + 128 trained waves"},
                "mean_segments_per_path": gcnt["segments"] / max(1, gcnt["paths"]),
                "note": "same scene, the reference's default options (surface RIS + volume MIS guiding, primary + secondary VSP): untimed for `value`"}
            g.close()
        if not args.no_cpu_baseline and world == 1:
            # the relMSE leg needs the same waves on both sides: at least 4 spp (training waves would make the oracle
            # side minutes long for fog-guided: that workload reports cpu_baseline on its untrained first waves only)
            rec, cpu_film, waves = cpu_baseline(pkg, scene, prm if not guided else pkg.app_f_params(), W, H)
            out["cpu_baseline"] = rec
            if not guided:
                out["relmse"] = relmse_leg(pkg, scene, prm, W, H, cpu_film, waves, local_rank, torch)
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
