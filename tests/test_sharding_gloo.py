"""N>1 path on CPU: two gloo ranks, sample-index sharding, frame-end film all-reduce.  The
per-rank compute is the oracle renderer (the HIP path needs a GPU; its shard arithmetic is
covered on the GPU box by test_gpu_parity.py::test_sharded_waves_sum_to_unsharded); the stepping
and reduction code is the one bench.py runs."""
import importlib.util
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

W, H, STEPS = 48, 32, 3


def _load_sharding():
    spec = importlib.util.spec_from_file_location("vspg_sharding", os.path.join(ROOT, "vspg-pbrt-v4_amd", "sharding.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _worker(rank, world, port, out_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    sh = _load_sharding()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    scene = oracle_lib.fog_box_scene(W, H)
    prm = oracle_lib.app_f_params()
    r = oracle_lib.OracleRenderer(scene, prm, W, H, shard_index=rank, shard_count=world)
    for step in range(STEPS):
        w0, w1 = sh.step_wave_range(step, world)
        r.render_wave(w0, w1, 1)  # VSP buffer stays at its initial 0.5: no cross-rank state
    film = torch.from_numpy(r.film_f64().copy())
    paths = r.counters()["paths"]
    sh.frame_end_allreduce(dist, film, world)
    total_paths, = sh.sum_over_ranks(dist, [paths], world, "cpu")
    tmax = sh.max_over_ranks(dist, float(rank + 1), world, "cpu")
    if rank == 0:
        np.save(out_path, film.numpy())
        assert total_paths == W * H * STEPS * world
        assert tmax == float(world)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_render_equals_unsharded(tmp_path):
    import oracle_lib
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "film.npy")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = np.load(out)
    ref = oracle_lib.OracleRenderer(oracle_lib.fog_box_scene(W, H), oracle_lib.app_f_params(), W, H)
    ref.render_wave(0, STEPS * world, 1)
    want = ref.film_f64()
    assert np.array_equal(got[..., 3], want[..., 3])          # every pixel got STEPS*world samples
    assert np.allclose(got, want, rtol=1e-12, atol=1e-14)     # same samples, different summation grouping


def _worker_isg(rank, world, port, out_path, steps):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    sh = _load_sharding()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    scene = oracle_lib.fog_box_scene(W, H)
    prm = oracle_lib.app_f_params()
    import oracle_shard
    r = oracle_shard.OracleShard(oracle_lib.OracleRenderer(scene, prm, W, H, shard_index=rank, shard_count=world))
    sync = sh.ShardSync(dist, r, world, torch, wrap=oracle_shard.host_tensor(torch))
    for step in range(steps):
        w0, w1 = sh.step_wave_range(step, world)
        r.render_wave(w0, w1, 1)
        sync.post_process_step()          # the path bench.py runs: statistics all-reduced where the buffer updates
    film = torch.from_numpy(r.film_f64().copy())
    sh.frame_end_allreduce(dist, film, world)
    vsp, ready = r.vsp_buffer()
    if rank == 0:
        np.savez(out_path, film=film.numpy(), vsp=vsp, ready=ready)
    # every rank must hold the same buffer
    v = torch.from_numpy(vsp.copy())
    vmax, vmin = v.clone(), v.clone()
    dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(vmin, op=dist.ReduceOp.MIN)
    assert torch.equal(vmax, vmin), "ranks disagree on the VSP buffer"
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_render_with_buffer_updates_equals_one_renderer_stepping_two_waves(tmp_path):
    """post_process in the loop: 2 ranks + all-reduced VSP statistics == ONE renderer that renders two sample indices
    per step and post-processes with n_waves = 2 (same statistics up to float summation order)."""
    import oracle_lib
    world, steps = 2, 5
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "isg.npz")
    mp.spawn(_worker_isg, args=(world, port, out, steps), nprocs=world, join=True)
    got = np.load(out)
    ref = oracle_lib.OracleRenderer(oracle_lib.fog_box_scene(W, H), oracle_lib.app_f_params(), W, H)
    updates = 0
    for step in range(steps):
        ref.render_wave(step * world, (step + 1) * world, 1)
        updates += ref.isg_update_due(world)
        ref.post_process_step(world)
    assert updates == 3                       # wave counter 2, 4, 8 (the update at 1 merges into the first step)
    vsp, ready = ref.vsp_buffer()
    assert ready and bool(got["ready"])
    # The summed statistics differ from the single renderer's in float summation order only (last ulp).  From the first
    # update on, a pixel whose VSP differs in the last ulp samples distances that differ in the last ulp, and the hash-seeded
    # shadow-ray RNG (:1193) turns that into a different -- equally distributed -- path: the two renders are the same
    # estimator fed the same camera samples, not the same bits.  Pixels whose sums happened to round alike stay identical.
    dv = np.abs(got["vsp"] - vsp)
    print("VSP buffer: %.3f of the pixels bit-identical, mean |diff| %.2e, max %.2e" % ((dv == 0).mean(), dv.mean(), dv.max()))
    assert (dv == 0).mean() > 0.3 and dv.mean() < 2e-3 and dv.max() < 0.05
    want = ref.film_f64()
    assert np.array_equal(got["film"][..., 3], want[..., 3])
    ig, iw = got["film"][..., :3] / got["film"][..., 3:4], want[..., :3] / want[..., 3:4]
    assert abs(ig.mean() / iw.mean() - 1) < 0.02


def test_post_process_step_of_one_wave_is_post_process_wave():
    import oracle_lib
    a = oracle_lib.OracleRenderer(oracle_lib.fog_box_scene(W, H), oracle_lib.app_f_params(), W, H)
    b = oracle_lib.OracleRenderer(oracle_lib.fog_box_scene(W, H), oracle_lib.app_f_params(), W, H)
    for w in range(5):
        a.render_wave(w, w + 1, 1)
        a.post_process_wave()
        b.render_wave(w, w + 1, 1)
        b.post_process_step(1)
    assert np.array_equal(a.vsp_buffer()[0], b.vsp_buffer()[0])
    assert np.array_equal(a.film(), b.film())


def test_step_wave_range_covers_every_sample_once():
    sh = _load_sharding()
    for world in (1, 2, 4, 8):
        seen = []
        for step in range(5):
            w0, w1 = sh.step_wave_range(step, world)
            for rank in range(world):
                seen += [w for w in range(w0, w1) if w % world == rank]
        assert sorted(seen) == list(range(5 * world))
