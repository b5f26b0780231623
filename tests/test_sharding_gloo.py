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


def test_step_wave_range_covers_every_sample_once():
    sh = _load_sharding()
    for world in (1, 2, 4, 8):
        seen = []
        for step in range(5):
            w0, w1 = sh.step_wave_range(step, world)
            for rank in range(world):
                seen += [w for w in range(w0, w1) if w % world == rank]
        assert sorted(seen) == list(range(5 * world))
