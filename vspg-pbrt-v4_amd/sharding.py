"""Multi-GPU plumbing of the path: one process per GPU, sample-index sharding,
frame-end all-reduce of the float film tiles.  No data-path collective inside a wave.

The renderer (C-ABI `shard_index/shard_count`, include/vspg.h) runs wave w iff
w % world == rank, so handing every rank the SAME global wave range [step*world, (step+1)*world)
gives each rank exactly one 1-spp wave per step, and the union over ranks covers every sample
index once.  `torch.distributed` is only the transport (backend "nccl" == RCCL over xGMI on the
GPU box, "gloo" in the CPU tests)."""


def step_wave_range(step, world):
    """Global wave range every rank passes to render_wave() at `step`."""
    return step * world, (step + 1) * world


def frame_end_allreduce(dist, film_tensor, world):
    """Sum the per-rank film tiles {sum w*rgb, sum w} (RGBFilm accumulate contract, film.h:251-267)."""
    if world > 1:
        dist.all_reduce(film_tensor, op=dist.ReduceOp.SUM)
    return film_tensor


def max_over_ranks(dist, seconds, world, device):
    import torch
    if world <= 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, values, world, device):
    import torch
    if world <= 1:
        return [float(v) for v in values]
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]
