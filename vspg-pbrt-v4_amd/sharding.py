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


def ordered_after(torch, device, stream, fn):
    """Run fn() -- work torch issues on ITS current stream (torch.distributed's collectives) -- ordered after what `stream`
    holds and before what it gets next: the renderer's kernels run on the stream its caller names.  On a host without a
    device (the gloo transport of the CPU tests) there is nothing to order."""
    if device is None or not hasattr(torch, "cuda") or not torch.cuda.is_available():
        return fn()
    cur = torch.cuda.current_stream(device)
    if stream == cur.cuda_stream:
        return fn()
    if stream:
        ext = torch.cuda.ExternalStream(stream, device=device)
        cur.wait_stream(ext)
        out = fn()
        ext.wait_stream(cur)
        return out
    torch.cuda.synchronize(device)              # the null stream: no handle to wait on
    out = fn()
    torch.cuda.synchronize(device)
    return out


def frame_end_allreduce(dist, film_tensor, world, renderer=None, stream=None, torch=None, device=None):
    """Sum the per-rank film tiles {sum w*rgb, sum w} (RGBFilm accumulate contract, film.h:251-267).

    `film_tensor` wraps the renderer's film pointer, which a host may keep for the renderer's lifetime -- but a one-sample
    wave leaves its samples PARKED beside the film until the next launch (include/vspg.h, vspg_flush): the frame's last wave
    is only in the film after `renderer.flush(stream)`.  Pass the renderer and the stream its waves ran on and the flush
    happens here, at world 1 too (a frame end without the frame's film is not a frame end); pass `torch` and `device` as
    well and the collective is ordered behind that stream whatever torch's current stream is (without them the caller
    vouches that the two are the same stream)."""
    if renderer is not None:
        renderer.flush(stream)
    if world > 1:
        if torch is not None:
            ordered_after(torch, device, stream, lambda: dist.all_reduce(film_tensor, op=dist.ReduceOp.SUM))
        else:
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


def device_tensor(torch, device):
    """wrap(ptr, n): a float32 tensor over n floats of device memory at ptr (no copy)."""
    def wrap(ptr, n):
        class _Dev:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}
        return torch.as_tensor(_Dev(), device=device)
    return wrap


class ShardSync:
    """Cross-rank state of a sharded render: the image-space VSP statistics.

    Every rank accumulates the statistics of its own sample indices; on the steps where the buffer
    update falls (the global wave counter reaches 1, 2, 4, ... -- `isg_update_due`) the per-rank
    statistics are summed over the ranks into a scratch tensor and every rank runs the update on
    the SUM (`post_process_step(world, sum)`), so all ranks hold the buffer ONE renderer would hold
    that renders `world` sample indices per step (up to float summation order).  The renderer's own
    statistics are left as they are: nothing is counted twice.

    `renderer` speaks the C-ABI (include/vspg.h) as vspg-pbrt-v4_amd.Renderer wraps it: flush(stream), isg_stats_ptr(),
    isg_update_due(n), post_process_step(n, ptr_or_None, stream), set_exchange(fn).  `wrap(ptr, n)` makes a tensor of the
    transport's kind over memory the renderer owns (default: device memory, for RCCL)."""

    def __init__(self, dist, renderer, world, torch, device=None, wrap=None):
        self.dist, self.r, self.world, self.torch, self.device = dist, renderer, world, torch, device
        self.wrap = wrap or device_tensor(torch, device)
        self._stats = None
        self._sum = None
        # guiding-field training: the renderer's Field::Update sums its sufficient statistics over the ranks
        # through this hook (vspg_renderer_set_exchange), so every rank fits the same field from all ranks' samples
        if world > 1:
            renderer.set_exchange(self._exchange)

    def _ordered(self, stream, fn):
        return ordered_after(self.torch, self.device, stream, fn)

    def _exchange(self, ptr, n, stream):
        t = self.wrap(ptr, n)
        self._ordered(stream, lambda: self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM))

    def _stats_tensor(self, stream=None):
        # The pointer is fixed for the renderer's lifetime and wrapped once (no copy); what it shows is not: this rank's
        # latest one-sample wave is still PARKED beside the statistics (vspg_flush, include/vspg.h) and enters them here,
        # on the stream the wave ran on, before the sum below reads them.  (Round 3 wrapped the pointer once and never
        # flushed: from the second update on the all-reduced statistics lacked every rank's latest wave.)
        self.r.flush(stream)
        if self._stats is None:
            ptr, n = self.r.isg_stats_ptr()
            self._stats = self.wrap(ptr, n)
        return self._stats

    def post_process_step(self, stream=None):
        if self.world <= 1:
            self.r.post_process_step(1, None, stream)
            return
        total = None
        if self.r.isg_update_due(self.world):
            st = self._stats_tensor(stream)
            if self._sum is None:
                self._sum = self.torch.empty_like(st)

            def _sum_over_ranks():
                self._sum.copy_(st)
                self.dist.all_reduce(self._sum, op=self.dist.ReduceOp.SUM)
            # the wave that filled `st` ran on `stream`, and the update that reads the sum runs there next
            self._ordered(stream, _sum_over_ranks)
            total = self._sum
        self.r.post_process_step(self.world, total.data_ptr() if total is not None else None, stream)
