"""Test adapter: the CPU oracle renderer behind the method names of the HIP renderer (vspg-pbrt-v4_amd.Renderer), so that the
product's sharding code (vspg-pbrt-v4_amd/sharding.py) can be stepped over gloo on a host without a GPU.  The product knows
nothing of the oracle's shape; this file does the translating.  Test infrastructure only."""
import ctypes as C

import numpy as np


class OracleShard:
    def __init__(self, oracle_renderer):
        self.o = oracle_renderer
        self._stats = None  # a host buffer that stays put: the product wraps its address once, like the device pointer

    def __getattr__(self, name):      # render_wave, film_f64, vsp_buffer, counters, ...
        return getattr(self.o, name)

    def flush(self, stream=None):
        """The HIP renderer applies its parked samples; here: refresh the host copy the wrapped pointer shows."""
        st = self.o.isg_stats().reshape(-1)
        if self._stats is None:
            self._stats = np.empty_like(st)
        self._stats[:] = st

    def isg_stats_ptr(self):
        if self._stats is None:
            self.flush()
        return self._stats.ctypes.data, int(self._stats.size)

    def isg_update_due(self, n_waves=1):
        return self.o.isg_update_due(n_waves)

    def post_process_step(self, n_waves, stats_sum_ptr=None, stream=None):
        total = None
        if stats_sum_ptr:
            n = self._stats.size
            total = np.ctypeslib.as_array((C.c_float * n).from_address(int(stats_sum_ptr)))
        self.o.post_process_step(n_waves, total)

    def set_exchange(self, fn):
        pass                          # (the oracle's field update runs on one rank's samples; sharded training is a GPU test)


def host_tensor(torch):
    """wrap(ptr, n) for host memory: the gloo transport's counterpart of sharding.device_tensor."""
    def wrap(ptr, n):
        return torch.from_numpy(np.ctypeslib.as_array((C.c_float * n).from_address(int(ptr))))
    return wrap
