"""bench.py's N-rank path on the HIP renderer, rehearsed on ONE card (VSPG_BENCH_REHEARSE=1: both ranks on device 0, collectives over
gloo -- never a benchmark number): the driver's launch line (torch.distributed.run, one rank per "GPU"), sample-index sharding, the
VSP-statistics exchange at buffer updates (ShardSync), the film all-reduce at frame end, the max-over-ranks timing, one JSON line
from rank 0.  The 8-GPU run itself is the driver's; what can break without hardware is this plumbing."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["fog", "fog-guided", "cloud-scene"])
def test_two_rank_bench_rehearsal_on_one_card(gpu_pkg, workload):
    env = dict(os.environ, VSPG_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--steps", "4",
           "--warmup", "1", "--train-waves", "2", "--xres", "320", "--yres", "240", "--grid", "64", "--no-cpu-baseline", "--no-generic",
           "--no-fast-arith", "--no-reference-defaults"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]          # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["film_weight_ok"] is True                  # every pixel holds steps x world samples after the all-reduce
    assert d["value"] > 0 and abs(d["value"] - 2 * 320 * 240 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]   # whole-job paths / max-over-ranks time
    assert "x2" in d["config"]["parallelism"] and d["config"]["paths_per_step_per_gpu"] == 320 * 240
