"""bench.py's N > 1 code path end to end on the one-GPU box: two ranks on cuda:0,
gloo exchange (EG_BENCH_ONE_GPU=1) — sharding, the spin-up agreement across ranks, the barrier /
max-over-ranks timing, `single_gpu_reference`, the JSON line of rank 0.  The RCCL exchange itself needs
a second device; its one-rank form is covered in tests/test_gpu_model.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_on_one_gpu(scaling):
    env = dict(os.environ, EG_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # no launcher in the command: `python bench.py --gpus 2` starts its two ranks itself;
    # the strong case goes through torch.distributed.run the way the driver launches it
    if scaling == "weak":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"]
        env.pop("WORLD_SIZE", None)
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29518", os.path.join(ROOT, "bench.py"), "--gpus", "2"]
    cmd += ["--steps", "4", "--warmup", "1", "--batch", "4096", "--scaling", scaling, "--no-extra"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]              # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["scaling"] == scaling
    assert line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 8192
    # two ranks over gloo: no RCCL communicator ran, and the line says so
    assert line["config"]["rccl_ranks"] == 0 and "gloo" in line["config"]["collective"]
    assert line["exchange"]["allreduce_us"] > 0 and line["exchange"]["allreduce_floats"] == line["config"]["grad_bucket_floats"]
    assert line["value"] > 0 and line["single_gpu_reference"]["value"] > 0
    assert line["unit"] == "samples/s" and line["dtype"] == "f32" and line["roofline"]["frac"] > 0
    # the 1-GPU point of the series and the efficiency derived from it are top-level fields of every N > 1 line
    eff = line["value"] / (2 * line["single_gpu_reference"]["value"])
    assert abs(line["scaling_efficiency"] - eff) < 1e-3


def test_one_gpu_train_line_has_the_series_metric():
    """`--gpus 1 --workload train` prints the metric and unit of the N > 1 lines, so a 1/2/4/8 series has one unit."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "train", "--steps", "3", "--warmup", "1",
           "--batch", "4096", "--no-extra", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["unit"] == "samples/s"
    assert line["metric"] == "train samples/s dense net 784-512-10 (data parallel)"
