"""The RCCL code path on the GPU: ONE fresh child process (tests/rccl_child.py) sets up a world-size-1
`nccl` process group on cuda:0 and runs raytracer_amd.distributed.render_sharded on a real DeviceScene against a plain
render.  A fresh process because a process group is process-wide state (and this one has long initialised the GPU);
one rank because the box has one GPU — more ranks are the driver's 8-GPU run.  bench.py's own multi-rank plumbing is
covered the same way: `torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` in both scaling modes."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(_free_port())
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_render_sharded_over_a_one_rank_nccl_group_equals_a_plain_render():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py")], env=_env(), capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "RCCL_CHILD_OK" in p.stdout


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_runs_under_the_launcher_with_one_rank(scaling):
    """bench.py under torch.distributed.run with one rank: the nccl group, the accumulator all-reduce, the MAX/SUM of
    the timings and the teardown all run; the line says which scaling it measured."""
    env = _env()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--spp", "64", "--width", "256", "--scaling", scaling, "--no-peer-leg", "--no-cpu-baseline", "--no-other-configs"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
    out = json.loads(line)
    assert out["scaling"] == scaling and out["n_gpus"] == 1 and out["dtype"] == "f64" and out["value"] > 0
    assert "RCCL" in out["config"]["parallelism"]
    # (the peer leg, the CPU baseline and the other configs are switched off above: this test is about the group)
    assert out["roofline"]["bound"] == "valu_fp64" and 0 < out["roofline"]["frac"] < 1
