"""`python bench.py --gpus N` must work when invoked plainly (the driver does exactly that): the script spawns its own
ranks before touching a GPU and passes rank 0's line through.  Rehearsed here on CPU (`--dry-run`: gloo, no kernels) through
the same launcher code, and once more under torch.distributed.run, the way the driver starts multi-GPU runs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    return p.returncode, lines, p.stderr.decode()[-2000:]


@pytest.mark.parametrize("n", [1, 2])
def test_plain_invocation_spawns_its_ranks(n):
    rc, lines, err = _run([sys.executable, "bench.py", "--gpus", str(n), "--dry-run", "--steps", "3", "--warmup", "1"])
    assert rc == 0, err
    assert len(lines) == 1, (lines, err)
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["value"] > 0 and rec["higher_is_better"] is True


def test_eight_ranks_agree_on_every_capture_call():
    """BASELINE config 4's rank count (the driver's 8-GPU scaling run): `bench.py --gpus 8 --dry-run` rehearses the launcher, the
    barrier / MAX timing and -- with a real collective in every step -- dmel_amd.GraphedStep's re-capture decisions on eight gloo
    ranks whose pictures of lambd arrive 0 / 5 / 2 / 7 executions late: identical capture calls everywhere, every forward covered"""
    rc, lines, err = _run([sys.executable, "bench.py", "--gpus", "8", "--dry-run", "--steps", "4", "--warmup", "1"], timeout=600)
    assert rc == 0, err
    assert len(lines) == 1, (lines, err)
    rec = json.loads(lines[0])
    reh = rec["config"]["graphed_step_rehearsal"]
    assert rec["n_gpus"] == 8 and reh["same_on_every_rank"] is True and len(reh["capture_calls"]) >= 3
    assert reh["collectives"] >= reh["calls"] * reh["steps_per_replay"]


def test_under_torch_distributed_run():
    rc, lines, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", "29611", "bench.py", "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "0"])
    assert rc == 0, err
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2, (lines, err)


def test_a_failing_rank_fails_the_launcher():
    rc, lines, err = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"])     # no GPU here: the ranks assert
    assert rc != 0 and not lines
