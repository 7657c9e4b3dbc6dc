"""GPU: RCCL runs once.  A fresh child process creates a REAL process group of one rank on backend "nccl"
(= RCCL on ROCm) and drives the k-cut data-parallel step of bench.py - graph | all_reduce(async) on RCCL's stream | graph | ... |
clip + Adam - for 20 iterations next to the unsplit step; see tests/rccl_world1_step.py for what it asserts.  A one-GPU box cannot
produce more multi-GPU evidence than this: it executes every collective call, the stream ordering between graph replays and
RCCL's stream, the capture of the plan's segments while a process group is alive, and the broadcast at start-up."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _env(port_off):
    return dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", HRP_DIST_WORLD1="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
                MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + (os.getpid() + port_off) % 2000))


def test_k_cut_step_over_a_real_rccl_group_of_one_rank():
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_world1_step.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900, env=_env(0))
    assert r.returncode == 0, r.stdout[-4000:]
    assert "rccl world-1: ok" in r.stdout and "backend nccl" in r.stdout, r.stdout[-2000:]


def test_bench_line_under_the_world1_switch_reports_nccl_and_the_overlap():
    """bench.py with HRP_DIST_WORLD1=1: the line says backend nccl and ar_overlap.enabled (the k-cut step passed its self-check on
    this machine); B = 8 keeps the run short - the number is not a benchmark."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "8", "--steps", "5", "--warmup", "2", "--no-extra",
                        "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=_env(1))
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["backend"] == "nccl" and line["config"]["rccl_ranks"] == 1, line["config"]
    assert line["ar_overlap"]["enabled"] is True and len(line["ar_overlap"]["cuts"]) >= 2, line.get("ar_overlap")
    assert line["n_gpus"] == 1 and line["value"] > 0
