"""`python bench.py --gpus N` as the driver runs it (no torchrun around it, no WORLD_SIZE): the process is a launcher that
never touches the GPU, starts N rank processes through torch.distributed.run and relays rank 0's single JSON line.
Runs here on CPU with `--dry-run-ranks` (rendezvous + one gloo all-reduce of ones, no workload, no GPU call)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.update(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", **extra)   # a parent that needed a GPU would fail here
    return env


@pytest.mark.timeout(300)
def test_plain_invocation_starts_its_own_ranks_and_relays_one_line():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-ranks"], capture_output=True, text=True,
                         env=_env(), timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout                      # ONE line on stdout, everything else went to stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo"
    assert d["gpu_initialised"] is False


@pytest.mark.timeout(300)
def test_launcher_passes_the_ranks_failure_on():
    # an unknown flag makes every rank exit non-zero in argparse: the launcher must not print a line and must fail
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-ranks", "--no-such-flag"],
                         capture_output=True, text=True, env=_env(), timeout=280)
    assert res.returncode != 0
    assert res.stdout.strip() == ""


def test_rank_count_mismatch_is_refused_by_a_launched_rank():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-run-ranks"], capture_output=True, text=True,
                         env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=120)
    # WORLD_SIZE is set: this process is a rank, not a launcher, and its launcher started the wrong number of ranks
    assert res.returncode != 0 and "different number of ranks" in res.stderr
    assert res.stdout.strip() == ""
