"""bench.py --gpus N without a launcher starts N ranks itself (before touching the GPU) and fails loudly when a rank dies.
CPU only: the hidden --selftest-spawn flag replaces the measured job by a gloo all_reduce."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_plain_shell_launch_runs_n_ranks_and_prints_one_line():
    p = _run(["--gpus", "3", "--selftest-spawn"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]      # gloo itself chats on stdout
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 3 and rec["sum"] == 6.0          # 1 + 2 + 3: all three ranks took part in the collective


def test_a_dying_rank_fails_the_job():
    p = _run(["--gpus", "2", "--selftest-spawn"], {"BENCH_SELFTEST_FAIL_RANK": "1"})
    assert p.returncode != 0
    assert "rank 1 exited with code 7" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]      # no JSON line from a failed job


def test_gpus_must_match_the_launcher_world_size():
    p = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert p.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in (p.stderr + p.stdout)
