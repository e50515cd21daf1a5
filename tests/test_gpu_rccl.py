"""RCCL is executed ONCE on the one GPU of the test box (missing 2): a fresh child process builds a one-rank
"nccl" process group on cuda:0 and runs the package's collectives and bench.py's on device tensors, so that the library load,
the device_id= initialisation and the dtype handling (complex64 as its real view, float32 records + int64 ids in one int32
block, float64 sums) are proven before the first multi-GPU run.  The pytest process itself is never re-exec'd; the children
initialise the GPU themselves."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_one_rank_rccl_runs_the_packages_collectives_on_device_tensors():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py")], env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    rec = _json_line(p.stdout)
    assert rec["backend"] == "nccl" and rec["rccl_loaded"]
    assert rec["gather_equal"] and rec["all_gather_equal"] and rec["broadcast_equal"] and rec["bench_gather_equal"]


@pytest.mark.parametrize("workload,frames", [("cfg2_2x2_eva_ls_mmse", 512), ("cfg5_4x4_2048_fullcov", 64)])
def test_bench_rehearses_its_collectives_through_one_rank_rccl(workload, frames):
    """bench.py --rccl-rehearsal: the measured job with its N > 1 control flow switched on at N = 1 (process group, barrier,
    gather of the metric sums, MAX-reduction of the time; config 5 also broadcasts the prepared filter W)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rccl-rehearsal", "--workload", workload, "--frames", str(frames),
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(p.stdout.strip().splitlines()) == 1, p.stdout[:2000]      # RCCL's version banner and the like must not reach stdout
    rec = _json_line(p.stdout)
    assert rec["n_gpus"] == 1 and rec["config"]["collectives"] == "executed (nccl, world 1)"
    assert rec["value"] > 0 and rec["est_mse"]["mmse"] > 0
