"""GPU: bench.py's multi-rank code path (rendezvous, barriers, max-over-ranks timing, per-rank kernel times, the final
gather of the strong-scaling mode) with two ranks on the one GPU of the test box.  RCCL refuses two ranks on one device,
so the collectives go through gloo here (PAI_BENCH_BACKEND=gloo); everything else is the path `torchrun ... bench.py
--gpus N` takes on an 8-GPU node."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_on_one_gpu(scaling):
    env = dict(os.environ, PAI_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29700 + os.getpid() % 200 + (1 if scaling == "strong" else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "6000",
           "--scaling", scaling, "--no-cpu-baseline", "--no-extras"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == scaling and line["parity_checked"] is True
    assert len(line["per_rank_kernel_ms"]) == 2 and all("k_dec_a" in k for k in line["per_rank_kernel_ms"])
    if scaling == "strong":
        assert line["config"]["batch_total"] == 6000 and line["gather_ms"] is not None
    else:
        assert line["config"]["batch_total"] == 12000 and line["gather_ms"] is None
    assert line["value"] > 0
