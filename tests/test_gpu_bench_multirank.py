"""GPU: bench.py's multi-rank code path (rendezvous, barriers, max-over-ranks timing, per-rank kernel times, the final
gather of the strong-scaling mode) with two ranks on the one GPU of the test box.  RCCL refuses two ranks on one device,
so the collectives go through gloo here (PAI_BENCH_BACKEND=gloo); everything else is the path `python bench.py --gpus N`
(self-launching) or `torchrun ... bench.py --gpus N` takes on an 8-GPU node."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_two_ranks_on_one_gpu(scaling):
    """`python bench.py --gpus 2` with no torchrun around it: bench.py starts the two ranks itself."""
    env = dict(os.environ, PAI_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "6000",
           "--no-cpu-baseline", "--no-extras"] + (["--scaling", "weak"] if scaling == "weak" else [])     # strong is the default
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == scaling and line["parity_checked"] is True
    assert line["metric"] == json.loads((ROOT / "BASELINE.json").read_text())["metric"]
    assert sorted(r["rank"] for r in line["ranks_seen"]) == [0, 1] and line["collective_backend"] == "gloo"
    assert len(line["per_rank_kernel_ms"]) == 2 and all("k_dec_a" in k for k in line["per_rank_kernel_ms"])
    if scaling == "strong":
        assert line["config"]["batch_total"] == 6000 and line["gather_ms"] is not None
        other = line["weak_scaling"]
        assert other["scaling"] == "weak" and other["batch_total"] == 12000
    else:
        assert line["config"]["batch_total"] == 12000 and line["gather_ms"] is None
        other = line["strong_scaling"]
        assert other["scaling"] == "strong" and other["batch_total"] == 6000
    assert line["value"] > 0 and other["value"] > 0 and other["parity_checked"] is True


@pytest.mark.parametrize("config,scaling,batch", [("headline", "strong", 1003), ("cfg5", "weak", 40)])
def test_bench_eight_ranks_on_one_gpu(config, scaling, batch):
    """The 8-rank arrangement of BASELINE configs[3]/[4] as a dry run on the one GPU (gloo collectives): eight processes,
    each building its own key handles and fixed-base table, a ragged strong-scaling split (1003 = 7 x 126 + 121) of the
    2048-bit configuration and the weak arrangement of the 4096-bit one (--config cfg5), rendezvous / barriers /
    max-over-ranks timing / parity reduction / final gather included."""
    env = dict(os.environ, PAI_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--batch", str(batch),
           "--config", config, "--scaling", scaling, "--no-cpu-baseline", "--no-extras"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["scaling"] == scaling and line["parity_checked"] is True
    assert sorted(r["rank"] for r in line["ranks_seen"]) == list(range(8))
    assert len(line["per_rank_kernel_ms"]) == 8
    assert line["config"]["key_bits"] == {"headline": 2048, "cfg5": 4096}[config] and line["config"]["baseline_config"] == config
    if scaling == "strong":
        assert line["config"]["batch_total"] == batch and line["gather_ms"] is not None
        assert line["weak_scaling"]["batch_total"] == 8 * batch and line["weak_scaling"]["parity_checked"] is True
    else:
        assert line["config"]["batch_total"] == 8 * batch
        assert line["strong_scaling"]["batch_total"] == batch and line["strong_scaling"]["parity_checked"] is True
    assert line["roofline"]["frac"] is not None and line["value"] > 0


def test_bench_single_gpu_config_lines_carry_roofline_and_cpu_baseline():
    """`python bench.py --config cfg5` (and cfg4) on one GPU: a line with roofline + cpu_baseline for the 4096 / 3072-bit
    configurations (small batch here; the driver runs the full sizes)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PAI_BENCH_BACKEND")}
    for config, bits, nl, prof_batch in (("cfg5", 4096, 72, 1 << 18), ("cfg4", 3072, 56, 1 << 20)):
        cmd = [sys.executable, str(ROOT / "bench.py"), "--config", config, "--batch", "4096", "--steps", "1", "--warmup", "1",
               "--no-extras", "--cpu-seconds", "2"]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
        assert res.returncode == 0, res.stderr[-3000:]
        line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
        assert line["config"]["key_bits"] == bits and f"{bits}-bit key" in line["metric"] and line["parity_checked"] is True
        rf, cpu = line["roofline"], line["cpu_baseline"]
        assert f"k_dec_a_padic<{nl}>" in rf["kernel"] and 0 < rf["frac"] < 1.2 and rf["peak_sustained"] < rf["peak"]
        assert rf["traffic"] is not None and rf["traffic_source"]["profile_batch"] == prof_batch
        assert cpu["value"] > 0 and cpu["cores"] >= 1 and "port" in cpu["kind"]


def test_bench_refuses_more_ranks_than_gpus():
    """RCCL needs one device per rank: asking for more GPUs than the box has must fail loudly, not run fewer ranks."""
    import torch

    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PAI_BENCH_BACKEND")}
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n), "--steps", "1", "--no-extras",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env, cwd=str(ROOT))
    assert res.returncode != 0 and "GPU(s) are visible" in res.stderr
