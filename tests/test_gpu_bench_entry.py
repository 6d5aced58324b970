"""The driver's multi-GPU entry point, started the way the driver starts it: `python bench.py --gpus 2 ...` as a FRESH child process
(bench.py launches its own ranks through torch.distributed.run before anything touches the GPU).  One-GPU boxes cannot run RCCL with two
ranks (it refuses duplicate devices), so the ranks talk gloo (SC_DIST_BACKEND): the code path - self-launch, rendezvous on 127.0.0.1, rank-major
shards, embedding all-gather, row-sharded loss head with its statistics exchange, bucketed SUM all-reduces, barrier + MAX-over-ranks timing, rank 0's
JSON line - is the one the 8-GPU job runs; only the transport differs."""
import json
import math
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(global_batch):
    env = dict(os.environ, SC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--model", "test-small",
           "--global-batch", str(global_batch), "--local-batch", "64", "--cpu-baseline", "0"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=540)
    assert p.returncode == 0, f"bench.py --gpus 2 exited {p.returncode}\n{p.stderr[-3000:]}"
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("global_batch", [128, 512])
def test_bench_two_ranks_as_child_process(global_batch):
    """global batch 128: 64 pairs per rank, the plain step.  512: 256 pairs per rank, run by four micro-batches of 64 through
    Trainer.step_cached (embeddings gathered once, ONE loss head over the 512 pairs, every gradient bucket all-reduced once)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a, b = _bench(global_batch), _bench(global_batch)
    for out in (a, b):
        assert out["metric"] == "image-text pairs/sec" and out["unit"] == "pairs/s" and out["higher_is_better"] is True
        assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 1 and out["warmup"] == 1
        cfg = out["config"]
        assert cfg["global_batch"] == global_batch and cfg["local_batch"] == global_batch // 2 and cfg["micro_batch"] == 64 and cfg["parallelism"] == "dp2"
        assert math.isfinite(out["last_loss"]) and out["value"] > 0 and out["ms_per_step"] > 0
        assert abs(out["value"] - global_batch / (out["ms_per_step"] * 1e-3)) <= 0.01 * out["value"]      # whole-job pairs/s, not per GPU
        assert "roofline" in out and "cpu_baseline" not in out      # the CPU baseline is an N = 1 leg
    assert a["last_loss"] == b["last_loss"], "the two-rank step is not reproducible run to run"
