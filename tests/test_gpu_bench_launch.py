"""`python bench.py --gpus 2` run PLAINLY on the GPU box: the script starts its own two ranks (both on the one GPU there,
gloo between them - RCCL refuses two ranks on one device), rank 0's single JSON line comes back through the parent with
n_gpus = 2, BASELINE config 5's batched-file leg over 2 x 1250 files bit-exact against the compiled reference's hashes,
the CPU baseline on the N > 1 line too."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_means_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "5", "--warmup", "2",
                        "--no-saturated", "--no-extras"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["scaling"] == "weak"
    assert line["bit_exact_vs_reference_golden"] is True
    c5 = line["config5"]
    assert c5["files"] == 2500 and c5["backend"] == "gloo" and c5["bit_exact_vs_reference_golden"] is True
    cpu = line["cpu_baseline"]
    assert cpu["cores"] == 1 and cpu["value"] > 0 and cpu["kind"] in ("reference", "port")
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
    # two steps in flight on every rank, both pipelines' outputs identical, rounds 2-3's one-pipeline figure beside the headline
    assert line["config"]["steps_in_flight"] == 2 and line["outputs_identical_across_pipelines"] is True
    assert line["one_pipeline"]["value"] > 0 and line["serial"]["value"] > 0


def test_bench_refuses_more_rccl_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--no-saturated", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert "needs %d GPUs" % n in p.stderr
