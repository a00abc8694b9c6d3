"""bench.py on the GPU box: the N > 1 launch path invoked plainly (two ranks sharing GPU 0, keys exchanged through gloo --
the in-library RCCL exchange cannot put two ranks on one device), and the one-rank run through the sharded handle and its
RCCL communicator (--force-dist) with logical shards."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_started_by_bench_itself():
    rec = run_bench("--gpus", "2", "--backend", "gloo", "--rows", "140625", "--batch", "256", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0",
                    "--no-extras", "--no-pmc")
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["steps"] == 2
    assert rec["config"]["identical_keys_to_exact_scan"] is True and rec["config"]["planted_queries_found"] is True
    assert rec["config"]["path"] == "mfma"                       # 78125 rows per rank, 256 queries: the default dispatch
    assert "k_scan_l2" in rec["roofline"]["kernel"]
    assert rec["roofline"]["achieved"] > 0 and rec["roofline_mfma"]["achieved_tflops"] > 0


def test_one_rank_through_the_sharded_handle_and_rccl():
    rec = run_bench("--gpus", "1", "--force-dist", "--shards-per-device", "2", "--rows", "250000", "--batch", "512", "--steps", "2", "--warmup", "1",
                    "--cpu-seconds", "0", "--no-extras", "--no-pmc")
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    assert "ncclAllReduce" in rec["config"]["key_exchange"]
    assert rec["config"]["exchange_us_per_step"] is not None and rec["config"]["exchange_us_per_step"] >= 0
    assert rec["config"]["identical_keys_to_exact_scan"] is True and rec["config"]["planted_queries_found"] is True
