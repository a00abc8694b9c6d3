"""bench.py's contract is ONE JSON line on stdout; libraries underneath (RCCL's version banner) print there too, so bench.py
moves file descriptor 1 to stderr and writes its line to a private duplicate of the original stdout."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_only_the_result_line_reaches_stdout():
    code = (
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "bench.capture_stdout()\n"
        "print('python-level noise')\n"
        "os.write(1, b'C-level noise, e.g. the RCCL banner\\n')\n"
        "bench.emit_line('{\"value\": 1}')\n"
    )
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout == '{"value": 1}\n'
    assert "python-level noise" in out.stderr and "C-level noise" in out.stderr


def test_plain_invocation_with_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus N` invoked plainly (as the driver does) must start the N ranks itself and print ONE JSON line
    with n_gpus = N. --dry-run keeps this CPU test off the GPU: the ranks rendezvous (gloo), split the rows and reduce
    packed keys with the product's sharding arithmetic, nothing is scanned and `value` is null."""
    import json

    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run", "--batch", "64"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True and rec["value"] is None
    assert rec["config"]["exchange_ok"] is True
    assert rec["metric"].startswith("query-vectors/sec")


def test_gpus_flag_must_match_the_launcher():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr
