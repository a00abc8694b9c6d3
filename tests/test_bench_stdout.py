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
