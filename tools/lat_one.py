import sys, time, gc
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
fir = ge.load_package()
rng = np.random.default_rng(1)
rows = rng.random((3030, 1536), dtype=np.float32)
q = rng.random((1, 1536), dtype=np.float32)
g = fir.Gallery(rows, None, 0, 0)
for _ in range(200): g.search_top1(q)
gc.disable()
ts = []
for _ in range(2000):
    t0 = time.perf_counter(); g.search_top1(q); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
print("median %.1f us  p10 %.1f  p90 %.1f" % (np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90)))
