"""Does this library leave libc's rand() stream alone? (The HIP runtime's start-up draws from it; fir_runtime_init_ gives
that start-up a private state.) usage: python tools/rand_probe.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
libc = ctypes.CDLL("libc.so.6")
libc.srand(5); ref = [libc.rand() for _ in range(3)]
libc.srand(5)
import __graft_entry__ as ge
fir = ge.load_package()          # imports torch, loads the library
after_import = libc.rand()
libc.srand(5)
rows = np.random.default_rng(1).random((3000, 64), dtype=np.float32)
g = fir.Gallery(rows, None, 0, 0)        # HIP initialisation happens here
after_create = libc.rand()
libc.srand(5)
g.search_top1(rows[:3])
after_search = libc.rand()
libc.srand(5)
g2 = fir.Gallery(rows, None, 0, 0); g2.search_top1(rows[:9]); g2.close()
after_second = libc.rand()
print("reference first rand:", ref[0])
print("after import        :", after_import, after_import == ref[0])
print("after first create  :", after_create, after_create == ref[0])
print("after search        :", after_search, after_search == ref[0])
print("after second gallery:", after_second, after_second == ref[0])
