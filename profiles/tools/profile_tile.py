"""Phase times of k_enumerate_tile<E> (library built with -DTW_PROFILE_TILE -DTW_PROFILE_TILE_E=<E>, path in TW_PROFILE_LIB):
10 ns ticks of thread 0 of every workgroup of that class, per pass.  WL=alibaba: the Alibaba-shape slice, else the media shape."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from traceweaver_amd import synth
from traceweaver_amd.engine import Engine

if os.environ.get("WL") == "alibaba":
    import bench
    sys.argv = ["bench.py", "--workload", "alibaba"]
    units, truth, _ = bench.make_units(bench.parse_args(), 1000)
else:
    units, truth = synth.make_workload(1000, 100000, services=synth.MEDIA_SERVICES, replicas=int(os.environ.get("TW_REPLICAS", "16")), concurrency=1.6)
eng = Engine(0, lib_path=os.environ.get("TW_PROFILE_LIB", "scratch/variants/proftile6.so"))
eng.load(units)
lib = eng._lib
lib.tw_debug_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
prev = np.zeros(16, dtype=np.uint64)
names = ["cut-offs", "slices", "masks/lists/scans", "terms", "tuples", "ranks", "results"]
for pass_no in (1, 2):
    if pass_no == 1:
        eng.run_pass1()
    else:
        eng.fit_mixtures(); eng.run_pass2()
    cur = np.zeros(16, dtype=np.uint64); lib.tw_debug_profile(eng._h, ctypes.c_void_p(cur.ctypes.data))
    a = (cur - prev).astype(np.float64); prev = cur.copy()
    wg = max(a[8], 1.0)
    print("pass %d: %d workgroups, mean lifetime %.1f us, %.1f segments, %.0f items, %.0f tuple slots per workgroup" % (pass_no, a[8], a[7] / wg / 100.0, a[9] / wg, a[10] / wg, a[11] / wg))
    print("   per workgroup (us): " + ", ".join("%s %.1f" % (n, a[k] / wg / 100.0) for k, n in enumerate(names)))
