"""Item / wavefront lifetimes of k_enumerate_heavy<4, 32> on the bench workload (library built with -DTW_PROFILE, path in
TW_PROFILE_LIB): is the kernel's tail one long item or an uneven split?"""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from traceweaver_amd import synth
from traceweaver_amd.engine import Engine
conc = float(os.environ.get("TW_CONC", "1.6")); n_in = int(os.environ.get("TW_NIN", "100000"))
if os.environ.get("WL") == "alibaba":   # the Alibaba-shape slice; build the library with -DTW_PROFILE_E=<class> (default 4)
    import bench
    sys.argv = ["bench.py", "--workload", "alibaba"]
    units, truth, _ = bench.make_units(bench.parse_args(), 1000)
else:
    units, truth = synth.make_workload(1000, n_in, services=synth.MEDIA_SERVICES, replicas=int(os.environ.get("TW_REPLICAS", "4")), concurrency=conc)
eng = Engine(0, lib_path=os.environ.get("TW_PROFILE_LIB", "scratch/prof.so")); eng.load(units)
lib = eng._lib
lib.tw_debug_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
prev = np.zeros(16, dtype=np.uint64)
hprev = np.zeros(16, dtype=np.uint64)
for pass_no in (1, 2):
    if pass_no == 1: eng.run_pass1()
    else: eng.fit_mixtures(); eng.run_pass2()
    t = eng.timing()
    cur = np.zeros(16, dtype=np.uint64); lib.tw_debug_profile(eng._h, ctypes.c_void_p(cur.ctypes.data))
    a = cur - prev; prev = cur.copy()
    print("pass %d enumerate group %.2f ms; E=4 narrow wavefront kernel: %d items, mean %.1f us, longest %.1f us (%d tuples), items >= 100 us: %d"
          % (pass_no, t["enumerate"], a[14], a[13] / max(a[14], 1) / 100.0, (int(cur[12]) >> 24) / 100.0, int(cur[12]) & 0xffffff, a[15]))
    print("   wavefronts with work %d, mean lifetime %.1f us, longest %.1f us; sum of item time / 4096 = %.1f us"
          % (a[9], a[8] / max(a[9], 1) / 100.0, int(cur[6]) / 100.0, a[13] / 100.0 / 4096))
    tot = float(a[:5].sum())
    print("   phases: " + ", ".join("%s %.0f%%" % (n, 100.0 * a[k] / tot) for k, n in enumerate(os.environ.get("TW_PHASE_NAMES", "stage,tables,tuple list,walk+top5,results").split(","))))
    print("   longest item: tuple list %.1f us, walk+top5 %.1f us" % (int(cur[7]) / 100.0, int(cur[10]) / 100.0))
    info = int(cur[5])
    print("   longest item: flags prune=%d replay=%d list=%d counted=%d log=%d list_scored=%d list_all=%d tables=%d, part %d of %d, staged candidates %s"
          % (info & 1, info >> 1 & 1, info >> 2 & 1, info >> 3 & 1, info >> 4 & 1, info >> 5 & 1, info >> 6 & 1, info >> 7 & 1, info >> 16 & 255, info >> 8 & 255,
             [info >> (24 + 5 * e) & 31 for e in range(8)]))
    if hasattr(lib, "tw_debug_profile_hist"):
        h = np.zeros(16, dtype=np.uint64); lib.tw_debug_profile_hist(eng._h, ctypes.c_void_p(h.ctypes.data))
        d = h - hprev; hprev = h.copy()
        print("   items by duration: " + ", ".join("%s %d" % (("<%.0f us" % (5.12 * 2 ** (b + 1))) if b < 15 else "more", int(d[b])) for b in range(16) if d[b]))
