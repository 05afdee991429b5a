import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from traceweaver_amd import synth
from traceweaver_amd.engine import Engine
units, truth = synth.make_workload(1000, 100000, services=["par4"], replicas=4, concurrency=1.6)
eng = Engine(0, lib_path=os.environ.get("TW_PROFILE_LIB", "scratch/prof.so")); eng.load(units)
lib = eng._lib
lib.tw_debug_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
def read():
    a = np.zeros(16, dtype=np.uint64); lib.tw_debug_profile(eng._h, ctypes.c_void_p(a.ctypes.data)); return a
eng.run_pass1(); t = eng.timing()
a = read()
names = ["a0 load lo/hi", "a1 stage+tab", "a2 score", "a3 insert+sync", "a4 write", "items", "max wave total", "fetch", "lifetime sum", "waves"]
tot = float(a[0]+a[1]+a[2]+a[3]+a[4]+a[7])
print("pass1 enum ms", t["enumerate"])
for k in (0,1,2,3,4,7):
    print("%-16s %6.1f%%  per item %8.0f ticks" % (names[k], 100.0*a[k]/tot, a[k]/max(a[5],1)))
print("items", a[5], "waves", a[9], "lifetime/wave ticks", a[8]/max(a[9],1), "sum phases/wave", tot/max(a[9],1))
