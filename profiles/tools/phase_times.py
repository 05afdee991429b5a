import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from traceweaver_amd import synth
from traceweaver_amd.engine import Engine
n_in = int(os.environ.get("N_IN", "100000")); reps = int(os.environ.get("REPS", "4"))
units, truth = synth.make_workload(1000, n_in, services=synth.MEDIA_SERVICES, replicas=reps, concurrency=1.6)
spans = sum(u.n_spans for u in units)
for lib in sys.argv[1:]:
    eng = Engine(0, lib_path=None if lib == "default" else lib); eng.load(units)
    rows = []
    for it in range(4):
        t0 = time.perf_counter(); eng.run_pass1(); t1 = eng.timing(); eng.fit_mixtures(); eng.run_pass2(); t2 = eng.timing(); dt = time.perf_counter() - t0
        rows.append((dt * 1e3, t1, t2))
    dt, t1, t2 = rows[-1]
    print("%-40s TILE=%s step %.2f ms | p1: pass %.2f enum %.2f sel %.2f win %.2f rep %.2f par %.2f | p2: pass %.2f enum %.2f sel %.2f | fit %.2f" % (
        os.path.basename(lib), os.environ.get("TW_TILE", "-"), dt, t1["pass"], t1["enumerate"], t1["select"], t1.get("windows", -1), t1.get("repair", -1), t1.get("params", -1),
        t2["pass"], t2["enumerate"], t2["select"], t2["fit"]))
    eng.close()
import ctypes
eng = Engine(0); eng.load(units)
out = np.zeros(16, dtype=np.int32)
eng.run_pass1(); eng._lib.tw_debug_worklists(eng._h, ctypes.c_void_p(out.ctypes.data)); print("pass1 worklists: select windows", out[0], "heavy spans by E", out[1:].tolist())
eng.fit_mixtures(); eng.run_pass2(); eng._lib.tw_debug_worklists(eng._h, ctypes.c_void_p(out.ctypes.data)); print("pass2 worklists: select windows", out[0], "heavy spans by E", out[1:].tolist())
r = eng.results(2, fields=("unit_stats",)); print("windows total", sum(int(x["n_windows"]) for x in r), "in-spans", sum(u.n_in for u in units))
