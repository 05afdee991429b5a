import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from traceweaver_amd import synth
from traceweaver_amd.engine import Engine
conc = float(os.environ.get("TW_CONC", "1.6")); n_in = int(os.environ.get("TW_NIN", "100000"))
if os.environ.get("TW_WORKLOAD") == "nodejs": units, truth = synth.make_nodejs_workload(1000, n_in, concurrency=conc, replicas=4)
else: units, truth = synth.make_workload(1000, n_in, services=synth.MEDIA_SERVICES, replicas=4, concurrency=conc)
eng = Engine(0, lib_path=os.environ.get("TW_PROFILE_LIB", "scratch/profsel.so")); eng.load(units)
lib = eng._lib
lib.tw_debug_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
def report(pass_no, t, a, out):
  names = ["0 load cands", "1 adjacency", "2 components", "3 comp setup", "4 brute", "5 dfs path", "6 loop overhead", "7 write"]
  tot = float(a[:8].sum())
  print("pass", pass_no, "select ms", t["select"], "windows", out[0])
  for k in range(8): print("%-16s %6.1f%%  per window %8.0f ticks (10 ns)" % (names[k], 100.0*a[k]/tot, a[k]/max(out[0],1)))
  print("   longest window: search %.1f us, of it matching bound %.1f us in %d calls, %d nodes, E=%d" % (int(cur[10]) / 100.0, int(cur[11]) / 100.0, int(cur[12]), int(cur[13]), int(cur[14])))
  print("longest window: %.1f us, m=%d, search nodes / 16 = %d ; sum of window times / 1792 WGs = %.2f ms" % ((int(a[8]) >> 24) / 100.0, (int(a[8]) >> 16) & 0xff, int(a[8]) & 0xffff, int(a[9]) / 100.0 / 1792 / 1000.0))
prev = np.zeros(16, dtype=np.uint64)
for pass_no in (1, 2):
  if pass_no == 1: eng.run_pass1()
  else: eng.fit_mixtures(); eng.run_pass2()
  t = eng.timing()
  cur = np.zeros(16, dtype=np.uint64); lib.tw_debug_profile(eng._h, ctypes.c_void_p(cur.ctypes.data))
  a = cur - prev; a[8] = cur[8]; prev = cur
  out = np.zeros(16, dtype=np.int32); lib.tw_debug_worklists(eng._h, ctypes.c_void_p(out.ctypes.data))
  report(pass_no, t, a, out)
  st = eng.results(pass_no, fields=("unit_stats",))
  print("   search nodes (all windows): %d" % sum(int(r["search_nodes"]) for r in st))
