import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from traceweaver_amd import synth
from traceweaver_amd.engine import Engine
units, truth = synth.make_workload(1000, 100000, services=synth.MEDIA_SERVICES, replicas=4, concurrency=1.6)
spans = sum(u.n_spans for u in units)
eng = Engine(0)
for it in range(3):
    t0 = time.perf_counter(); eng.load(units); t1 = time.perf_counter()
    eng.run_pass1(); eng.fit_mixtures(); eng.run_pass2(); t2 = time.perf_counter()
    r = eng.results(2, fields=("parent",)); t3 = time.perf_counter()
    print("load (alloc + H2D + descriptors) %.1f ms | compute %.1f ms | parents D2H %.1f ms | spans/s incl. both: %.3g" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, spans/(t3-t0)))
