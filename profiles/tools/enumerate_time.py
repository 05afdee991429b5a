"""Enumeration time per pass (HIP events of the engine, minimum of three steps) for one or more builds of the library on the
default workload (media shape, 25.6 M spans) or the Alibaba-shape slice: what the variant sweeps of round 3 were measured with
(per-thread limit, staging width, register caps: DESIGN.md 12).  No torch.

    python profiles/tools/enumerate_time.py default scratch/variants/lm96.so          # "default" = traceweaver_amd/lib/libtwgpu.so
    WL=alibaba python profiles/tools/enumerate_time.py default scratch/variants/x.so
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from traceweaver_amd import synth
from traceweaver_amd.engine import Engine

wl = os.environ.get("WL", "media")
libs = sys.argv[1:] or ["default"]
if wl == "alibaba":
    import bench

    sys.argv = ["bench.py", "--workload", "alibaba"]
    units, truth, _ = bench.make_units(bench.parse_args(), 1000)
else:
    units, truth = synth.make_workload(1000, 100000, services=synth.MEDIA_SERVICES, replicas=16, concurrency=1.6)
for lib in libs:
    eng = Engine(0, lib_path=None if lib == "default" else lib)
    eng.load(units)
    e1, e2, h1, h2, p1, p2 = [], [], [], [], [], []
    for _ in range(3):
        eng.run_pass1()
        t = eng.timing()
        e1.append(t["enumerate"]); h1.append(t["host_enum_submit"]); p1.append((t["pass"], t["host_pass"]))
        eng.fit_mixtures()
        eng.run_pass2()
        t = eng.timing()
        e2.append(t["enumerate"]); h2.append(t["host_enum_submit"]); p2.append((t["pass"], t["host_pass"]))
    print(wl, lib, "enumerate ms pass1 %.2f pass2 %.2f; host submit of the enumeration %.2f / %.2f ms; pass (events, host wall) %s / %s" % (
        min(e1), min(e2), min(h1), min(h2), "%.2f, %.2f" % min(p1), "%.2f, %.2f" % min(p2)))
    eng.close()
