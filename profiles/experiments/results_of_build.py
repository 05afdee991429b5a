"""Results of one build (parents, top-5 lists, scores, choices) on media / media at concurrency 4 / nodejs shapes, pickled:
run once per build (TW_LIB = library, default the host-emulation build) and compare the pickles -- the three patches of this
directory leave every array identical to HEAD (checked on the emulation build)."""
import sys, os, pickle
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')]
os.environ["TW_TILE"]=os.environ.get("TW_TILE","1"); os.environ["TW_COOP_THREADS"]=os.environ.get("TW_COOP_THREADS","1")
import numpy as np
from traceweaver_amd import synth
from traceweaver_amd.engine import Engine
out=sys.argv[1]
res={}
for name,(svc,kw) in {"media":(synth.MEDIA_SERVICES,dict(concurrency=1.6)),"media_c4":(synth.MEDIA_SERVICES,dict(concurrency=4.0)),"nodejs":(synth.NODEJS_SERVICES,dict(concurrency=4.0,granularity_us=1000,mean_service_us=6000.0,gap_us=1500.0))}.items():
    units,truth=synth.make_workload(7, 2500, services=svc, replicas=1, **kw)
    eng=Engine(0, lib_path=os.environ.get('TW_LIB', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'hostemu', '_build', 'libtwgpu_emu.so')))
    eng.load(units); eng.set_truth(truth)
    eng.run_pass1(); eng.fit_mixtures(); eng.run_pass2()
    r=eng.results(2)
    res[name]=[(x["parent"].copy(), x["topk_idx"].copy(), x["topk_score"].copy(), x["chosen"].copy()) for x in r]
    print(name, [round(float(e["accuracy"]),4) for e in eng.evaluate()])
    eng.close()
pickle.dump(res, open(out,"wb"))
