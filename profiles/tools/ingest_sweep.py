"""Parser-thread sweep of the native loader on the machine at hand (host work only; run on the GPU box through gpurun to
pick the default thread count of tw_corpus_add_files for its host): hotel-shape and alibaba-shape corpora, several
repetitions per thread count (the first ones after an idle period are slow on virtual machines whose idle vCPUs are parked).

    python profiles/tools/ingest_sweep.py [n_traces] [threads,threads,...] > gpurun_out/ingest_sweep.json
"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from traceweaver_amd.ingest import Corpus  # noqa: E402


def main():
    n_traces = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    threads = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,4,8,16,32,64").split(",")]
    out = {"host_cores": os.cpu_count(), "n_traces": n_traces, "shapes": {}}
    for kind in ("hotel", "alibaba"):
        with tempfile.TemporaryDirectory() as d:
            t0 = time.perf_counter()
            paths, fix = bench._corpus(kind, d, n_traces)
            rows = {"write_s": time.perf_counter() - t0}
            for th in threads:
                runs = []
                for _ in range(5):
                    c = Corpus()
                    t0 = time.perf_counter()
                    counts = c.add_files(paths, first_span=None, max_traces=0, threads=th, fix=fix)
                    t1 = time.perf_counter()
                    c.units()
                    runs.append((t1 - t0, time.perf_counter() - t1))
                    c.close()
                best = min(a + b for a, b in runs)
                rows[str(th)] = {"spans": counts["spans"], "add_files_s": [round(a, 4) for a, _ in runs], "units_s": [round(b, 4) for _, b in runs],
                                 "best_spans_per_s": counts["spans"] / best}
            out["shapes"][kind] = rows
    print(json.dumps(out))


if __name__ == "__main__":
    main()
