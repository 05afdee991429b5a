"""End to end from JSON files at a larger size than bench.py's default legs (one GPU): `kind:n_traces` pairs.

    python profiles/tools/e2e_scale.py alibaba:60000 hotel:60000 > gpurun_out/e2e_scale.json
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def main():
    out = {"host_cores": os.cpu_count(), "runs": []}
    for spec in sys.argv[1:] or ["alibaba:60000"]:
        kind, n = spec.split(":")
        r = bench.end_to_end(0, kind=kind, n_traces=int(n))
        r["kind"] = kind
        out["runs"].append(r)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
