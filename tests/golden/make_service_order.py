#!/usr/bin/env python3
"""Writes tests/golden/service_order.json: for every corpus the reference ships, the services in the order the reference's
executor hands them to the predictor (executor.py:1068-1080: the key order of `out_spans_by_process`, which the native
loader reproduces -- tests/test_ingest.py) -- the order in which a seeded reference run consumes numpy's global RNG.

TEST INFRASTRUCTURE ONLY; needs the reference's data directory, so it is run by hand in the build container (the GPU box
has none).  tests/test_gpu_parity.py::test_seeded_chain_on_every_corpus replays the frozen runs (tests/golden/ref_*.npz)
in this order without the data.

    python tests/golden/make_service_order.py [path of libtwgpu or of the host-emulation build]
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
REF_DATA = "/root/reference/data"


def main():
    from test_ingest import REFERENCE_CORPORA, reference_files
    from traceweaver_amd.ingest import REFERENCE_FIX, Corpus

    lib = sys.argv[1] if len(sys.argv) > 1 else None
    if lib is None:
        from tests.hostemu.build_emu import build

        lib = build()
    out = {}
    for name, rel, fix in REFERENCE_CORPORA:
        first_span, surgery = REFERENCE_FIX[fix]
        c = Corpus(lib_path=lib)
        c.add_files(reference_files(os.path.join(REF_DATA, rel)), first_span=first_span, max_traces=1001, fix=surgery)
        units, _, _ = c.units()
        out[name] = [u.service for u in units]
        c.close()
    with open(os.path.join(HERE, "service_order.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
