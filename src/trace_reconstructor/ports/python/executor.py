#!/usr/bin/env python3
"""Where the reference keeps its executor (src/trace_reconstructor/ports/python/executor.py): the experiment drivers
(`exps/exp*/run_experiment.sh`) call this path with the reference's flags, relative to the tree root.  This file only
forwards to `traceweaver_amd.executor` (same flags, same result files; see its docstring for what is served).

`--relative_path` is resolved against the root of this tree -- put (or link) `data/` and `exps/` next to `src/` --
or against $TRACEWEAVER_ROOT when that is set (e.g. an untouched checkout of the reference).
"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", ".."))
sys.path.insert(0, ROOT)

from traceweaver_amd import executor  # noqa: E402

if __name__ == "__main__":
    argv = sys.argv[1:]
    if "--project_root" not in argv:
        argv += ["--project_root", os.environ.get("TRACEWEAVER_ROOT", ROOT)]
    executor.main(argv)
