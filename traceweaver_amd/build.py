"""Builds traceweaver_amd/lib/libtwgpu.so with hipcc for gfx950 (in-tree, so that the prebuilt library
travels with the repository snapshot to the GPU box)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libtwgpu.so")
SOURCES = ["tw_engine.hip", "tw_kernels.h", "tw_tile.h", "tw_lean.h", "tw_device.h", "tw_fit.h", "tw_eval.h", "tw_skip.h", "tw_load.h", "tw_baselines.h", "tw_ingest.cpp"]

# -ffp-contract=off: scores are chains of plain IEEE double operations in the reference's order;
# an FMA would change the last bit and with it the resolution of exact ties (DESIGN.md "Scores").
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread",
         "-DTW_FIT_SEED_ATTR=__attribute__((amdgpu_waves_per_eu(4)))"]


def source_digest():
    """sha256 over the kernel / engine sources and the ABI header: profiles record it (profiles/summarize.py) so that
    bench.py only quotes HBM-traffic counters taken from exactly the code that is running."""
    import hashlib

    h = hashlib.sha256()
    for d in [os.path.join(SRC, f) for f in sorted(SOURCES)] + [os.path.join(REPO, "include", "traceweaver_amd.h")]:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(OUT):
        return True
    deps = [os.path.join(SRC, f) for f in SOURCES] + [os.path.join(REPO, "include", "traceweaver_amd.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps)


def build(force=False, extra_flags=()):
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: the HIP engine cannot be built (there is no CPU fallback)")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc] + FLAGS + list(extra_flags) + ["-I", os.path.join(REPO, "include"), "-I", SRC,
                                                 os.path.join(SRC, "tw_engine.hip"), os.path.join(SRC, "tw_ingest.cpp"), "-o", OUT]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
