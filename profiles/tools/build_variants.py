"""Builds libtwgpu with alternative compile-time settings side by side (scratch/variants/<name>.so; scratch/ is not
tracked but travels to the GPU box), so that one gpurun call can time all of them with batch_sweep.py --lib.
The settings only move resources (registers per thread, LDS per workgroup): every variant computes the same results.

    python profiles/tools/build_variants.py [name ...]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from traceweaver_amd import build as B  # noqa: E402

WPE = lambda n: "__attribute__((amdgpu_waves_per_eu(%d)))" % n  # noqa: E731

VARIANTS = {
    "base": [],
    "memo64": ["-DTW_MEMO_SLOTS=64"],
    "memo32": ["-DTW_MEMO_SLOTS=32"],
    "tile_big": ["-DTW_TILE_ITEMS=1024", "-DTW_TILE_GRID=1536"],
    "tile128": ["-DTW_TILE_MAX=128"],
    "heavy3": ["-DTW_HEAVY_ATTR=" + WPE(3)],
    "heavy4": ["-DTW_HEAVY_ATTR=" + WPE(4)],
    "hw_all2": ["-DTW_HEAVY_WAVES(E)=2"],                 # round 5: every endpoint-count class of k_enumerate_heavy at two wavefronts per SIMD
    "tile384": ["-DTW_TILE_MAX=384"],
    "tile512": ["-DTW_TILE_MAX=512"],
    "tile768g": ["-DTW_TILE_MAX=768", "-DTW_TILE_ITEMS=1024", "-DTW_TILE_GRID=1536"],
    "parts16": ["-DTW_MAX_PARTS=16"],
    "parts48s1k": ["-DTW_DEFER_TUPLES=2048", "-DTW_SPLIT_TUPLES=1024"],
    "dptrace": ["-DTW_DP_TRACE"],
    "prof4": ["-DTW_PROFILE", "-DTW_PROFILE_E=4"],
    "prof8": ["-DTW_PROFILE", "-DTW_PROFILE_E=8"],
    "prof7": ["-DTW_PROFILE", "-DTW_PROFILE_E=7"],
    "log256": ["-DTW_PART_LOG_CAP=256"],
    "log512": ["-DTW_PART_LOG_CAP=512"],
    "hw_e6": ["-DTW_HEAVY_WAVES(E)=((E)<=6?2:1)"],
    "proftile6": ["-DTW_PROFILE_TILE", "-DTW_PROFILE_TILE_E=6"],
    "proftile8": ["-DTW_PROFILE_TILE", "-DTW_PROFILE_TILE_E=8"],
    "lean3": ["-DTW_LEAN_ATTR=__attribute__((amdgpu_waves_per_eu(3,3)))"],   # round 6: k_enumerate_lean held to 168 / 128 registers
    "lean4": ["-DTW_LEAN_ATTR=__attribute__((amdgpu_waves_per_eu(4,4)))"],
    "lean2": ["-DTW_LEAN_ATTR=__attribute__((amdgpu_waves_per_eu(2,2)))"],
    "dp96": ["-DTW_DP_CAP=96", "-DTW_DP_SLOTS=128"],       # round 6: tables of k_select_dp (states a level, hash slots): LDS per workgroup 69 -> 34 / 46 KB
    "dp192": ["-DTW_DP_CAP=192", "-DTW_DP_SLOTS=256"],
}


def build(name):
    out = os.path.join(REPO, "scratch", "variants", name + ".so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + VARIANTS[name] + ["-I", os.path.join(REPO, "include"), "-I", B.SRC,
                                                                os.path.join(B.SRC, "tw_engine.hip"), os.path.join(B.SRC, "tw_ingest.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return name, r.returncode, r.stderr[-2000:]


if __name__ == "__main__":
    names = sys.argv[1:] or list(VARIANTS)
    with ThreadPoolExecutor(4) as ex:
        for name, rc, err in ex.map(build, names):
            print(name, rc, err if rc else "")
