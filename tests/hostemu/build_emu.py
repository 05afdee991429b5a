"""Builds libtwgpu_emu.so: the engine source compiled with g++ against the host-emulation shim.
TEST INFRASTRUCTURE ONLY -- see tests/hostemu/hip/hip_runtime.h."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build", "libtwgpu_emu.so")
SRC = os.path.join(REPO, "traceweaver_amd", "csrc")


def build(force=False, production=False):
    """production=True: the thresholds of the HIP build (long enumerations from 768 tuples, split from 4096 grid points, tuple lists
    of 2^15 entries) instead of the tiny ones that make the small test units take every route."""
    out = OUT.replace("libtwgpu_emu.so", "libtwgpu_emu_prod.so") if production else OUT
    deps = [os.path.join(SRC, f) for f in ("tw_engine.hip", "tw_kernels.h", "tw_tile.h", "tw_lean.h", "tw_device.h", "tw_fit.h", "tw_eval.h", "tw_skip.h", "tw_load.h", "tw_baselines.h", "tw_ingest.cpp")]
    deps += [os.path.join(REPO, "include", "traceweaver_amd.h"), os.path.join(HERE, "hip", "hip_runtime.h"),
             os.path.join(HERE, "rocprim", "rocprim.hpp")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    small = [] if production else [
        "-DTW_BIG_PRODUCT=40", "-DTW_SPLIT_MIN=96", "-DTW_SPLIT_GRAIN=24", "-DTW_GRID_TARGET=16",   # enumerations are split from ~100 grid points on, prefixes walked
        "-DTW_FRONTIER_CAP=512", "-DTW_FRONTIER_BIG_CAP=8192", "-DTW_FRONTIER_BIG_SLOTS=3",   # small lists: own buffers, pool slots and the walk all occur in the tests
        "-DTW_MATCH_NODES_1=48", "-DTW_MATCH_NODES=256",
        "-DTW_TILE_SMALL", "-DTW_TILE_MAX=24",
        "-DTW_DP_CAP=24", "-DTW_DP_SLOTS=32", "-DTW_DP_CAP_SMALL=6", "-DTW_DP_SLOTS_SMALL=8",   # small tables of the level-by-level solver: both overflow routes (k_select_dp, then the depth-first search) occur
        "-DTW_DEFER_TUPLES=40", "-DTW_SPLIT_TUPLES=12", "-DTW_DEFER_PREFIXES=24", "-DTW_DEFER_PREFIX_GRAIN=6",   # spans of the deferring classes (five endpoints and more) with more than 40 tuples are cut into list parts of a dozen
        "-DTW_LIST_SCORE_MAX=64",   # listed enumerations of up to 64 tuples are scored from the list, longer ones counted and walked: both occur
        "-DTW_PART_LOG_CAP=12",   # log-mode parts: a dozen entries, so that some logs overflow and the span is enumerated again as a whole
        "-DTW_FIT_HASH_SLOTS=512",   # refit: rows of more than 384 distinct gap values take the sort route, the others the hash table
        "-DTW_PRUNE_MIN=48", "-DTW_PRUNE_GRID=8", "-DTW_LEAN_GRID=40"]   # the wavefront kernel's walk is pruned from 48 grid points on   # the tile kernel: short slices (windows beyond them go to the wavefront kernel), several segments per tile   # the selection search consults the matching relaxation early
    subprocess.check_call(
        ["g++", "-x", "c++", "-std=c++17", "-O2", "-g", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unknown-pragmas",
         # the emulated LDS arrays are static locals of kernel templates: as GNU-unique symbols the two builds of this library
         # (tiny / production table sizes) would share them when a test process loads both
         "-fno-gnu-unique", "-Wl,-Bsymbolic"] + small +
        ["-I", HERE, "-I", os.path.join(REPO, "include"), "-I", SRC, os.path.join(SRC, "tw_engine.hip"),
         "-x", "c++", os.path.join(SRC, "tw_ingest.cpp"), "-pthread", "-o", out])
    return out


if __name__ == "__main__":
    print(build(force=True))
