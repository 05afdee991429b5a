"""Builds HEAD and HEAD + the patches of this directory, cumulatively, side by side:

    scratch/variants/s0_head.so, s1_select_mid.so, s2_dense_terms.so, s3_segmented_lists.so, s4_select_big.so, s5_conflict_lanes.so, s6_light3.so (= s5 with other resources)

(scratch/ is not tracked but travels to the GPU box).  The patches are applied to copies of the sources under a temporary
directory; the working tree is not touched.  Then, in one gpurun call:  bash profiles/experiments/run_stack.sh
"""
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from traceweaver_amd import build as B  # noqa: E402

STACK = [("s0_head", None), ("s1_select_mid", "select_mid_instantiation.patch"), ("s2_dense_terms", "dense_mixture_terms.patch"),
         ("s3_segmented_lists", "segmented_selection_lists.patch"), ("s4_select_big", "select_big_instantiation.patch"),
         ("s5_conflict_lanes", "conflict_relation_lanes.patch")]


def main():
    tmp = tempfile.mkdtemp(prefix="tw_stack_")
    work = os.path.join(tmp, "work")
    os.makedirs(os.path.join(work, "traceweaver_amd"))
    shutil.copytree(B.SRC, os.path.join(work, "traceweaver_amd", "csrc"))
    jobs = []
    for name, patch in STACK:
        if patch:
            subprocess.check_call(["git", "apply", "--unsafe-paths", "--directory", work, os.path.join(HERE, patch)], cwd=tmp)
        src = os.path.join(tmp, name)
        shutil.copytree(os.path.join(work, "traceweaver_amd", "csrc"), src)
        jobs.append((name, src, []))
    # the top of the stack once more with the per-thread kernels capped to 168 VGPRs and narrower term tables (three wavefronts
    # per SIMD; neutral without the dense term evaluation, r02d_variants.jsonl `light3` -- the terms that fall off the table were
    # what it cost)
    jobs.append(("s6_light3", jobs[-1][1], ["-DTW_LIGHT_ATTR=__attribute__((amdgpu_waves_per_eu(3)))",
                                            "-DTW_LIGHT_TABW(E)=((E)==1?8:(E)==2?6:(E)==3?4:(E)==4?3:2)"]))
    os.makedirs(os.path.join(REPO, "scratch", "variants"), exist_ok=True)

    def build(job):
        name, src, extra = job
        out = os.path.join(REPO, "scratch", "variants", name + ".so")
        cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + extra + ["-I", os.path.join(REPO, "include"), "-I", src,
                                                  os.path.join(src, "tw_engine.hip"), os.path.join(src, "tw_ingest.cpp"), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return name, r.returncode, r.stderr[-1500:]

    with ThreadPoolExecutor(7) as ex:
        for name, rc, err in ex.map(build, jobs):
            print(name, "ok" if rc == 0 else "FAILED\n" + err)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
