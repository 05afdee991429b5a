"""The prepared kernel patches of profiles/experiments/ (emulation-verified, waiting for GPU time: README there) must keep
applying, in their order, to the kernel sources of the tree -- a change to those sources that breaks one shows up here, not on
the GPU box."""
import os
import shutil
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("git") is None, reason="git is not installed")
def test_patch_stack_applies_in_order(tmp_path):
    sys.path.insert(0, os.path.join(REPO, "profiles", "experiments"))
    import build_stack

    work = tmp_path / "work"
    (work / "traceweaver_amd").mkdir(parents=True)
    shutil.copytree(os.path.join(REPO, "traceweaver_amd", "csrc"), work / "traceweaver_amd" / "csrc")
    for name, patch in build_stack.STACK:
        if patch is None:
            continue
        path = os.path.join(REPO, "profiles", "experiments", patch)
        assert os.path.exists(path), patch
        r = subprocess.run(["git", "apply", "--unsafe-paths", "--directory", str(work), path], cwd=str(tmp_path), capture_output=True, text=True)
        assert r.returncode == 0, "%s (%s) does not apply any more:\n%s" % (patch, name, r.stderr[-1500:])
    # the patched sources differ from the tree's (the patches did something) and still hold the kernels they touch
    text = open(work / "traceweaver_amd" / "csrc" / "tw_kernels.h").read()
    for needle in ("SelectLdsMid", "SelectLdsBig", "score_term_mix_x", "sel_segments"):
        assert needle in text, needle
