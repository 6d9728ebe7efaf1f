"""The build's scratch-memory guard (rag_amd/csrc/check_resources.py): its parser on a canned
compiler report, and -- when the library was built here -- the reports of the real build."""
import glob
import importlib.util
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "rag_amd", "csrc")


def _checker():
    spec = importlib.util.spec_from_file_location("check_resources", os.path.join(CSRC, "check_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


REPORT = """\
x.hip:86:1: remark: Function Name: _ZN3rag18narrow_topk_kernelILi1ELi2ELi4EEEvNS_10NarrowArgsE [-Rpass-analysis=kernel-resource-usage]
   86 | __global__ void narrow_topk_kernel(NarrowArgs a) {
      | ^
x.hip:86:1: remark:     VGPRs: 197 [-Rpass-analysis=kernel-resource-usage]
x.hip:86:1: remark:     ScratchSize [bytes/lane]: %d [-Rpass-analysis=kernel-resource-usage]
x.hip:86:1: remark:     Occupancy [waves/SIMD]: 2 [-Rpass-analysis=kernel-resource-usage]
x.hip:86:1: remark:     VGPRs Spill: %d [-Rpass-analysis=kernel-resource-usage]
x.hip:9:1: remark: Function Name: _ZN3rag11fuse_kernelENS_8FuseArgsE [-Rpass-analysis=kernel-resource-usage]
x.hip:9:1: remark:     ScratchSize [bytes/lane]: 64 [-Rpass-analysis=kernel-resource-usage]
"""


def test_guard_reads_the_compiler_report(tmp_path):
    chk = _checker()
    good = tmp_path / "good.usage.txt"
    good.write_text(REPORT % (0, 0))
    assert chk.main([str(good)]) == 0                      # the cold kernel's scratch is not its business
    ks = list(chk.kernels(str(good)))
    assert ks[0]["ScratchSize"] == "0" and ks[0]["VGPRs"] == "197" and ks[1]["ScratchSize"] == "64"
    bad = tmp_path / "bad.usage.txt"
    bad.write_text(REPORT % (48, 12))
    assert chk.main([str(bad)]) == 1
    empty = tmp_path / "none.usage.txt"
    empty.write_text("nothing here\n")
    assert chk.main([str(empty)]) == 2


def test_built_library_passes_the_guard():
    reports = sorted(glob.glob(os.path.join(CSRC, "*.usage.txt")))
    if not reports:
        pytest.skip("library not built in this tree")
    assert _checker().main(reports) == 0
