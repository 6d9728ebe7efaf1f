#!/usr/bin/env python3
"""Build-time guard: the hot kernels must use no private (scratch) memory.

Reads the `-Rpass-analysis=kernel-resource-usage` remarks the Makefile saved next to every
object (`*.usage.txt`) and fails when a kernel whose name contains one of HOT reports a
non-zero `ScratchSize [bytes/lane]` or a spilled VGPR (ALLOW lists the one bounded
exception).  A spill in these kernels turns an HBM- or MFMA-bound loop into a scratch-bound
one without changing any result, so no parity test would catch it.

usage: check_resources.py a.usage.txt [b.usage.txt ...]
"""
import re
import sys

HOT = ("gemm_topk_kernel", "narrow_topk_kernel", "flat_scan_kernel", "merge_keys_kernel", "merge_keys_small_kernel",
       "mmr_simmat_kernel", "rescore_select_kernel", "direct_scores_kernel", "direct_select_kernel", "ivf_lists_kernel")
# Known and bounded: bytes of scratch per lane a kernel may use.  The fp32 256 x 256 shape sits at
# the 256-register limit of two waves per SIMD (128 accumulators, two fragment sets, loader
# addresses); the allocator parks loader addresses and the work-item bounds in scratch and reloads
# them once per 256-row tile (about 11 loads per ~160 us of MFMA work), none between the first and
# the last MFMA of a K step.  Anything beyond this bound is a regression.
ALLOW = {"gemm_topk_kernelILi256ELi256ELi8ELb0ELb0E": 96}
NAME = re.compile(r"remark: Function Name: (\S+)")
FIELD = re.compile(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass")


def kernels(path):
    cur = None
    with open(path, errors="replace") as f:
        for line in f:
            m = NAME.search(line)
            if m:
                cur = {"name": m.group(1)}
                yield cur
                continue
            m = FIELD.search(line)
            if m and cur is not None:
                cur[m.group(1).strip()] = m.group(2)


def main(paths):
    seen, bad = 0, []
    for p in paths:
        for k in list(kernels(p)):
            if not any(h in k["name"] for h in HOT):
                continue
            seen += 1
            scratch = int(k.get("ScratchSize", "0"))
            vspill = int(k.get("VGPRs Spill", "0"))
            allow = max([v for n, v in ALLOW.items() if n in k["name"]] or [0])
            if scratch > allow or (vspill and not allow):
                bad.append(f"{k['name']}: ScratchSize {scratch} B/lane, {vspill} VGPRs spilled ({p})")
    if seen == 0:
        print("check_resources: no hot kernel found in", " ".join(paths), file=sys.stderr)
        return 2
    for b in bad:
        print("check_resources: " + b, file=sys.stderr)
    if not bad:
        print(f"check_resources: {seen} hot kernels, no scratch beyond the stated allowance")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
