"""Identity of the kernel build: a hash over the kernel sources and the build flags.  bench.py puts it on its line and the
profile summaries under profiles/ carry the one of the build they were measured on, so that counter figures are never quoted
for kernels they were not collected from."""
import hashlib
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
KERNEL_SOURCES = ("rt_kernels.hip", "rt_photon_build.hip", "rt_dev.h")


def build_flags():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    out = {}
    for key in ("FLAGS", "DEVFLAGS", "KERNELFLAGS"):
        m = re.search(r"^%s\s*:?=\s*(.*)$" % key, mk, re.M)
        out[key] = m.group(1).strip() if m else ""
    return out


def kernel_source_sha16():
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(open(os.path.join(CSRC, name), "rb").read())
    h.update(repr(sorted(build_flags().items())).encode())
    return h.hexdigest()[:16]
