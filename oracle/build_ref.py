#!/usr/bin/env python
"""Build the part of the reference that compiles from its own sources: the Cython CPU ops
operator_py/cython/{bbox,cpu_nms}.pyx (bbox_overlaps_cython, greedy_nms, soft_nms).

TEST INFRASTRUCTURE ONLY (second, independent oracle for IoU / NMS / soft-NMS).

The sources are compiled where they lie under /root/reference; nothing is copied into the repo.
Outputs (extension modules + the cythonized C in a build dir) go to oracle/_ref/, which is
git-ignored but travels to the GPU box with the snapshot.  cpu_nms.pyx uses NumPy-1 aliases
(np.int_t / np.int, cpu_nms.pyx:45-49) that NumPy 2 removed; the build feeds Cython a patched
temporary copy (np.intp_t / np.intp): a type-alias change, no arithmetic is touched.

operator_cxx/ (the C++/CUDA operators) needs MXNet/mshadow/dmlc headers and nvcc: unbuildable here.
"""
import os
import re
import shutil
import subprocess
import sys
import sysconfig
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SIMPLEDET_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "operator_py", "cython")
OUT = os.path.join(HERE, "_ref")
MODULES = ("bbox", "cpu_nms")


def up_to_date():
    for m in MODULES:
        so = os.path.join(OUT, m + sysconfig.get_config_var("EXT_SUFFIX"))
        src = os.path.join(SRC, m + ".pyx")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            return False
    return True


def main():
    if not os.path.isdir(SRC):
        print("build_ref: %s not present; keeping prebuilt oracle/_ref" % SRC)
        return 0
    os.makedirs(OUT, exist_ok=True)
    open(os.path.join(OUT, "__init__.py"), "a").close()
    if up_to_date():
        return 0
    import numpy as np
    tmp = tempfile.mkdtemp(prefix="sd_ref_build_")
    try:
        for m in MODULES:
            text = open(os.path.join(SRC, m + ".pyx")).read()
            if m == "cpu_nms":
                text = re.sub(r"np\.int_t", "np.intp_t", text)
                text = re.sub(r"dtype=np\.int\)", "dtype=np.intp)", text)
            pyx = os.path.join(tmp, m + ".pyx")
            open(pyx, "w").write(text)
            subprocess.check_call([sys.executable, "-m", "cython", "-3", "--fast-fail", pyx, "-o",
                                   os.path.join(tmp, m + ".c")], cwd=tmp,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            so = os.path.join(OUT, m + sysconfig.get_config_var("EXT_SUFFIX"))
            # the reference's setup.py uses distutils defaults (-O2-ish, no -march, no fast-math)
            cmd = ["gcc", "-O2", "-fPIC", "-shared", "-fno-strict-aliasing", "-w",
                   "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
                   "-I" + sysconfig.get_paths()["include"], "-I" + np.get_include(),
                   os.path.join(tmp, m + ".c"), "-o", so]
            subprocess.check_call(cmd)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
