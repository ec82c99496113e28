#!/usr/bin/env python
"""Compile the reference's C++/CUDA operator sources, unmodified and where they lie under
/root/reference/operator_cxx, into oracle/_ref/libref_<op>.so.

TEST INFRASTRUCTURE ONLY (reference-run pins for the CPU oracle and the HIP kernels).

MXNet / mshadow / dmlc / nnvm / CUDA are not installable here, so the sources are compiled
against oracle/mxshim/ (a stand-in for the few hundred lines of those APIs the operators touch;
see mxshim.h and cuemu.h).  No reference source is copied into the repository:
  * .cc files are passed to g++ by their path under /root/reference;
  * .cu files need ONE textual rewrite because `kernel<<<grid, block>>>(args)` is not C++:
    each launch becomes CUEMU_LAUNCH(kernel, (grid, block), (args)).  The rewritten text goes to
    a temporary directory that is deleted after the compile.
Every library gets its own copy of the registry/driver (mxshim/runtime.cc) and is dlopen'ed with
RTLD_LOCAL by oracle/refmx.py, so same-named inline helpers of different operators
(utils::NonMaximumSuppression in nms.cc and proposal_v3.cc, ...) never meet in one image.

Flags follow MXNet's CPU build where it matters for numbers: -std=c++11, -O2, SSE2 baseline
(no FMA contraction possible), no -ffast-math.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SIMPLEDET_REFERENCE", "/root/reference")
CXX_ROOT = os.path.join(REF, "operator_cxx")
SHIM = os.path.join(HERE, "mxshim")
OUT = os.path.join(HERE, "_ref")

# library name -> reference sources (relative to operator_cxx/)
LIBS = {
    "roi_align_v2": ["contrib/roi_align_v2.cc", "contrib/roi_align_v2.cu"],
    "roi_pooling_v1": ["roi_pooling_v1.cc", "roi_pooling_v1.cu"],
    "proposal_target": ["proposal_target.cc"],
    "proposal_target_v2": ["proposal_target_v2.cc"],
    "decodebbox": ["contrib/decodebbox.cc"],
    "generate_anchor": ["contrib/generate_anchor.cc", "contrib/generate_anchor.cu"],
    "nms": ["contrib/nms.cc", "contrib/nms.cu"],
    "proposal_v3": ["contrib/proposal_v3.cc", "contrib/proposal_v3.cu"],
    # includes "../coco_api/common/maskApi.h" (github.com/RogerChern/cocoapi, NOT vendored): resolved
    # to mxshim/coco_api/common/maskApi.h + oracle/mask_api.c, a restatement of pycocotools'
    # rleFrPoly / rleDecode.  The op around it (sampling, polygon -> RoI frame) is reference code.
    "proposal_mask_target": ["proposal_mask_target.cc"],
}
# sources of THIS repository linked into a reference library (stand-ins for un-vendored third party)
EXTRA = {"proposal_mask_target": [os.path.join(HERE, "mask_api.c")]}

CXXFLAGS = ["-std=c++11", "-O2", "-fPIC", "-w", "-pthread", "-ffp-contract=off", "-fno-fast-math",
            "-fvisibility=default"]


def _match(text, i, open_ch, close_ch):
    """index just past the bracket that closes text[i] == open_ch (string/char literals skipped)."""
    depth = 0
    n = len(text)
    while i < n:
        c = text[i]
        if text.startswith("//", i):
            i = text.index("\n", i)
            continue
        if text.startswith("/*", i):
            i = text.index("*/", i) + 2
            continue
        if c == '"' or c == "'":
            q = c
            i += 1
            while i < n and text[i] != q:
                i += 2 if text[i] == "\\" else 1
        elif c == open_ch:
            depth += 1
        elif c == close_ch:
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1
    raise ValueError("unbalanced %s%s" % (open_ch, close_ch))


def rewrite_launches(text):
    """`name[<targs>] <<< cfg >>> ( args )` -> `CUEMU_LAUNCH(name[<targs>], (cfg), (args))`."""
    out = []
    pos = 0
    pat = re.compile(r"([A-Za-z_][A-Za-z0-9_:]*(?:\s*<[^<>;(){}]*>)?)\s*<<<")
    while True:
        m = pat.search(text, pos)
        if not m:
            out.append(text[pos:])
            break
        end_cfg = text.index(">>>", m.end())
        cfg = text[m.end():end_cfg]
        j = end_cfg + 3
        while text[j].isspace():
            j += 1
        assert text[j] == "(", "kernel launch without argument list"
        k = _match(text, j, "(", ")")
        out.append(text[pos:m.start()])
        out.append("CUEMU_LAUNCH(%s, (%s), %s)" % (m.group(1).strip(), cfg.strip(), text[j:k]))
        pos = k
    return "".join(out)


def barrier_kernels(text):
    """names of __global__ functions whose body calls __syncthreads()."""
    names = []
    for m in re.finditer(r"__global__\s+(?:\w+\s+)*?void\s+(\w+)\s*\(", text):
        i = _match(text, m.end() - 1, "(", ")")
        while text[i] != "{":
            i += 1
        body_end = _match(text, i, "{", "}")
        if "__syncthreads" in text[i:body_end]:
            names.append(m.group(1))
    return names


def up_to_date(lib, srcs):
    so = os.path.join(OUT, "libref_%s.so" % lib)
    if not os.path.exists(so):
        return False
    t = os.path.getmtime(so)
    deps = [os.path.join(CXX_ROOT, s) for s in srcs] + [__file__] + EXTRA.get(lib, [])
    for root, _, files in os.walk(SHIM):
        deps += [os.path.join(root, f) for f in files]
    for s in srcs:  # the -inl.h next to each source
        inl = os.path.join(CXX_ROOT, re.sub(r"\.(cc|cu)$", "-inl.h", s))
        if os.path.exists(inl):
            deps.append(inl)
    return all(os.path.getmtime(d) <= t for d in deps)


def build_lib(lib, srcs, tmp):
    incs = ["-I" + os.path.join(SHIM, "include"), "-I" + SHIM]
    objs = []
    for s in srcs:
        src = os.path.join(CXX_ROOT, s)
        sdir = os.path.dirname(src)
        # quoted includes: the source's own directory first (automatic for .cc, -I for the
        # rewritten .cu), then the stand-ins for "./operator_common.h", "../mshadow_op.h", ...
        rel = ["-I" + sdir, "-I" + os.path.join(SHIM, "op"), "-I" + os.path.join(SHIM, "op", "contrib")]
        obj = os.path.join(tmp, lib + "_" + os.path.basename(s).replace(".", "_") + ".o")
        if s.endswith(".cu"):
            text = open(src).read()
            bk = barrier_kernels(text)
            cu = os.path.join(tmp, lib + "_" + os.path.basename(s) + ".cc")
            with open(cu, "w") as f:
                f.write('#line 1 "%s"\n' % src)
                f.write(rewrite_launches(text))
            cmd = ["g++"] + CXXFLAGS + incs + rel + ["-include", os.path.join(SHIM, "cuemu.h"),
                                                    '-DCUEMU_BARRIER_KERNELS="%s"' % ",".join(bk),
                                                    "-c", cu, "-o", obj]
        else:
            cmd = ["g++"] + CXXFLAGS + incs + rel + ["-c", src, "-o", obj]
        subprocess.check_call(cmd)
        objs.append(obj)
    for extra in EXTRA.get(lib, []):
        obj = os.path.join(tmp, lib + "_" + os.path.basename(extra).replace(".", "_") + ".o")
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-w", "-ffp-contract=off", "-I" + HERE, "-c", extra, "-o", obj])
        objs.append(obj)
    rt = os.path.join(tmp, lib + "_runtime.o")
    subprocess.check_call(["g++"] + CXXFLAGS + incs + ["-c", os.path.join(SHIM, "runtime.cc"), "-o", rt])
    so = os.path.join(OUT, "libref_%s.so" % lib)
    subprocess.check_call(["g++", "-shared", "-pthread", "-Wl,-z,defs"] + objs + [rt, "-lm", "-o", so])


def main(argv):
    if not os.path.isdir(CXX_ROOT):
        print("build_ref_cxx: %s not present; keeping prebuilt oracle/_ref" % CXX_ROOT)
        return 0
    os.makedirs(OUT, exist_ok=True)
    want = [a for a in argv[1:] if not a.startswith("--")] or sorted(LIBS)
    tmp = tempfile.mkdtemp(prefix="sd_refcxx_")
    try:
        for lib in want:
            if up_to_date(lib, LIBS[lib]) and "--force" not in argv:
                continue
            build_lib(lib, LIBS[lib], tmp)
            print("build_ref_cxx: built libref_%s.so" % lib)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main([a for a in sys.argv]))
