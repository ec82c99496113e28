#!/usr/bin/env python
"""Golden fixtures produced by THE REFERENCE'S OWN operator sources.

Run in the build container (needs /root/reference):
    python oracle/build_ref_cxx.py && python tests/golden/make_golden_cxx.py
oracle/build_ref_cxx.py compiles operator_cxx/{roi_align_v2,roi_pooling_v1,proposal_target,
generate_anchor,nms,proposal_v3,decodebbox,...}.{cc,cu} where they lie (against oracle/mxshim/)
into oracle/_ref/libref_*.so; this script runs every case of tests/refcases.py through them and
stores
    ref_cxx_digests.json   SHA-256 of each bit-exact output (+ shape, dtype)
    ref_cxx_arrays.npz     the outputs that are compared with a tolerance
Inputs are regenerated from seeds (simpledet_amd.synth), so the fixtures stay small.  Tests read
only these two files; /root/reference is not needed to run them.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests import refcases  # noqa: E402


def main():
    digests, arrays = {}, {}
    for name in sorted(refcases.CASES):
        case = refcases.CASES[name]
        res = refcases.run_case(name, "ref")
        digests[name] = {}
        for k, v in res.items():
            assert not np.isnan(v).any(), (name, k, "the reference left part of the output unwritten")
            digests[name][k] = {"sha256": refcases.digest(v), "shape": list(v.shape), "dtype": str(v.dtype)}
            if case["kind"] != "exact" or k in case["hip_close"]:
                arrays["%s/%s" % (name, k)] = v
        print("%-34s %s" % (name, " ".join("%s%s" % (k, tuple(v.shape)) for k, v in res.items())))
    with open(os.path.join(HERE, "ref_cxx_digests.json"), "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "ref_cxx_arrays.npz"), **arrays)


if __name__ == "__main__":
    main()
