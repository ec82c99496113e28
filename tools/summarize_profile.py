#!/usr/bin/env python
"""Condense a tools/profile_round.sh output directory into the files committed under profiles/:
  <dir>/kernel_stats.csv   rocprofv3 --kernel-trace --stats rows of our kernels (sd::*)
  <dir>/pmc_summary.json   per-kernel average of every collected counter + derived HBM bytes
FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3); MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE
counts 128-B read requests as 64 B for wide coalesced streams, so the read side is reported both
raw and x2, next to TCC_EA0_RDREQ x 64 B and the calibration stream (sd::hbm_stream_copy) whose
true byte count is known.
"""
import collections
import csv
import glob
import json
import os
import sys


def main(out):
    res = {"kernels": {}, "calibration": {}}
    stats_rows = []
    for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            stats_rows.append(row)
    with open(os.path.join(out, "kernel_stats.csv"), "w") as fo:
        w = csv.writer(fo)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs",
                    "StdDev"])
        for r in stats_rows:
            w.writerow([r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage",
                                       "MinNs", "MaxNs", "StdDev")])
            print("%-90s calls %5s avg %9.1f us" % (r["Name"][:90], r["Calls"],
                                                     float(r["AverageNs"]) / 1e3))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in sorted(glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "sd::" not in k:
                continue
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta[k] = {"grid": int(row["Grid_Size"]), "wg": int(row["Workgroup_Size"]),
                       "lds": int(row["LDS_Block_Size"]), "vgpr": int(row["VGPR_Count"]),
                       "sgpr": int(row["SGPR_Count"]), "scratch": int(row["Scratch_Size"])}
    for k, cs in agg.items():
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        d["n_dispatch"] = len(next(iter(cs.values())))
        if "FETCH_SIZE" in d:
            d["fetch_bytes_raw"] = d["FETCH_SIZE"] * 1024
            d["fetch_bytes_x2_gfx950"] = 2 * d["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in d:
            d["write_bytes"] = d["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
            d["l2_hit_rate"] = d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
        d.update(meta[k])
        res["kernels"][k] = d
        print("==", k[:100])
        print("   ", {c: ("%.5g" % v if isinstance(v, float) else v) for c, v in d.items()})
    # which kernel sources these counters belong to (bench.py quotes a profile only for the same hash)
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("roi_align_common.h", "roi_align_lists.h", "roi_align_fwd.hip", "roi_align_bwd.hip", "roi_align_prep.hip",
              "common.h", "runtime.hip"):
        h.update(open(os.path.join(root, "simpledet_amd", "csrc", f), "rb").read())
    res["kernel_source_sha256"] = h.hexdigest()
    # ... and per file, for the quotes that concern other kernels (bench.py: NMS L2 hit rates need nms.hip /
    # soft_nms.hip / common.h unchanged)
    res["source_sha256"] = {}
    for f in sorted(glob.glob(os.path.join(root, "simpledet_amd", "csrc", "*.h*"))):
        res["source_sha256"][os.path.basename(f)] = hashlib.sha256(open(f, "rb").read()).hexdigest()
    json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1])
