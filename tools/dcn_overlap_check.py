"""do the two halves of the DCN backward overlap?  start / end of each kernel of one backward (rocprofv3 --kernel-trace csv)"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
sd = [r for r in rows if "sd::" in r[2]]
idx = [i for i, r in enumerate(sd) if "col2im_chunk" in r[2]]
k = idx[len(idx) // 2]   # a backward in the middle of the run
last = sd[max(0, k - 6):k + 8]
t0 = last[0][0]
for s, e, k, q in last:
    print("%9.1f .. %9.1f us  (%7.1f)  q %s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, k))
