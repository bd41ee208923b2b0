"""Timeline of the last tracked frames of a rocprofv3 trace (--kernel-trace --memory-copy-trace): every copy and kernel with its
duration and the gap from the previous end. usage: python tools/trace_frame.py <trace dir> [n_ops]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
ops = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:56]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Size", "")))
ops.sort()
prev = None
for s, e, name in ops[-n:]:
    print(f"{name:62s} start {(s - ops[-n][0]) / 1e3:9.2f}  dur {(e - s) / 1e3:8.2f} us  gap {((s - prev) / 1e3 if prev else 0):8.2f} us")
    prev = e
