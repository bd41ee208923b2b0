import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 9 launches = one frame (copy not included): print name, duration, gap from previous end
tail = rows[-27:]
prev = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{r["Kernel_Name"][:60]:60s} dur {(e-s)/1e3:8.2f} us  gap {((s-prev)/1e3 if prev else 0):8.2f} us')
    prev = e
