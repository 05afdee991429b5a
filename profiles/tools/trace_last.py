import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find last k_block_params (start of the last pass 1)
idx = max(i for i, r in enumerate(rows) if "k_block_params" in r["Kernel_Name"])
# back up to the sorts before it
start = idx
while start > 0 and ("rocprim" in rows[start-1]["Kernel_Name"] or "fill" in rows[start-1]["Kernel_Name"].lower()): start -= 1
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    n = r["Kernel_Name"]
    n = n.replace("tw::", "").split("(")[0][:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > 20000 or "k_" in n:
        print("%9.3f ms  +%8.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n))
