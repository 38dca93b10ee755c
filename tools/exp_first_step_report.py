import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
g = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "pcg_ghost_kernel" in r["Kernel_Name"]]
print("solve durations (us), last 12:", [round(x, 1) for x in g[-12:]])
