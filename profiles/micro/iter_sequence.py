import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# last full iteration: find the last two k_global_pass and print the sequence between
idx = [i for i, r in enumerate(rows) if "k_global_pass" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
prev = int(rows[a]["End_Timestamp"])
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
    print("%8.2f us  gap %6.2f  dur %6.2f  %s grid %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, k[-30:], r.get("Grid_Size", "")))
    prev = e
