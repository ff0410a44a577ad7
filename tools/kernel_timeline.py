import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Stream_Id", "?")) for r in rows if "so::" in r["Kernel_Name"]]
ks.sort()
# take the last execute: last 33 kernels
ks = ks[-33:]
t0 = ks[0][0]
for a, b, n, sid in ks:
    print(f"{(a-t0)/1e3:9.1f} {(b-t0)/1e3:9.1f} {(b-a)/1e3:8.1f} us  stream {sid}  {n}")
print("span us", (max(b for a, b, n, s in ks) - t0) / 1e3)
