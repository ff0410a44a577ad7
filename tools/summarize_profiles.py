"""Condenses rocprofv3 output (tools/collect_profiles.sh) into the files kept under profiles/."""
import csv, glob, json, os, sys

src, dst = sys.argv[1], sys.argv[2]


def find(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


stats = find("trace/**/*kernel_stats.csv")
if stats:
    with open(stats) as f, open(os.path.join(dst, "bench_kernel_stats.csv"), "w", newline="") as g:
        w = csv.writer(g)
        for row in csv.reader(f):
            row[0] = row[0][:160]  # torch's RNG kernel has a 4 KB mangled name
            w.writerow(row)

res = {}
for name, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    p = find(f"{sub}/**/*counter_collection.csv")
    if not p:
        continue
    vals = []
    with open(p) as f:
        for row in csv.DictReader(f):
            if "k_resample_periodic" in row.get("Kernel_Name", "") and row.get("Counter_Name") == name:
                vals.append(float(row["Counter_Value"]))
    if vals:
        res[name] = {"dispatches": len(vals), "mean_KB_per_dispatch": sum(vals) / len(vals),
                     "min": min(vals), "max": max(vals)}
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    res["note"] = ("separate --pmc passes of `python3 bench.py --steps 200 --warmup 30 --cpu-seconds 0` "
                   "(kernel k_resample_periodic). Per /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE "
                   "under-reports wide coalesced reads by exactly 2x on gfx950, WRITE_SIZE is exact; corrected "
                   "HBM traffic per launch = 2*FETCH + WRITE.")
    res["corrected_bytes_per_launch"] = (2 * res["FETCH_SIZE"]["mean_KB_per_dispatch"]
                                         + res["WRITE_SIZE"]["mean_KB_per_dispatch"]) * 1024
with open(os.path.join(dst, "bench_pmc_hbm.json"), "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res)[:600])
