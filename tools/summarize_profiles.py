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
            row[0] = row[0][:200]  # torch's RNG kernel has a 4 KB mangled name
            w.writerow(row)

STAGES = {"k_rsos": ["k_rsos", "k_sos_poison"], "k_resample_periodic": ["k_resample_periodic"],
          "k_sos": ["k_sos_tiled", "k_sos_scan", "k_sos_onepass", "k_sos_batch", "k_sos_"], "k_pointwise": ["k_pointwise"]}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (kernel_sources_sha16: bench.py quotes these counters only while the kernel sources are the same)
res = {"note": ("separate --pmc passes of `python3 bench.py --workload W --no-secondary --steps 100 --warmup 10 "
                "--cpu-seconds 0`.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and "
                "WRITE_SIZE are in KB; FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950, WRITE_SIZE is "
                "exact; corrected HBM traffic = 2*FETCH + WRITE.  Per execute = summed over the stage's kernels / number of "
                "executes."), "kernel_sources_sha16": None, "stages": {}}
_h = os.path.join(src, "kernel_sources_sha16.txt")  # (written by collect_profiles.sh before the passes)
res["kernel_sources_sha16"] = open(_h).read().strip() if os.path.exists(_h) else bench.kernel_sources_sha16()
res["kernel_sources_sha16_taken"] = "before the PMC passes (collect_profiles.sh)" if os.path.exists(_h) else "when summarised"
# (workload tags: the sub-directories fetch_<tag> / write_<tag> that exist -- tools/collect_profiles.sh, tools/collect_r06.sh)
_tags = sorted({os.path.basename(d)[len("fetch_"):] for d in glob.glob(os.path.join(src, "fetch_*")) if os.path.isdir(d)})
for wl in (_tags or ["ns", "config3"]):
    per = {}
    for name, sub in (("FETCH_SIZE", f"fetch_{wl}"), ("WRITE_SIZE", f"write_{wl}")):
        p = find(f"{sub}/**/*counter_collection.csv")
        if not p:
            continue
        with open(p) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != name:
                    continue
                kn = row.get("Kernel_Name", "")
                for stage, keys in STAGES.items():
                    if any(k in kn for k in keys):
                        d = per.setdefault(stage, {}).setdefault(name, {})
                        short = kn.split("(")[0][-90:]
                        e = d.setdefault(short, [0, 0.0])
                        e[0] += 1
                        e[1] += float(row["Counter_Value"])
    for stage, d in per.items():
        if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
            continue
        nexec = max(v[0] for v in d["FETCH_SIZE"].values()) if stage == "k_rsos" else min(v[0] for v in d["FETCH_SIZE"].values())  # launched once per execute
        fetch_kb = sum(v[1] for v in d["FETCH_SIZE"].values()) / nexec
        write_kb = sum(v[1] for v in d["WRITE_SIZE"].values()) / nexec
        res["stages"][f"{wl}:{stage}"] = {
            "executes": nexec, "FETCH_SIZE_KB_per_execute": fetch_kb, "WRITE_SIZE_KB_per_execute": write_kb,
            "corrected_bytes_per_execute": (2 * fetch_kb + write_kb) * 1024,
            "kernels": {k: {"dispatches": v[0], "FETCH_KB_mean": v[1] / v[0],
                            "WRITE_KB_mean": d["WRITE_SIZE"].get(k, [1, 0.0])[1] / max(1, d["WRITE_SIZE"].get(k, [1, 0.0])[0])}
                        for k, v in d["FETCH_SIZE"].items()}}
with open(os.path.join(dst, "bench_pmc_hbm.json"), "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res)[:1500])
