import sys, collections
rows = [l.split() for l in open(sys.argv[1]) if l.startswith("[rs-trace]")]
idx = max(i for i, r in enumerate(rows) if r[2] == "w00" and r[3] == "it00")
rows = rows[idx:]
d = collections.defaultdict(dict); role = {}
for r in rows:
    w = int(r[2][1:]); it = int(r[3][2:]); role[w] = r[1]
    d[w][it] = [int(x) for x in r[4:]]
lo, hi = 8, 40
print("wave role period | k0->k5  k5->k6(wait)  k6->k7(rmw)  k7->k1  k1->k2(barrier)  k2->k3(issue)  k3->k4(gains)")
for w in sorted(d):
    its = [i for i in range(lo, hi) if i in d[w] and i + 1 in d[w]]
    if not its: continue
    def seg(a, b):
        v = [d[w][i][b] - d[w][i][a] for i in its if d[w][i][a] > 0 and d[w][i][b] > 0]
        return sum(v) / len(v) if v else float('nan')
    per = sum(d[w][i + 1][0] - d[w][i][0] for i in its) / len(its)
    if role[w] == "C":
        print(f"w{w:02d} C {per:8.0f} | mfma+stores {seg(0,3):8.0f}  gains {seg(3,4):8.0f}  ->barrier {seg(4,1):6.0f}  barrier wait {seg(1,2):8.0f}")
        continue
    print(f"w{w:02d} {role[w]} {per:8.0f} | {seg(0,5):8.0f} {seg(5,6):8.0f} {seg(6,7):8.0f} {seg(7,1):8.0f} {seg(1,2):8.0f} {seg(2,3):8.0f} {seg(3,4):8.0f}")
