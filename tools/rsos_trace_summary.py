"""Summary of a SIGOPS_RSOS_TRACE run (stderr of any sink through k_rsos): per role, the mean cycles between
its stamps over the recorded iterations of workgroup 0's last execute."""
import statistics
import sys

lines = [l.split() for l in open(sys.argv[1]) if l.startswith('[rsos-trace]')]
waves = sorted({l[2] for l in lines})
per_exec = len({(l[2], l[3]) for l in lines})
last = lines[-per_exec:]


def rows(w):
    return [[int(x) for x in l[4:]] for l in last if l[2] == w]


for w in waves:
    R = rows(w)
    kind = [l[1] for l in last if l[2] == w][0]
    if len(R) < 3:
        continue
    if kind == 'L':
        R = [r for r in R if r[0] >= 0 and r[4] >= 0]
        per = statistics.mean(R[i + 1][0] - R[i][0] for i in range(len(R) - 1))
        print(f"{w} loader: chunk period {per:.0f}  issue {statistics.mean(r[1]-r[0] for r in R):.0f}  "
              f"landing wait {statistics.mean(r[3]-r[2] for r in R):.0f}  gain+publish {statistics.mean(r[4]-r[3] for r in R):.0f}  "
              f"issue->retire {statistics.mean(r[2]-r[1] for r in R):.0f}")
    elif kind == 'C':
        R = [r for r in R if r[0] >= 0 and r[3] >= 0]
        per = statistics.mean(R[i + 1][0] - R[i][0] for i in range(len(R) - 1))
        # stamps: 0 top of a pair of blocks, 3 its end
        print(f"{w} chain: period of a PAIR of blocks {per:.0f} (min {min(R[i + 1][0] - R[i][0] for i in range(len(R) - 1))}, "
              f"max {max(R[i + 1][0] - R[i][0] for i in range(len(R) - 1))})  body {statistics.mean(r[3]-r[0] for r in R):.0f}")
    else:
        R = [r for r in R if min(r[:4]) >= 0]
        if len(R) < 3:
            continue
        per = statistics.mean(R[i + 1][0] - R[i][0] for i in range(len(R) - 1))
        # stamps: 0 block start, 1 input there, 2 resampled (LDS batch + KS MFMAs), 3 end of block (D and T parts, D.x out,
        # back part of the previous block); 4 / 5 inside that back part (state there / stored), recorded under the
        # PREVIOUS block's index
        print(f"{w} y: block period {per:.0f}  input wait {statistics.mean(r[1]-r[0] for r in R):.0f}  "
              f"LDS batch + resample {statistics.mean(r[2]-r[1] for r in R):.0f}  D, T, out, back(prev) {statistics.mean(r[3]-r[2] for r in R):.0f}  "
              f"[back: C part + store {statistics.mean([r[5]-r[4] for r in R if r[4] >= 0 and r[5] >= 0] or [0]):.0f}]")
