"""Timeline of a graph-replayed solve from a rocprofv3 kernel trace: per kernel the in-kernel duration and the gap to the previous
kernel's end, averaged over the solves of the run, factoring and non-factoring k_dense launches apart.
usage (GPU box):  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $REPO/bench.py --steps 30 --warmup 5 --no-klt --no-cpu-baseline --no-scaling-window --no-pmc
                  python tests/prof_timeline.py /tmp/tl"""
import csv, glob, os, re, sys
from collections import defaultdict

rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = ["k_linearize", "k_reduce", "k_dense", "k_backsub"]
short = lambda n: next((k for k in names if re.search(r"\b%s\b" % k, n)), None)
seq = [(s, e, short(n)) for s, e, n in rows if short(n)]
# slots = consecutive [linearize, reduce, dense, backsub]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
slots = []
i = 0
while i + 3 < len(seq):
    if [x[2] for x in seq[i:i + 4]] == names:
        slots.append(seq[i:i + 4])
        i += 4
    else:
        i += 1
# keep the second half of the run (steady state, graph replays)
slots = slots[len(slots) // 2:]
prev_end = None
for sl in slots:
    factoring = (sl[2][1] - sl[2][0]) > 35000
    idle = (sl[0][1] - sl[0][0]) < 6000  # a no-op slot of a finished solve
    for s, e, k in sl:
        key = k if k != "k_dense" else ("k_dense/factoring" if factoring else "k_dense/other")
        if idle:
            key = k + " (slot after done)"
        dur[key] += e - s
        cnt[key] += 1
        if prev_end is not None and s - prev_end < 200000:
            gap[key] += s - prev_end
        prev_end = e
print("%-34s %8s %10s %10s" % ("kernel", "launches", "in-kernel", "gap before"))
for k in sorted(dur):
    print("%-34s %8d %8.2f us %8.2f us" % (k, cnt[k], dur[k] / cnt[k] / 1e3, gap[k] / cnt[k] / 1e3))
tot = sum(dur.values()) + sum(gap.values())
print("sum over %d slots: %.1f us per slot (in-kernel %.1f, gaps %.1f)" % (len(slots), tot / len(slots) / 1e3, sum(dur.values()) / len(slots) / 1e3, sum(gap.values()) / len(slots) / 1e3))
