"""Idle gaps per stream in one steady-state step of a rocprofv3 --kernel-trace CSV.
Usage: python tools/trace_gaps.py <kernel_trace.csv> <anchor-substring> [step-index] [min-gap-us]"""
import csv
import re
import sys

path, anchor = sys.argv[1], sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
min_gap = float(sys.argv[4]) if len(sys.argv) > 4 else 40.0
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", "?"), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if anchor in r[3]]
i0, i1 = starts[which], starts[which + 1]
t0, t1 = rows[i0][0], rows[i1][0]
print("step %.1f us" % ((t1 - t0) / 1e3))
by = {}
for s, e, st, name in rows[i0:i1]:
    by.setdefault(st, []).append((s, e, name))


def short(name):
    m = re.search(r"([A-Za-z_0-9]+)(<[^(]*>)?\(", name)
    return ((m.group(1) + (m.group(2) or "")) if m else name)[:50]


for st, ks in sorted(by.items()):
    busy = sum(e - s for s, e, _ in ks) / 1e3
    print("stream %s: %d kernels, busy %.0f us" % (st, len(ks), busy))
    prev_e, prev_n = t0, "(step start)"
    for s, e, name in ks:
        if (s - prev_e) / 1e3 >= min_gap:
            print("   gap %7.1f us at +%7.1f  between %s  and  %s" % ((s - prev_e) / 1e3, (prev_e - t0) / 1e3, short(prev_n), short(name)))
        prev_e, prev_n = max(prev_e, e), name
