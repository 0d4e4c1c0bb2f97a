"""Per-stream timeline of one steady-state step from a rocprofv3 --kernel-trace CSV: every kernel >= min_us with its start offset.
Usage: python tools/exp/timeline.py <kernel_trace.csv> <anchor-substring> [step-index] [min-us]"""
import csv
import re
import sys

path, anchor = sys.argv[1], sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 15.0
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", "?"), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if anchor in r[3]]
i0, i1 = starts[which], starts[which + 1]
t0, t1 = rows[i0][0], rows[i1][0]
print("step %.1f us" % ((t1 - t0) / 1e3))


def short(name):
    m = re.search(r"([A-Za-z_0-9]+)(<[^(]*>)?\(", name)
    return ((m.group(1) + (m.group(2) or "")) if m else name)[:44]


for s, e, st, name in rows[i0:i1]:
    if (e - s) / 1e3 >= min_us:
        print("  +%7.1f .. +%7.1f  (%6.1f us)  stream %-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, st, short(name)))
