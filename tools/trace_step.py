"""Print one steady-state step of a rocprofv3 --kernel-trace CSV as a timeline: start offset, duration, queue, kernel.
Usage: python tools/trace_step.py <kernel_trace.csv> [anchor-substring] [step-index]"""
import csv
import re
import sys

path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "hash_bucket_i64_kernel"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if anchor in r[4]]
i0, i1 = starts[which], starts[which + 1]
t0 = rows[i0][0]
print("step = %.1f us, %d kernels" % ((rows[i1][0] - t0) / 1e3, i1 - i0))
prev_end = {}
for s, e, q, st, name in rows[i0:i1]:
    m = re.search(r"([A-Za-z_0-9]+)(<[^(]*>)?\(", name)
    short = (m.group(1) + (m.group(2) or "")) if m else name[:60]
    print("%9.1f  +%8.1f us  q%-3s s%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, st, short[:90]))
