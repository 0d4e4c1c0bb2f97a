"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (average counter value per launch).
Usage: python tools/pmc_traffic.py <counter_collection.csv> <out.json>"""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"(?:\)::|::|^)([A-Za-z_0-9]+)(<[^(]*>)?\(", name)
    base = m.group(1) if m else name[:40]
    targs = (m.group(2) or "") if m else ""
    return base + targs.replace(" ", "")


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        k = r["Kernel_Name"]
        if "at::native" in k or "rocclr" in k:
            continue
        agg[short(k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: {c: {"calls": len(v), "avg": sum(v) / len(v)} for c, v in d.items()} for k, d in agg.items()}
    json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
    for k, d in sorted(out.items()):
        print(k, {c: round(v["avg"], 1) for c, v in d.items()})


if __name__ == "__main__":
    main()
