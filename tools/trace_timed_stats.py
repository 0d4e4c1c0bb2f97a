"""Per-kernel statistics of a rocprofv3 kernel trace restricted to the TIMED launches of bench.py: the last `--last N`
dispatches of every kernel (bench.py launches each hot-path kernel once per step; its warm-up steps come first and are
what makes rocprofv3's own --stats average a few percent higher than bench.py's HIP-event mean).
Usage: python tools/trace_timed_stats.py <kernel_trace.csv> --last 20 [--min-calls 20] > stats.csv"""
import argparse
import collections
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--last", type=int, default=20)
    ap.add_argument("--min-calls", type=int, default=20)
    a = ap.parse_args()
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(a.trace)):
        per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows = []
    for name, ev in per.items():
        if len(ev) < a.min_calls:
            continue
        ev.sort()
        # kernels launched k times per step (k = calls / steps) keep their last k * N dispatches
        k = max(1, round(len(ev) / (a.last + 5)))
        d = [e - s for s, e in ev[-a.last * k:]]
        rows.append((sum(d), name, len(d), sum(d) / len(d), min(d), max(d)))
    rows.sort(reverse=True)
    print("Name,TimedCalls,TotalDurationNs,AverageNs,MinNs,MaxNs")
    for tot, name, n, avg, mn, mx in rows:
        print('"%s",%d,%d,%.1f,%d,%d' % (name.replace('"', "'"), n, tot, avg, mn, mx))


if __name__ == "__main__":
    main()
