"""Per-kernel SQ stall picture from one rocprofv3 --pmc pass (tools/collect_sq_pmc.sh).

MI355X_MICROARCH.md: SQ_WAIT_ANY (wave parked on s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall: MFMA dependency / pipe)
+ SQ_ACTIVE_INST_ANY ~= SQ_WAVE_CYCLES (disjoint, quad-cycles); SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per 32x32x16 bf16
MFMA); SQ_LDS_BANK_CONFLICT = extra LDS cycles out of SQ_LDS_IDX_ACTIVE."""
import collections
import csv
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.search(r"([A-Za-z_0-9]+)(<[^(]*>)?\(", name)
    return ((m.group(1) + (m.group(2) or "")) if m else name)[:44]


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if "at::native" in k or "rocclr" in k or "rocprim" in k.lower():
                continue
            agg[short(k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("%-44s %5s %8s %8s %8s %10s %9s" % ("kernel (mean per launch)", "n", "wait%", "stall%", "active%", "mfma_busy/busy", "lds_confl%"))
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
        mean = {c: sum(v) / len(v) for c, v in d.items()}
        wc = mean.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        busy = mean.get("SQ_BUSY_CYCLES", 0.0)
        lds = mean.get("SQ_LDS_IDX_ACTIVE", 0.0)
        print("%-44s %5d %8.1f %8.1f %8.1f %10.3f %9.1f" % (
            k, len(d["SQ_WAVE_CYCLES"]), 100 * mean.get("SQ_WAIT_ANY", 0) / wc, 100 * mean.get("SQ_WAIT_INST_ANY", 0) / wc,
            100 * mean.get("SQ_ACTIVE_INST_ANY", 0) / wc, mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / busy if busy else float("nan"),
            100 * mean.get("SQ_LDS_BANK_CONFLICT", 0) / lds if lds else float("nan")))
        print("    raw: " + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(mean.items())))


if __name__ == "__main__":
    main()
