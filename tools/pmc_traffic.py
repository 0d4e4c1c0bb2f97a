"""Per-kernel HBM traffic from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as
MI355X_MICROARCH.md's HBM section prescribes), averaged per launch and corrected for gfx950:
FETCH_SIZE under-reports wide coalesced reads by 2x  ->  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.

Usage: python tools/pmc_traffic.py <out.json> <counter_collection.csv> [<counter_collection.csv> ...]
"""
import collections
import csv
import json
import re
import sys


# The guide's x2 is calibrated on wide coalesced STREAMING reads and says to calibrate other patterns on a known byte count.
# Round 2 assumed the fused first layer's gather (LDS-DMA, 8 lanes x 16 B per 128-byte line, structured buffer resource, random rows)
# was tallied in full because the raw counter happened to equal that kernel's byte budget.  Round 3 calibrated exactly that
# instruction and lane mapping on a known byte count (tools/exp/ldsdma_fetch_calib.hip, profiles/r03_fetch_calib.json): FETCH_SIZE
# reports 0.500 of the bytes for the gather, as for the plain streaming read -- the x2 applies to every kernel here.
FETCH_FACTOR = {}
FETCH_FACTOR_BASIS = {}


def short(name):
    m = re.search(r"(?:\)::|::|^)([A-Za-z_0-9]+)(<[^(]*>)?\(", name)
    base = m.group(1) if m else name[:40]
    targs = (m.group(2) or "") if m else ""
    return base + targs.replace(" ", "")


def main():
    out_path, csvs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in csvs:
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if "at::native" in k or "rocclr" in k or "rocprim" in k.lower():
                continue
            agg[short(k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels = {}
    for k, d in sorted(agg.items()):
        f = d.get("FETCH_SIZE")
        w = d.get("WRITE_SIZE")
        if not f or not w:
            continue
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        ff = FETCH_FACTOR.get(k, 2)
        kernels[k] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "launches": len(f), "fetch_factor": ff,
                      "hbm_bytes_corrected": (ff * fk + wk) * 1024}
        if k in FETCH_FACTOR:
            kernels[k]["fetch_factor_basis"] = FETCH_FACTOR_BASIS[k]
    note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --steps 4 --warmup 2 "
            "--no-cpu-baseline` (MI355X). Units KB (1024 B) per launch, averaged over launches. gfx950 correction "
            "(MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of wide coalesced reads -> hbm_bytes = "
            "(2*FETCH_SIZE + WRITE_SIZE) * 1024. Calibration: hash_bucket_i64_kernel reads 13.63 MB and writes 13.63 MB.")
    # bench.py event name -> profiled kernel (with DR_FUSE_K3=0 the forward and the dgrad of the first layer are the same kernel
    # instantiation: their traffic is then the average of the two; names absent from the profile are dropped by bench.py)
    # (round 4: the kernels are templates over the operand mode -- <..., 1> = f16x2, the default; the first name present wins)
    def pick(*names):
        return next((n for n in names if n in kernels), names[0])
    event_names = {"emb_linear_fwd_L0": pick("bf3_emb_linear_kernel<1>", "bf3_emb_linear_kernel<0>", "bf3_emb_linear_kernel"),
                   "emb_pool_fwd": "emb_pool_fwd_sv_kernel<16,8>",
                   "emb_pool_bwd": "emb_bwd_sorted_kernel<16,4,false>", "hash_bucket_i64": "hash_bucket_i64_kernel",
                   "emb_pool_bwd_fused_dgrad_L0": "h2_occ_nt_kernel<8,0>",
                   "linear_fwd_L0": pick("bf3_gemm_rs_kernel<0,0,1,1,1>", "bf3_gemm_rs_kernel<0,0,1,1,0>", "bf3_gemm_rs_kernel<0,0,1,1>"),
                   "linear_bwd_dx_L0": pick("bf3_gemm_rs_kernel<0,0,1,1,1>", "bf3_gemm_rs_kernel<0,0,1,1,0>", "bf3_gemm_rs_kernel<0,0,1,1>"),
                   "tower_tail_fused": pick("tower_tail_fused_kernel<8>", "tower_tail_fused_kernel<4>"),
                   "linear_bwd_dw_L0": pick("bf3_gemm_tn_rs_kernel<1,1>", "bf3_gemm_tn_rs_kernel<1,0>", "bf3_gemm_tn_rs_kernel<1>")}
    h2 = any(n.endswith(",1,1,1>") or n in ("bf3_emb_linear_kernel<1>", "bf3_gemm_tn_rs_kernel<1,1>") for n in event_names.values() if n in kernels)
    import hashlib
    import os
    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "deep_recommenders_amd", "lib", "libdr_hotpath.so")
    lib_sha = hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None
    json.dump({"_note": note, "kernels": kernels, "event_names": event_names, "gemm_split": "f16x2" if h2 else "bf16x3",
               "lib_sha256": lib_sha},
              open(out_path, "w"), indent=1, sort_keys=True)
    for k, v in kernels.items():
        print("%-60s %10.1f MB" % (k, v["hbm_bytes_corrected"] / 1e6))


if __name__ == "__main__":
    main()
