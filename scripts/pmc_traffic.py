#!/usr/bin/env python3
"""Aggregate the two rocprofv3 --pmc passes of scripts/pmc_traffic.sh (FETCH_SIZE, WRITE_SIZE; counter
unit KB per dispatch) into bytes per launch and per step.  gfx950 correction (MI355X_MICROARCH.md, HBM
section): FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE as is; both
count the L2s' fabric requests, Infinity-Cache hits included.
usage: pmc_traffic.py fetch.csv write.csv n_steps out.json"""
import csv
import json
import re
import sys

SPARSE = ("k_sort_columns", "k_dedup_", "k_finish_catchup", "k_emb_fm_fwd", "k_emb_fm_bwd", "k_sparse_update_multi",
          "k_rs_", "k_build_keys", "k_scatter_unique", "k_catchup_rows", "k_adam_catchup", "k_sparse_adam",
          "k_emb_fused")
GEMM = ("k_gemm_f32_pipe", "k_gemm_f32_pair", "k_gemm_f32_multi", "k_gemm_f32<", "k_gemm_x6")


def load(path, counter):
    agg = {}
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"]
        if "(" in name and not name.startswith("void"):
            name = name.split("(")[0]
        name = re.sub(r"^void ", "", name).split("(")[0]
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"]) * 1024.0
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    steps = float(sys.argv[3])
    rows = []
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        n = max(fetch.get(k, [0, 0])[0], write.get(k, [0, 0])[0])
        f = fetch.get(k, [1, 0.0])
        w = write.get(k, [1, 0.0])
        rows.append((k, n, 2.0 * f[1] / max(f[0], 1), w[1] / max(w[0], 1)))
    print("%-58s %8s %12s %12s %12s" % ("kernel", "launches", "FETCH x2 MB", "WRITE MB", "total MB"))
    per_kernel, sparse_step, gemm_bytes, gemm_n = {}, 0.0, 0.0, 0
    for k, n, f, w in rows:
        print("%-58s %8d %12.2f %12.2f %12.2f" % (k[:58], n, f / 1e6, w / 1e6, (f + w) / 1e6))
        base = k.split("<")[0]
        per_kernel[base] = per_kernel.get(base, 0.0) + (f + w) * n / max(sum(
            r[1] for r in rows if r[0].split("<")[0] == base), 1)
        if k.startswith(SPARSE):
            sparse_step += (f + w) * n / steps
        if k.startswith(GEMM):
            gemm_bytes += (f + w) * n
            gemm_n += n
    out = {"source": "scripts/pmc_traffic.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                     "FETCH doubled per the gfx950 note)", "includes_infinity_cache_hits": True,
           "steps": steps, "per_kernel_bytes_per_launch": per_kernel,
           "sparse_traffic_bytes_per_step": sparse_step,
           "traffic_bytes_per_launch": gemm_bytes / max(gemm_n, 1),
           "kernel": "mean over the MFMA GEMM launches of the DeepFM step"}
    print("sparse path: %.2f MB per step; GEMM mean %.2f MB per launch over %d launches"
          % (sparse_step / 1e6, out["traffic_bytes_per_launch"] / 1e6, gemm_n))
    with open(sys.argv[4], "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
