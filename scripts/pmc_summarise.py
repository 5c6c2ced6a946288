"""Summarise rocprofv3 --pmc passes (one counter_collection.csv per pass) into a per-kernel table and, for the
TCC FETCH_SIZE / WRITE_SIZE passes, the HBM-side traffic file bench.py reads (profiles/pmc_traffic.json).

    python scripts/pmc_summarise.py --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write \
        --workload c3_wide --out profiles/r01_pmc_hbm_traffic_c3_wide.txt --json profiles/pmc_traffic.json
    python scripts/pmc_summarise.py --sq gpurun_out/pmc_sq --out profiles/r01_pmc_sq_gemm_c3_wide.txt

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE count
kilobytes per dispatch; on gfx950 FETCH_SIZE reports half of the bytes moved by 16-byte-per-lane streaming loads, so
the read side is doubled ("x2").  traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch (median over launches).
"""
import argparse
import collections
import csv
import glob
import json
import os
import statistics


def load(dirname):
    rows = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> counter -> dispatch -> value
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                d = rows[r["Kernel_Name"]][r["Counter_Name"]]
                k = (path, r["Dispatch_Id"])
                d[k] = d.get(k, 0.0) + float(r["Counter_Value"])  # per-XCD / per-instance rows add up
    for path in glob.glob(os.path.join(dirname, "**", "*results.db"), recursive=True):  # rocpd (SQLite) output
        import sqlite3
        con = sqlite3.connect(path)
        for name, counter, disp, value in con.execute(
                "select kernel_name, counter_name, dispatch_id, value from counters_collection"):
            d = rows[name][counter]
            k = (path, disp)
            d[k] = d.get(k, 0.0) + float(value)
    return rows


def kernel_stats(dirname, out, split_by_grid=True):
    """rocprofv3 --kernel-trace --stats -> per-kernel calls / total / average / min / max / percentage as CSV."""
    import sqlite3
    agg = collections.defaultdict(list)
    for path in glob.glob(os.path.join(dirname, "**", "*results.db"), recursive=True):
        con = sqlite3.connect(path)
        for name, dur in con.execute("select name, duration from kernels"):
            agg[name].append(dur)
    for path in glob.glob(os.path.join(dirname, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                # one row per (kernel, launch geometry): a kernel launched on several shapes (the trunk GEMM on the block
                # shape and on the input layer's) gets one line per shape, so averages are per shape
                name = r["Kernel_Name"]
                if split_by_grid:
                    name += "  [grid %sx%sx%s]" % (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1),
                                                   int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1),
                                                   int(r["Grid_Size_Z"]) // max(int(r["Workgroup_Size_Z"]), 1))
                agg[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = float(sum(sum(v) for v in agg.values())) or 1.0
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        # MedianNs / MainShapeAverageNs: a kernel whose launches share a grid but not a problem size (the trunk's forward
        # GEMM runs the K = 64 input layer on the same 8192-tile grid as the K = 2048 blocks) is summarised by its median
        # and by the average over the launches within a factor two of the median, i.e. over its main shape only
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "MedianNs",
                    "MainShapeCalls", "MainShapeAverageNs"])
        for k in sorted(agg, key=lambda k: -sum(agg[k])):
            v = agg[k]
            m = statistics.median(v)
            main = [t for t in v if 0.5 * m <= t <= 2.0 * m]
            w.writerow([k, len(v), sum(v), "%.1f" % (sum(v) / len(v)), "%.3f" % (100.0 * sum(v) / total), min(v), max(v),
                        "%.1f" % m, len(main), "%.1f" % (sum(main) / len(main))])
    for k in sorted(agg, key=lambda k: -sum(agg[k]))[:8]:
        m = statistics.median(agg[k])
        main = [t for t in agg[k] if 0.5 * m <= t <= 2.0 * m]
        print("%-80s n=%4d avg %10.1f us  main shape n=%4d avg %10.1f us  %5.1f%%" %
              (k[:80], len(agg[k]), sum(agg[k]) / len(agg[k]) / 1e3, len(main), sum(main) / len(main) / 1e3,
               100.0 * sum(agg[k]) / total))


def med(d):
    return statistics.median(d.values()) if d else 0.0


# bench.py's roofline names -> substring of the kernel symbol that runs it in the training step
ROLE = {
    # (the wide configuration's forward and dX GEMMs run on the 128x256-tile kernel; DUN_GEMM_WIDE=0: on dma4)
    "gemm_fwd": ("gemm_f32_wide_kernel<true, true>", "gemm_f32_dma4_kernel<true, true>"),
    "gemm_dx": ("gemm_f32_wide_kernel<true, false>", "gemm_f32_dma4_kernel<true, false>"),  # (only with DUN_DX_TRANSPOSED=0)
    "gemm_dw": "gemm_f32_dma4_kernel<false, false>",
    "head_fwd": "head_fwd_kernel",
    "head_bwd": "head_bwd_kernel",
    "elbo": "ll_gauss_kernel",
    "bn_apply": "bn_apply_kernel",
    "bn_apply_head": "bn_apply_head_kernel",
    "head_obj": "head_obj_kernel",
    "bn_bwd_apply": "bn_bwd_apply_kernel",
    "bn_bwd_reduce": "bn_bwd_reduce_kernel",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq")
    ap.add_argument("--kt", help="directory of a --kernel-trace --stats run; writes the per-kernel CSV to --out")
    ap.add_argument("--workload", default="c3_wide")
    ap.add_argument("--out", required=True)
    ap.add_argument("--json")
    ap.add_argument("--header", default="")
    a = ap.parse_args()
    if a.kt:
        kernel_stats(a.kt, a.out)
        return
    lines = []
    if a.header:
        lines.append(a.header)
    if a.fetch and a.write:
        fe, wr = load(a.fetch), load(a.write)
        lines.append("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes). Counter unit KB; per dispatch "
                     "(median over n launches).")
        lines.append("gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads 1/2 of the bytes of 16-B/lane "
                     "streaming loads -> x2 column.  traffic = 2*FETCH + WRITE.")
        lines.append("")
        # A kernel symbol that runs two different jobs in the step (gemm_f32_wide_kernel<true, true>: the forward GEMM and,
        # against the transposed weights, the dX GEMM with its residual / Z reads) has two clusters of FETCH_SIZE: split
        # its dispatches at the midpoint (the write pass is joined on the dispatch id: same program, same launch order)
        for k in list(fe):
            d = fe[k].get("FETCH_SIZE", {})
            if len(d) >= 20 and min(d.values()) > 0 and max(d.values()) / min(d.values()) > 1.25:
                vals = sorted(d.values())
                core = vals[len(vals) // 10: len(vals) - len(vals) // 10]  # (ignore odd launches such as the input layer)
                cut = 0.5 * (core[0] + core[-1])
                if core[-1] / core[0] > 1.25:
                    for tag, keep in (("  [less read: forward]", lambda v: v < cut), ("  [more read: dX]", lambda v: v >= cut)):
                        ids = {kk for kk, v in d.items() if keep(v)}
                        fe[k + tag] = {"FETCH_SIZE": {kk: d[kk] for kk in ids}}
                        dw = wr.get(k, {}).get("WRITE_SIZE", {})
                        by_id = {kk[1]: v for kk, v in dw.items()}
                        wr[k + tag] = {"WRITE_SIZE": {kk: by_id[kk[1]] for kk in ids if kk[1] in by_id}}
        table = {}
        for k in sorted(fe, key=lambda k: -med(fe[k].get("FETCH_SIZE", {})) * len(fe[k].get("FETCH_SIZE", {}))):
            f = med(fe[k].get("FETCH_SIZE", {}))
            w = med(wr.get(k, {}).get("WRITE_SIZE", {}))
            n = len(fe[k].get("FETCH_SIZE", {}))
            table[k] = (2 * f + w) * 1024.0
            lines.append("%-72s n=%4d  FETCH_SIZE med %9.1f MB (x2 = %9.1f MB)  WRITE_SIZE med %8.1f MB  traffic %9.1f MB"
                         % (k[:72], n, f / 1024, 2 * f / 1024, w / 1024, (2 * f + w) / 1024))
        if a.json:
            cur = {}
            if os.path.exists(a.json):
                cur = json.load(open(a.json))
            entry = {}
            for role, subs in ROLE.items():
                for sub in ((subs,) if isinstance(subs, str) else subs):
                    hits = [v for k, v in table.items() if sub in k and "[" not in k]
                    if hits:
                        entry[role] = hits[0]
                        break
            for role, tag in (("gemm_fwd", "[less read: forward]"), ("gemm_dx", "[more read: dX]")):
                hits = [v for k, v in table.items() if tag in k]
                if hits:
                    entry[role] = hits[0]
            cur[a.workload] = entry
            cur["_note"] = ("bytes per launch, (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes "
                            "over `bench.py --steps 1 --warmup 0`; written by scripts/pmc_summarise.py")
            json.dump(cur, open(a.json, "w"), indent=1, sort_keys=True)
    if a.sq:
        sq = load(a.sq)
        lines.append("rocprofv3 --pmc SQ pass; per dispatch medians.")
        for k in sorted(sq, key=lambda k: -sum(med(c) * len(c) for c in sq[k].values())):
            c = sq[k]
            n = max(len(v) for v in c.values())
            parts = ["%s=%.4g" % (name, med(v)) for name, v in sorted(c.items())]
            derived = []
            if "SQ_BUSY_CU_CYCLES" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c and med(c["SQ_BUSY_CU_CYCLES"]) > 0:
                derived.append("mfma_busy/cu_busy=%.3f" % (med(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / med(c["SQ_BUSY_CU_CYCLES"])))
            if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c and med(c["GRBM_GUI_ACTIVE"]) > 0:
                cyc = med(c["GRBM_GUI_ACTIVE"]) / 8.0  # summed over the 8 XCDs
                derived.append("cycles=%.4g  mfma_busy/(1024 SIMD*cycles)=%.3f"
                               % (cyc, med(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (1024.0 * cyc)))
            if "SQ_WAVE_CYCLES" in c and med(c["SQ_WAVE_CYCLES"]) > 0:
                wc = med(c["SQ_WAVE_CYCLES"])
                derived.append("of wave cycles: " + " ".join(
                    "%s %.2f" % (n_, med(c[n_]) / wc) for n_ in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")
                    if n_ in c))
            if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c and med(c["SQ_LDS_IDX_ACTIVE"]) > 0:
                derived.append("lds_conflict/active=%.4f" % (med(c["SQ_LDS_BANK_CONFLICT"]) / med(c["SQ_LDS_IDX_ACTIVE"])))
            lines.append("%s  n=%d" % (k[:100], n))
            lines.append("    " + "  ".join(parts))
            if derived:
                lines.append("    derived: " + "  ".join(derived))
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main()
