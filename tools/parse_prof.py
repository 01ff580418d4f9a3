#!/usr/bin/env python3
"""Summarise rocprofv3 output (gpurun_out/prof/*) into profiles/: a kernel-stats table and the
per-launch HBM traffic of each kernel from the PMC passes.

PMC handling follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are reported in KiB, collected in
SEPARATE passes (they do not fit one), and on gfx950 FETCH_SIZE reads exactly 1/2 of the bytes of a wide
(16 B/lane) coalesced streaming read, so the read side is doubled; WRITE_SIZE is exact for 16 B/lane stores.

usage: python tools/parse_prof.py gpurun_out/prof profiles r01
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

OURS = ("nn_", "corr_", "finalize", "pack_", "preshape", "rot_", "pose_", "transform", "grid_", "gridb_", "cell_", "row_", "sum_", "fps_", "resident_", "cand_", "aivs_")


def short(name):
    n = name.replace("void ", "")
    n = n.split("(")[0]
    return n.replace("kss::", "")


def read_counters(d):
    """returns {kernel: {counter: [values per dispatch]}}"""
    out = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            # the fused pass at C4 size (1M sources: 1954 workgroups of 512) is kept apart from the same symbol's C2-size launches
            try:
                if k.startswith("grid_pass_kernel") and int(float(r.get("Grid_Size", 0) or 0)) >= 900000:
                    k = k.replace("grid_pass_kernel", "grid_pass_c4_kernel", 1)
            except ValueError:
                pass
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    cmd = sys.argv[4] if len(sys.argv) > 4 else "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
    os.makedirs(dst, exist_ok=True)
    lines = ["# rocprofv3 summary %s" % tag, "",
             "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- %s`" % cmd, "",
             "| kernel | calls | total ms | avg us | % | min us | max us |", "|---|---|---|---|---|---|---|"]
    stats = {}
    for f in glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            stats[k] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"])}
            lines.append("| `%s` | %s | %.3f | %.2f | %s | %.2f | %.2f |" % (k, r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                         float(r["AverageNs"]) / 1e3, r["Percentage"], float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    summary = {}
    fetch = read_counters(os.path.join(src, "pmc_fetch"))
    write = read_counters(os.path.join(src, "pmc_write"))
    sq = read_counters(os.path.join(src, "pmc_sq"))
    for k, v in read_counters(os.path.join(src, "pmc_tcc")).items():
        for c2, vals in v.items():
            sq[k][c2] = vals
    lines += ["", "## HBM traffic per launch (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes)", "",
              "FETCH_SIZE and WRITE_SIZE are KiB; read side x2 (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section).", "",
              "| kernel | launches | FETCH_SIZE KiB (raw avg) | read bytes (x2 corrected) | WRITE_SIZE KiB (avg) | write bytes | HBM bytes / launch |",
              "|---|---|---|---|---|---|---|"]
    for k in sorted(set(list(fetch) + list(write))):
        if not k.startswith(OURS):
            continue
        fv = fetch.get(k, {}).get("FETCH_SIZE", [])
        wv = write.get(k, {}).get("WRITE_SIZE", [])
        fa = sum(fv) / len(fv) if fv else 0.0
        wa = sum(wv) / len(wv) if wv else 0.0
        rb, wb = fa * 1024 * 2, wa * 1024
        key = k.split("<")[0].replace("_kernel", "")
        targs = [t.strip() for t in k[k.find("<") + 1:k.rfind(">")].split(",")] if "<" in k else []
        if key == "grid_pass" and len(targs) == 5 and targs[4] == "true":
            key = "grid_pass_chain"   # chained launches: one launch = many ICP passes (bench.py divides by its passes per launch)
        if key in summary and summary[key].get("launches", 0) > len(fv):
            continue   # template variants share a key: keep the one launched most (the ICP-iteration form)
        summary[key] = {"kernel": k, "launches": len(fv), "fetch_size_kib_raw": fa, "read_bytes_corrected": rb,
                        "write_size_kib": wa, "write_bytes": wb, "hbm_bytes_per_launch": rb + wb}
        if k in stats:
            summary[key]["avg_us"] = stats[k]["avg_ns"] / 1e3
        lines.append("| `%s` | %d | %.1f | %.3e | %.1f | %.3e | %.3e |" % (k, len(fv), fa, rb, wa, wb, rb + wb))
    if sq:
        lines += ["", "## SQ counters (separate pass), per launch averages", ""]
        for k in sorted(sq):
            if not k.startswith(("nn_", "corr_", "grid_", "gridb_", "cell_", "resident_", "cand_")):
                continue
            lines.append("`%s`:" % k)
            for c, v in sorted(sq[k].items()):
                lines.append("- %s = %.4g" % (c, sum(v) / len(v)))
            d = {c: sum(v) / len(v) for c, v in sq[k].items()}
            key = k.split("<")[0].replace("_kernel", "")
            if summary.get(key, {}).get("kernel", k) == k:
                summary.setdefault(key, {})["sq"] = d
    open(os.path.join(dst, "%s_rocprof_summary.md" % tag), "w").write("\n".join(lines) + "\n")
    json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
    for f in glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
        open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w").write(open(f).read())
    print("\n".join(lines))


if __name__ == "__main__":
    main()
