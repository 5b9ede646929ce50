#!/usr/bin/env python
"""Turn the rocprofv3 result databases under gpurun_out/prof/ into the small text summaries kept in profiles/.
usage: python scripts/summarize_profile.py gpurun_out/prof profiles/r01_wave_kernel"""
import glob
import os
import sqlite3
import sys


def rows(path, sql):
    db = sqlite3.connect(path)
    cur = db.cursor()
    r = cur.execute(sql).fetchall()
    cols = [d[0] for d in cur.description]
    db.close()
    return cols, r


def main(src, dst, key="carlike_n50_B1024_c4"):
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    out = []
    tr = os.path.join(src, "trace", "run_results.db")
    if os.path.exists(tr):
        out.append("## rocprofv3 --kernel-trace --stats  (python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs --no-parity-check)\n")
        out.append("| kernel | calls | total_us | avg_us | % |\n|---|---|---|---|---|\n")
        for name, calls, tot, avg, pct in rows(tr, "select name,total_calls,total_duration,average,percentage from top_kernels")[1]:
            short = name.replace("(anonymous namespace)::", "").split("(mpc::")[0][:100]
            out.append(f"| {short} | {calls} | {tot:.0f} | {avg:.0f} | {pct:.3f} |\n")
        c, r = rows(tr, "select name,duration,grid_x,workgroup_x,lds_size,scratch_size,vgpr_count,accum_vgpr_count,sgpr_count from kernels where name like '%mpc_ipm%'")
        out.append("\nper-dispatch of the solve kernel (duration ns, grid, workgroup, LDS B, scratch B/lane, VGPR, AGPR, SGPR):\n\n")
        for x in r:
            out.append("    " + ", ".join(str(v) for v in x[1:]) + "\n")
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        db = os.path.join(d, "run_results.db")
        if not os.path.exists(db):
            continue
        c, r = rows(db, "select counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
                        "where kernel_name like '%mpc_ipm%' group by counter_name")
        out.append(f"\n## rocprofv3 --pmc ({os.path.basename(d)}) -- per dispatch of the solve kernel\n\n| counter | dispatches | mean | min | max |\n|---|---|---|---|---|\n")
        for name, cnt, mean, mn, mx in r:
            out.append(f"| {name} | {cnt} | {mean:.6g} | {mn:.6g} | {mx:.6g} |\n")
    # HBM traffic record for bench.py's roofline.traffic: FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE counts 64 B per 128-B
    # request on gfx950 (MI355X_MICROARCH.md, HBM section) -> x2.  WRITE_SIZE is uncalibrated for partial (8-byte) stores.
    vals = {}
    for d in ("pmc_fetch", "pmc_write"):
        db = os.path.join(src, d, "run_results.db")
        if os.path.exists(db):
            for name, mean in rows(db, "select counter_name, avg(value) from counters_collection where kernel_name like "
                                       "'%mpc_ipm%' group by counter_name")[1]:
                vals[name] = mean
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        import json
        rec = {"kernel": "mpc_ipm_wave_kernel", "key": key, "fetch_size_kib": vals["FETCH_SIZE"], "write_size_kib": vals["WRITE_SIZE"],
               "bytes_per_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
               "source": os.path.basename(dst) + ".md", "command": "python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs --no-parity-check"}
        path = os.path.join(os.path.dirname(dst) or ".", "hbm_traffic.json")
        try:
            allrec = json.load(open(path))
            if not isinstance(allrec.get(key, {}), dict) or "kernel" in allrec:
                allrec = {}
        except (OSError, ValueError):
            allrec = {}
        allrec[key] = rec
        json.dump(allrec, open(path, "w"), indent=1)
        out.append(f"\nHBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE = {rec['bytes_per_launch'] / 1e6:.2f} MB\n")
    # every PMC mean of the solve kernel, per workload key: what bench.py's roofline line reads next to the live numbers (profiles/pmc_counters.json)
    allc = {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        db = os.path.join(d, "run_results.db")
        if os.path.exists(db):
            for name, mean in rows(db, "select counter_name, avg(value) from counters_collection where kernel_name like '%mpc_ipm%' group by counter_name")[1]:
                allc[name] = mean
    tr = os.path.join(src, "trace", "run_results.db")
    if os.path.exists(tr):
        r = rows(tr, "select avg(duration), count(*), max(lds_size), max(vgpr_count), max(accum_vgpr_count) from kernels where name like '%mpc_ipm%'")[1]
        if r and r[0][0]:
            allc["kernel_avg_ns"], allc["dispatches"], allc["lds_bytes"] = r[0][0], r[0][1], r[0][2]
    if allc:
        import json
        path = os.path.join(os.path.dirname(dst) or ".", "pmc_counters.json")
        try:
            store = json.load(open(path))
        except (OSError, ValueError):
            store = {}
        store[key] = dict(allc, source=os.path.basename(dst) + ".md")
        json.dump(store, open(path, "w"), indent=1, sort_keys=True)
    open(dst + ".md", "w").write("".join(out))
    print("".join(out))


if __name__ == "__main__":
    main(*sys.argv[1:4])
