"""Summarise rocprofv3 rocpd (.db) outputs of tools/profile_gpu.sh into profiles/<tag>_*.{md,csv}.
usage: python tools/summarize_prof.py <tag> [timed_steps] [cohorts] [workgroups_per_cu]     (one launch = one step of one cohort)"""
import glob, json, os, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cohorts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
timed = steps * cohorts                      # launches inside bench.py's timed region
envs_per_launch = 4096 // cohorts
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
lines = [f"# rocprofv3 summary `{tag}` — `python bench.py --steps {steps} --warmup 400 --no-cpu-baseline` (S24, 4096 envs, 1 MI355X)", "",
         f"One launch of `mjh_step_kernel` = one step of one cohort = {envs_per_launch} environments ({cohorts} cohorts on separate HIP streams, launches overlap).", ""]

def db(path):
    f = glob.glob(os.path.join(src, path, "*.db"))
    return sqlite3.connect(f[0]) if f else None

con = db("trace")
kern_avg_us = None
if con:
    lines += ["## `rocprofv3 --kernel-trace --stats` (all launches, incl. the 400 warm-up/settle steps)", "",
              "| kernel | calls | total (us) | average (us) | % |", "|---|---|---|---|---|"]
    csv = ["name,calls,total_us,average_us,percentage"]
    for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append(f"| `{r[0]}` | {r[1]} | {r[2]:.1f} | {r[3]:.2f} | {r[4]:.3f} |")
        csv.append(",".join(['"%s"' % r[0]] + [str(x) for x in r[1:]]))
    open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w").write("\n".join(csv) + "\n")
    rows = con.execute("select k.start, k.end, k.vgpr_count, k.accum_vgpr_count, k.sgpr_count, k.lds_size, k.scratch_size from kernels k where k.name like '%mjh_step_kernel%' order by k.start").fetchall()
    if rows:
        last = rows[-timed:]
        kern_avg_us = sum((e - s) for s, e, *_ in last) / len(last) / 1e3
        lines += ["", f"Average duration of `mjh_step_kernel` over the LAST {len(last)} launches (the timed region of bench.py): **{kern_avg_us:.1f} us**",
                  f"(dispatch resources: VGPR {rows[-1][2]}, AGPR {rows[-1][3]}, SGPR {rows[-1][4]}, LDS {rows[-1][5]} B/workgroup, scratch {rows[-1][6]} B/lane)", ""]
tot = {}
for sub, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    con = db(sub)
    if not con:
        continue
    rows = con.execute("select value from counters_collection where kernel_name like '%mjh_step_kernel%' and counter_name=? order by start", (name,)).fetchall()
    if rows:
        last = [r[0] for r in rows[-timed:]]
        tot[name] = sum(last) / len(last)
if tot:
    f, w = tot.get("FETCH_SIZE", 0.0), tot.get("WRITE_SIZE", 0.0)
    lines += ["## HBM traffic (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate passes; units of 1 KiB; last %d launches)" % timed, "",
              f"- FETCH_SIZE = {f:.1f} KiB/launch, WRITE_SIZE = {w:.1f} KiB/launch",
              f"- raw (FETCH+WRITE)*1024 = {(f + w) * 1024 / 1e6:.2f} MB/launch; with the gfx950 x2 read correction for wide coalesced reads (MI355X_MICROARCH.md §HBM): {(2 * f + w) * 1024 / 1e6:.2f} MB/launch",
              f"- algorithmic bytes = 800 B/env-step x {envs_per_launch} envs = {800 * envs_per_launch / 1e6:.2f} MB/launch", ""]
    lines += ["Calibration of FETCH_SIZE for THIS access pattern (4 B/lane rows of <=128 B per wave): the bytes one launch must read are",
              f"state rows 592 B + per-env parameter tables 360 B + time 4 B = 956 B/env -> {956 * envs_per_launch / 1e6:.2f} MB for {envs_per_launch} envs, plus the shared model tables;",
              "FETCH_SIZE reports about that figure un-doubled, so the x2 wide-read correction does not apply here and `roofline.traffic`",
              "records the RAW (FETCH+WRITE)*1024 bytes.", ""]
    json.dump({"tag": tag, "fetch_kib": f, "write_kib": w, "envs_per_launch": envs_per_launch, "bytes_per_launch": (f + w) * 1024, "bytes_per_launch_x2_read_bound": (2 * f + w) * 1024,
               "note": "rocprofv3 PMC, separate passes; raw FETCH_SIZE+WRITE_SIZE (4 B/lane row reads calibrate 1:1 against the known mandatory reads)"},
              open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
sqvals = {}
con = db("pmc_sq")
if con:
    for n in ("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU"):
        rows = con.execute("select value from counters_collection where kernel_name like '%mjh_step_kernel%' and counter_name=? order by start", (n,)).fetchall()
        last = [r[0] for r in rows[-timed:]]
        if last:
            sqvals[n] = sum(last) / len(last)
    tp = os.path.join(dst, "hbm_traffic.json")
    if os.path.exists(tp) and "SQ_ACTIVE_INST_VALU" in sqvals and "SQ_WAVE_CYCLES" in sqvals:
        tj = json.load(open(tp))
        wg_per_cu = int(sys.argv[4]) if len(sys.argv) > 4 else 9          # LDS-limited resident workgroups per CU (tools/timeline.py)
        per_wave = sqvals["SQ_ACTIVE_INST_VALU"] / sqvals["SQ_WAVE_CYCLES"]
        tj.update({"valu_active_frac_per_wave": per_wave, "workgroups_per_cu": wg_per_cu, "valu_issue_busy": per_wave * wg_per_cu / 4.0,
                   "valu_instructions_per_env_step": sqvals.get("SQ_INSTS_VALU", 0.0) / envs_per_launch})
        json.dump(tj, open(tp, "w"), indent=1)
if con:
    lines += ["## SQ counters (per launch, averages over the last %d launches)" % timed, "", "| counter | value | per wave |", "|---|---|---|"]
    names = [r[0] for r in con.execute("select distinct counter_name from counters_collection")]
    for n in sorted(names):
        rows = con.execute("select value from counters_collection where kernel_name like '%mjh_step_kernel%' and counter_name=? order by start", (n,)).fetchall()
        last = [r[0] for r in rows[-timed:]]
        if last:
            v = sum(last) / len(last)
            lines.append(f"| {n} | {v:.4g} | {v / envs_per_launch:.4g} |")
    lines.append("")
for jf in ("bench_trace.json",):
    p = os.path.join(src, jf)
    if os.path.exists(p):
        txt = open(p).read().strip().splitlines()
        if txt:
            lines += ["## bench.py line of the traced run", "", "```", txt[-1], "```", ""]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines))
