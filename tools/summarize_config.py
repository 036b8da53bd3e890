"""Summarise the rocprofv3 rocpd (.db) outputs of tools/profile_config.sh for one bench.py config.
usage: python tools/summarize_config.py <raw dir> <out dir> <tag> <config>
One "launch" of the bench line = one step of one cohort: a single mjh_step_kernel launch in the LDS-resident layout, the three
launches assemble (mjh_step_kernel) -> mjh_solve_kernel -> integrate (mjh_step_kernel) in the many-body layout."""
import glob, json, os, sqlite3, sys

raw, dst, tag, cfg = sys.argv[1:5]
os.makedirs(dst, exist_ok=True)
bench = None
bp = os.path.join(raw, "bench_trace.json")
if os.path.exists(bp):
    txt = [l for l in open(bp).read().strip().splitlines() if l.startswith("{")]
    if txt:
        bench = json.loads(txt[-1])
steps = bench["steps"] if bench else 100
cohorts = bench["config"]["cohorts"] if bench else 2
nenv = bench["config"]["envs_per_gpu"] if bench else 4096
timed = bench["roofline"]["launches"] if bench else steps * cohorts
envs_per_launch = bench["roofline"]["envs_per_launch"] if bench else nenv / cohorts
alg = bench["roofline"]["algorithmic_bytes_per_env_step"] if bench else 0
lines = [f"# rocprofv3 summary `{tag}` / config `{cfg}` — `python bench.py --config {cfg} --steps {steps} --warmup 20 --no-cpu-baseline --no-second-window` (1 MI355X)", "",
         f"{nenv} envs, {cohorts} cohorts on separate HIP streams; one step launch covers {envs_per_launch:.0f} envs; the timed region holds {timed} step launches.", ""]


def db(sub):
    f = glob.glob(os.path.join(raw, sub, "**", "*.db"), recursive=True)
    return sqlite3.connect(f[0]) if f else None


STEPK = ("mjh_step_kernel", "mjh_solve_kernel")
con = db("trace")
per_step_us = None
if con:
    lines += ["## `rocprofv3 --kernel-trace --stats` (all launches of the run, incl. settle and warm-up)", "", "| kernel | calls | total (us) | average (us) | % |", "|---|---|---|---|---|"]
    csv = ["name,calls,total_us,average_us,percentage"]
    for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append(f"| `{r[0][:110]}` | {r[1]} | {r[2]:.1f} | {r[3]:.2f} | {r[4]:.3f} |")
        csv.append(",".join(['"%s"' % r[0]] + [str(x) for x in r[1:]]))
    open(os.path.join(dst, f"{tag}_{cfg}_kernel_stats.csv"), "w").write("\n".join(csv) + "\n")
    rows = con.execute("select k.name, k.start, k.end, k.vgpr_count, k.sgpr_count, k.lds_size, k.scratch_size from kernels k where k.name like '%mjh_step_kernel%' or k.name like '%mjh_solve_kernel%' or k.name like '%mjh_dense_%' or k.name like '%mjh_solve_mixed%' order by k.start").fetchall()
    if rows:
        per = 3 if any("mjh_solve_kernel" in r[0] for r in rows) else 1
        if any("mjh_dense_" in r[0] for r in rows):
            per = 5        # assemble -> dense build -> dense solve -> block solve (cohorts without a long-sweeping env, envs beyond the capacity) -> integrate
        last = rows[-timed * per:]
        tot = sum(e - s for _, s, e, *_ in last) / 1e3
        per_step_us = tot / max(timed, 1)
        lines += ["", f"Step kernels per launch: {per}.  Summed kernel time per step launch over the LAST {timed} launches (the timed region): **{per_step_us:.1f} us**"]
        seen = {}
        for n, s, e, vg, sg, lds, scr in last:
            k = n.split("(")[0]
            a = seen.setdefault(k, [0, 0.0, vg, sg, lds, scr]); a[0] += 1; a[1] += (e - s) / 1e3
        for k, a in seen.items():
            lines.append(f"- `{k}`: {a[0]} launches, mean {a[1] / a[0]:.1f} us; VGPR {a[2]}, SGPR {a[3]}, LDS {a[4]} B/workgroup, scratch {a[5]} B/lane")
        lines.append("")
tot = {}
for sub, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    con = db(sub)
    if not con:
        continue
    acc = 0.0; cnt = 0
    rows = con.execute("select kernel_name, value from counters_collection where (kernel_name like '%mjh_step_kernel%' or kernel_name like '%mjh_solve_kernel%' or kernel_name like '%mjh_dense_%' or kernel_name like '%mjh_solve_mixed%') and counter_name=? order by start", (name,)).fetchall()
    per = 3 if any("mjh_solve_kernel" in r[0] for r in rows) else 1
    if any("mjh_dense_" in r[0] for r in rows):
        per = 5
    last = rows[-timed * per:]
    tot[name] = sum(v for _, v in last) / max(timed, 1)
    bykern = {}
    for k, v in last:
        kk = k.split("(")[0]; bykern.setdefault(kk, [0, 0.0]); bykern[kk][0] += 1; bykern[kk][1] += v
    tot[name + "_by"] = {k: a[1] / a[0] for k, a in bykern.items()}
if "FETCH_SIZE" in tot or "WRITE_SIZE" in tot:
    f, w = tot.get("FETCH_SIZE", 0.0), tot.get("WRITE_SIZE", 0.0)
    meas = (f + w) * 1024
    lines += [f"## HBM traffic (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate passes; units of 1 KiB; per step launch, last {timed})", "",
              f"- FETCH_SIZE = {f:.1f} KiB, WRITE_SIZE = {w:.1f} KiB per step launch -> raw {(meas) / 1e6:.3f} MB; with the guide's x2 read correction for wide coalesced reads {((2 * f + w) * 1024) / 1e6:.3f} MB",
              f"- per env-step: **{meas / envs_per_launch:.0f} B measured** (x2-read bound {((2 * f + w) * 1024) / envs_per_launch:.0f} B) against **{alg} B algorithmic** (4 (2 nq + 6 nv), SURVEY.md §8-d D5): ratio {meas / envs_per_launch / max(alg, 1):.2f}",
              ]
    for nm in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, v in tot.get(nm + "_by", {}).items():
            lines.append(f"  - {nm} of `{k}`: {v:.1f} KiB per launch of that kernel")
    if per_step_us:
        lines.append(f"- achieved HBM rate of the step kernels: {meas / (per_step_us * 1e-6) / 1e9:.1f} GB/s measured traffic, {alg * envs_per_launch / (per_step_us * 1e-6) / 1e9:.2f} GB/s algorithmic, of 8000 GB/s peak")
    lines.append("")
    json.dump({"tag": f"{tag}_{cfg}", "config": cfg, "fetch_kib": f, "write_kib": w, "envs_per_launch": envs_per_launch, "bytes_per_launch": meas,
               "bytes_per_launch_x2_read_bound": (2 * f + w) * 1024, "bytes_per_env_step": meas / envs_per_launch, "algorithmic_bytes_per_env_step": alg,
               "kernel_us_per_step_launch": per_step_us,
               "note": "rocprofv3 PMC, separate passes; raw FETCH_SIZE + WRITE_SIZE of the step kernels per step launch"},
              open(os.path.join(dst, f"{tag}_{cfg}_traffic.json"), "w"), indent=1)
con = db("pmc_sq")
if con:
    lines += [f"## SQ counters of `mjh_step_kernel` (per launch, averages over the last {timed} launches)", "", "| counter | value | per env |", "|---|---|---|"]
    names = [r[0] for r in con.execute("select distinct counter_name from counters_collection")]
    for n in sorted(names):
        rows = con.execute("select value from counters_collection where kernel_name like '%mjh_step_kernel%' and counter_name=? order by start", (n,)).fetchall()
        last = [r[0] for r in rows[-timed:]]
        if last:
            v = sum(last) / len(last)
            lines.append(f"| {n} | {v:.4g} | {v / envs_per_launch:.4g} |")
    lines.append("")
if bench:
    json.dump(bench, open(os.path.join(dst, f"{tag}_{cfg}_bench.json"), "w"))
    lines += ["## bench.py line of the traced run", "", "```", json.dumps(bench), "```", ""]
open(os.path.join(dst, f"{tag}_{cfg}_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines[:60]))
