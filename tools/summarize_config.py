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


def mangled_key(name):
    """'void mjh_step_kernel<2, true, false, true>' -> 'mjh_step_kernelILi2ELb1ELb0ELb1EE' (the piece of the mangled name that identifies the instance)"""
    import re
    m = re.search(r"(mjh_\w+)(?:<([^>]*)>)?", name)
    if not m:
        return name
    base, args = m.group(1), m.group(2)
    if not args:
        return base
    enc = "".join(("Lb1E" if a.strip() == "true" else "Lb0E" if a.strip() == "false" else f"Li{a.strip()}E") for a in args.split(","))
    return f"{base}I{enc}E"


def code_object_vgprs():
    """{instance key: .vgpr_count} from the gfx950 code object embedded in libmjhip.so (llvm-readelf --notes)"""
    import re, subprocess, tempfile
    out = {}
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mujoco_sim_amd", "libmjhip.so")
    try:
        data = open(lib, "rb").read()
        idx = 0
        while True:
            i = data.find(b"\x7fELF", idx)
            if i < 0:
                break
            if data[i + 18:i + 20] == (224).to_bytes(2, "little"):
                shoff = int.from_bytes(data[i + 40:i + 48], "little"); shentsize = int.from_bytes(data[i + 58:i + 60], "little"); shnum = int.from_bytes(data[i + 60:i + 62], "little")
                with tempfile.NamedTemporaryFile(suffix=".elf") as f:
                    f.write(data[i:i + shoff + shentsize * shnum]); f.flush()
                    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
                for b in re.split(r"\n\s+- \.agpr_count:", "\n" + txt)[1:]:
                    nm = re.search(r"\.name:\s+(\S+)", b); vg = re.search(r"\.vgpr_count:\s+(\d+)", b)
                    if nm and vg:
                        km = re.match(r"_Z(\d+)", nm.group(1))
                        if km:
                            st = len(km.group(0)); base = nm.group(1)[st:st + int(km.group(1))]
                            tm = re.match(r"I(?:L[ib]\d+E)+E", nm.group(1)[st + int(km.group(1)):])
                            out[base + (tm.group(0) if tm else "")] = int(vg.group(1))
            idx = i + 4
    except Exception:
        pass
    return out


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
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    acc = "k.accum_vgpr_count" if "accum_vgpr_count" in cols else "0"
    rows = con.execute(f"select k.name, k.start, k.end, k.vgpr_count, k.sgpr_count, k.lds_size, k.scratch_size, {acc} from kernels k where k.name like '%mjh_step_kernel%' or k.name like '%mjh_solve_kernel%' or k.name like '%mjh_dense_%' or k.name like '%mjh_solve_mixed%' or k.name like '%mjh_window_kernel%' order by k.start").fetchall()
    if rows:
        per = 3 if any("mjh_solve_kernel" in r[0] for r in rows) else (2 if any("mjh_window_kernel" in r[0] for r in rows) else 1)
        if any("mjh_dense_" in r[0] for r in rows):
            per = 5        # assemble -> dense build -> dense solve -> block solve (cohorts without a long-sweeping env, envs beyond the capacity) -> integrate
        last = rows[-timed * per:]
        tot = sum(e - s for _, s, e, *_ in last) / 1e3
        per_step_us = tot / max(timed, 1)
        lines += ["", f"Step kernels per launch: {per}.  Summed kernel time per step launch over the LAST {timed} launches (the timed region): **{per_step_us:.1f} us**"]
        seen = {}
        for n, s, e, vg, sg, lds, scr, av in last:
            k = n.split("(")[0]
            a = seen.setdefault(k, [0, 0.0, vg, sg, lds, scr, av]); a[0] += 1; a[1] += (e - s) / 1e3
        notes = code_object_vgprs()
        for k, a in seen.items():
            # (rocprofv3's vgpr_count column is not the allocation that sets the occupancy on gfx950's unified register file — it reads
            #  about half of it; the code object's .vgpr_count note, what tools/kernel_resources.sh prints, is)
            co = notes.get(mangled_key(k))
            reg = f"{co} unified registers (code object .vgpr_count; rocprofv3's column: {a[2]})" if co else f"rocprofv3 vgpr_count {a[2]} (+ {a[6]} accum)"
            lines.append(f"- `{k}`: {a[0]} launches, mean {a[1] / a[0]:.1f} us; {reg}, SGPR {a[3]}, LDS {a[4]} B/workgroup, scratch {a[5]} B/lane")
        lines.append("")
tot = {}
for sub, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    con = db(sub)
    if not con:
        continue
    acc = 0.0; cnt = 0
    rows = con.execute("select kernel_name, value from counters_collection where (kernel_name like '%mjh_step_kernel%' or kernel_name like '%mjh_solve_kernel%' or kernel_name like '%mjh_dense_%' or kernel_name like '%mjh_solve_mixed%' or kernel_name like '%mjh_window_kernel%') and counter_name=? order by start", (name,)).fetchall()
    per = 3 if any("mjh_solve_kernel" in r[0] for r in rows) else (2 if any("mjh_window_kernel" in r[0] for r in rows) else 1)
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
sq = {}
for sub in ("pmc_sq", "pmc_sq2"):
    con = db(sub)
    if not con:
        continue
    for kn, cn, v in con.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like '%mjh_%' and kernel_name not like '%order_kernel%' and kernel_name not like '%export%' order by start"):
        sq.setdefault(kn.split("(")[0], {}).setdefault(cn, []).append(v)
sqsum = {}
if sq:
    lines += [f"## SQ counters per kernel (`--pmc`, own passes; averages over each kernel's last {timed} launches; one launch = {envs_per_launch:.0f} envs)", ""]
    for kn, cs in sq.items():
        m = {c: sum(v[-timed:]) / len(v[-timed:]) for c, v in cs.items()}
        sqsum[kn] = m
        lines += [f"`{kn}`", "", "| counter | per launch | per env |", "|---|---|---|"]
        for c in sorted(m):
            lines.append(f"| {c} | {m[c]:.4g} | {m[c] / envs_per_launch:.4g} |")
        d = []
        if m.get("SQ_INSTS_VALU") and m.get("SQ_THREAD_CYCLES_VALU") and m.get("SQ_ACTIVE_INST_VALU"):
            # SQ_ACTIVE_INST_VALU: cycles (x4 quad-cycles) the VALU is busy with wave instructions; SQ_THREAD_CYCLES_VALU: the same per
            # active lane — their ratio / 64 is the fraction of lanes live in the average VALU instruction
            d.append(f"active lanes per VALU instruction = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64 = **{m['SQ_THREAD_CYCLES_VALU'] / m['SQ_ACTIVE_INST_VALU'] / 64:.3f}**")
        if m.get("SQ_BUSY_CYCLES") and m.get("SQ_ACTIVE_INST_VALU"):
            d.append(f"SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = {m['SQ_ACTIVE_INST_VALU'] / m['SQ_BUSY_CYCLES']:.3f}")
        if m.get("SQ_WAVE_CYCLES") and m.get("SQ_WAVES"):
            d.append(f"wave cycles per wave = {m['SQ_WAVE_CYCLES'] / m['SQ_WAVES']:.4g}")
        if d:
            lines += ["", "; ".join(d)]
        lines.append("")
    json.dump(sqsum, open(os.path.join(dst, f"{tag}_{cfg}_sq.json"), "w"), indent=1)
    # whole step launch: VALU wave-instructions per env-step and the fraction of lanes live in them, into the traffic record (bench.py
    # turns the first into an issue fraction with the rate IT measures: x env-steps/s / (1024 SIMDs x 2.4 GHz / 4 clocks))
    # (only the kernels that run in EVERY step launch of the timed region: an instance met a few times while the scene settles — the window kernel's
    #  tier variant of S24's first steps — is listed above but is not part of a step)
    every = {kn for kn, cs in sq.items() if max(len(v) for v in cs.values()) >= timed}
    vi = sum(m.get("SQ_INSTS_VALU", 0.0) for kn, m in sqsum.items() if kn in every); va = sum(m.get("SQ_ACTIVE_INST_VALU", 0.0) for kn, m in sqsum.items() if kn in every)
    vt = sum(m.get("SQ_THREAD_CYCLES_VALU", 0.0) for kn, m in sqsum.items() if kn in every)
    tp = os.path.join(dst, f"{tag}_{cfg}_traffic.json")
    if vi > 0 and os.path.exists(tp):
        tj = json.load(open(tp))
        steps_per_launch = (bench or {}).get("config", {}).get("steps_per_launch", 1) or 1
        tj["valu_instr_per_env_step"] = vi / envs_per_launch / steps_per_launch
        tj["valu_lane_util"] = (vt / va / 64.0) if va > 0 and vt > 0 else None
        tj["valu_note"] = "rocprofv3 --pmc SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_THREAD_CYCLES_VALU, own passes, summed over the kernels of one step launch"
        json.dump(tj, open(tp, "w"), indent=1)
        lines += [f"Whole step launch: **{tj['valu_instr_per_env_step']:.0f} VALU wave-instructions per env-step**, active lanes per VALU instruction **{(tj['valu_lane_util'] or 0):.3f}**", ""]
if bench:
    json.dump(bench, open(os.path.join(dst, f"{tag}_{cfg}_bench.json"), "w"))
    lines += ["## bench.py line of the traced run", "", "```", json.dumps(bench), "```", ""]
open(os.path.join(dst, f"{tag}_{cfg}_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines[:60]))
