"""Prints kernel statistics from a rocprofv3 rocpd .db (first found under the directory): top kernels, and the mean duration of every
position of the repeating launch sequence over the last N kernels (many-body chain: assemble / solve / integrate are the same kernel
name at different positions).  usage: python tools/kstats.py <dir> [last_n] [period]"""
import glob, os, sqlite3, sys
d = sys.argv[1]; last_n = int(sys.argv[2]) if len(sys.argv) > 2 else 300; period = int(sys.argv[3]) if len(sys.argv) > 3 else 0
f = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
con = sqlite3.connect(f[0])
print("name | calls | total us | avg us | %")
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{r[0][:90]} | {r[1]} | {r[2]/1e3:.1f} | {r[3]/1e3:.2f} | {r[4]:.2f}")
rows = con.execute("select k.name, k.start, k.end, k.stream_id from kernels k order by k.start").fetchall()
rows = [r for r in rows if "mjh_" in r[0] and "order_kernel" not in r[0] and "export" not in r[0]][-last_n:]
if period:
    # group by stream, then by position in the sequence
    by = {}
    for n, s, e, st in rows:
        by.setdefault(st, []).append((n.split("(")[0], (e - s) / 1e3, s / 1e3, e / 1e3))
    for st, lst in by.items():
        k = len(lst) // period * period; lst = lst[-k:]
        print(f"stream {st}: {k // period} sequences")
        for p in range(period):
            v = [lst[i][1] for i in range(p, k, period)]
            print(f"   pos {p}: {lst[p][0][:70]}  mean {sum(v)/len(v):.1f} us  max {max(v):.1f}")
        gaps = [lst[i + 1][2] - lst[i][3] for i in range(k - 1)]
        print(f"   mean gap between consecutive kernels {sum(gaps)/len(gaps):.1f} us; sequence period {(lst[-1][3] - lst[0][2]) / (k // period):.1f} us")
