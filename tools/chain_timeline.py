"""Timeline of the many-body launch chain from a rocprofv3 kernel trace (rocpd .db): per phase mean duration (the assemble and the
integrate launch are the same kernel: told apart by what follows on the same queue), and the last few steps as a table.
usage: python tools/chain_timeline.py <raw trace dir> [steps shown]"""
import glob, os, sqlite3, sys
raw = sys.argv[1]; show = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = glob.glob(os.path.join(raw, "**", "*.db"), recursive=True)
con = sqlite3.connect(f[0])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = con.execute(f"select name, start, end, {qcol or '0'} from kernels where name like '%mjh_%' order by start").fetchall()
short = lambda n: n.split("(")[0].replace("void ", "").replace("mjh_", "")
byq = {}
for n, s, e, q in rows:
    byq.setdefault(q, []).append((short(n), s, e))
phase = {}
events = []
for q, ks in byq.items():
    for i, (n, s, e) in enumerate(ks):
        if n.startswith("step_kernel"):
            nxt = ks[i + 1][0] if i + 1 < len(ks) else ""
            n = "assemble" if ("dense" in nxt or "solve" in nxt) else "integrate"
        phase.setdefault(n, []).append((e - s) / 1e3)
        events.append((s, e, q, n))
print("columns:", qcol, " queues:", len(byq))
for n, d in sorted(phase.items()):
    tail = d[-300:]
    print(f"{n:28s} launches {len(d):5d}  mean of last {len(tail)}: {sum(tail) / len(tail):8.1f} us   max {max(tail):8.1f}")
events.sort()
last = events[-show * 5 * max(len(byq), 1):]
t0 = last[0][0]
for s, e, q, n in last:
    print(f"  q{q}  {n:22s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us")
