"""Launch timeline of one step kernel (debug tool): when each env started/ended, concurrency, tail.
python tools/timeline.py [nenv] [settle]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
import mujoco_sim_amd as ms
from mujoco_sim_amd import capi
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 400
m = ms.scene("s24"); e = ms.Engine(m, nenv); e.load_s24()
e.step(settle); e.synchronize()
L = capi.load()
L.mjh_debug_stage_raw.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
raw = np.zeros((nenv, 20), dtype=np.int64)
for rep in range(2):
    rc = L.mjh_debug_stage_raw(e.h, 0, raw.ctypes.data)
    assert rc == 0
hw = raw[:, 18] & 0xffffffff; xcc = (raw[:, 18] >> 32) & 0xf
t0 = raw[:, 16].min()
start = (raw[:, 16] - t0) * 0.01; end = (raw[:, 17] - t0) * 0.01   # us (100 MHz wall clock)
dur = end - start
print("shader ticks per us: %.0f" % np.median((raw[:, 15] - raw[:, 0]) / np.maximum(dur, 1e-3)))
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1; simd = (hw >> 4) & 3
print("makespan %.0f us; sum(dur) %.3e; sum/makespan = mean concurrency %.0f" % (end.max(), dur.sum(), dur.sum() / end.max()))
print("dur mean %.0f  p10 %.0f p50 %.0f p90 %.0f max %.0f" % (dur.mean(), *np.percentile(dur, [10, 50, 90]), dur.max()))
print("start: first-wave envs (start < 5%% of makespan): %d ; last start at %.2f of makespan" % ((start < 0.05 * end.max()).sum(), start.max() / end.max()))
T = end.max()
for f in np.linspace(0.05, 0.95, 10):
    t = f * T
    print("  t=%.2f T: %4d envs active" % (f, ((start <= t) & (end > t)).sum()))
# per-XCD finish time
for x in range(8):
    s = xcc == x
    print("  xcc %d: %4d envs, busy sum %.3e, last end %.2f T, distinct (se,sh,cu) %d" % (x, s.sum(), dur[s].sum(), end[s].max() / T, len(set(zip(se[s], sh[s], cu[s])))))
x0 = xcc == 0
cus = sorted(set(zip(se[x0], sh[x0], cu[x0])))
print("xcc0 per-CU: envs, busy/T, last end/T")
for c in cus:
    s = x0 & (se == c[0]) & (sh == c[1]) & (cu == c[2])
    print("   ", c, s.sum(), "%.2f %.2f" % (dur[s].sum() / T, end[s].max() / T), "simd counts", np.bincount(simd[s], minlength=4))
# duration of the same env alone vs. crowded: correlate dur with concurrency at its midpoint
mid = 0.5 * (start + end)
conc = np.array([((start <= t) & (end > t)).sum() for t in mid[::16]])
st = e.get_stats()
print("corr(dur, concurrency at midpoint) = %.2f" % np.corrcoef(dur[::16], conc)[0, 1])
print("dur by launch position (mean over 256-blocks):", np.round(dur.reshape(-1, 256).mean(1)).astype(int))
print("start by launch position (mean over 256-blocks):", np.round(start.reshape(-1, 256).mean(1)).astype(int))
last = np.argsort(-end)[:12]
for b in last:
    print("  block %4d xcc %d start %.0f dur %.0f end %.0f" % (b, xcc[b], start[b], dur[b], end[b]))
# stage split of a long env vs the median
k = np.argmax(dur)
print("longest env: block", k, "stamps (us):", np.round((raw[k, 1:16] - raw[k, 0]) / 2224.0).astype(int))
# predictor quality: cost model and previous-step duration vs this step's duration / PGS ticks
envid = raw[:, 19].copy(); pgs_prev = np.zeros(nenv); pgs_prev[envid] = (raw[:, 13] - raw[:, 12]); dur_prev = np.zeros(nenv); dur_prev[envid] = raw[:, 15] - raw[:, 0]
st_prev = e.get_stats().astype(float)
raw2 = np.zeros((nenv, 20), dtype=np.int64)
assert L.mjh_debug_stage_raw(e.h, 0, raw2.ctypes.data) == 0
env2 = raw2[:, 19]; pgs_now = np.zeros(nenv); pgs_now[env2] = raw2[:, 13] - raw2[:, 12]
st_now = e.get_stats().astype(float)
model_prev = st_prev[:, 2] * (st_prev[:, 1] + 24)
print("corr(prev cost model, PGS ticks now) %.3f ; corr(prev PGS ticks, PGS ticks now) %.3f" % (np.corrcoef(model_prev, pgs_now)[0, 1], np.corrcoef(pgs_prev, pgs_now)[0, 1]))
print("corr(niter prev, niter now) %.3f; corr(nefc prev, nefc now) %.3f" % (np.corrcoef(st_prev[:, 2], st_now[:, 2])[0, 1], np.corrcoef(st_prev[:, 1], st_now[:, 1])[0, 1]))
model_now = st_now[:, 2] * (st_now[:, 1] + 24)
print("corr(cost model now, PGS ticks now) %.3f  (model fidelity)" % np.corrcoef(model_now, pgs_now)[0, 1])
per_sweep = pgs_now / np.maximum(st_now[:, 2], 1)
A = np.stack([st_now[:, 1], st_now[:, 0], np.ones(nenv)], 1)
coef = np.linalg.lstsq(A, per_sweep, rcond=None)[0]
print("ticks per sweep ~ %.1f*nefc + %.1f*ncon + %.0f ; residual rel %.3f" % (*coef, np.std(per_sweep - A @ coef) / per_sweep.mean()))
pos2 = np.zeros(nenv, dtype=int); pos2[env2] = np.arange(nenv)
fw = pos2 < 2048 - 256   # first-wave envs of the 2nd launch
for nm, s in (("first wave", fw), ("second wave", ~fw)):
    x = model_now[s]; y = pgs_now[s]
    print(nm, "corr(model now, PGS ticks) %.3f; corr(niter, ticks) %.3f; corr(nefc, ticks) %.3f  mean ticks %.0f" % (np.corrcoef(x, y)[0, 1], np.corrcoef(st_now[s, 2], y)[0, 1], np.corrcoef(st_now[s, 1], y)[0, 1], y.mean()))
k = np.argmax(pgs_now)
print("slowest PGS env: stats", st_now[k], "ticks", pgs_now[k], "pos", pos2[k])
for it in (100,):
    s = fw & (st_now[:, 2] == it)
    print("first-wave envs with niter==100: n", s.sum(), "corr(nefc,ticks) %.3f" % np.corrcoef(st_now[s, 1], pgs_now[s])[0, 1], "ticks/sweep/row mean %.2f" % (pgs_now[s] / 100 / st_now[s, 1]).mean())
    print("   ticks percentiles", np.percentile(pgs_now[s], [5, 50, 95]).astype(int), " nefc percentiles", np.percentile(st_now[s, 1], [5, 50, 95]))
