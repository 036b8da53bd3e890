"""S24: would 32-row windows for the environments with many rows shorten the window kernel?  A model from measured per-env statistics:
16-row mode = 4 envs per wave, max(sweeps) x max(ceil(rows / 16)) window-sweeps of C16 clocks; 32-row mode = 2 envs per wave,
max(sweeps) x max(ceil(rows / 32)) of C32 clocks; waves list-scheduled longest first on 1024 SIMD slots (one wave per SIMD).
python tools/s24_hybrid_model.py [nenv] [settle]"""
import sys, os, heapq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mujoco_sim_amd as ms
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 400
m = ms.scene("s24"); e = ms.Engine(m, nenv); e.load_s24(); e.set_cohorts(3)
e.step(settle); e.synchronize()
C16, F = 1040.0, 2.4e9          # clocks per 16-row window-sweep as measured (345 us for 800 window-sweeps), shader clock
def makespan(durs, slots):
    h = [0.0] * slots; heapq.heapify(h)
    for d in sorted(durs, reverse=True):
        t = heapq.heappop(h); heapq.heappush(h, t + d)
    return max(h)
res = {}
for rep in range(10):
    e.step(3); e.synchronize()
    st = e.get_stats(); nefc, it = st[:, 1].astype(int), st[:, 2].astype(int)
    G = 3
    for C32ratio in (1.43, 1.6):           # 186 / 130 instructions; a pessimistic 208 / 130
        for T in (1 << 20, 112, 96, 80, 64):
            worst = 0.0
            for g in range(G):
                idx = np.arange(nenv * g // G, nenv * (g + 1) // G)
                heavy = idx[nefc[idx] > T]; light = idx[nefc[idx] <= T]
                durs = []
                hs = heavy[np.argsort(-(it[heavy] * ((nefc[heavy] + 31) // 32)), kind="stable")]
                for k in range(0, len(hs), 2):
                    w = hs[k:k + 2]; durs.append(it[w].max() * ((nefc[w].max() + 31) // 32) * C16 * C32ratio)
                ls = light[np.argsort(-(it[light] * ((nefc[light] + 15) // 16)), kind="stable")]
                for k in range(0, len(ls), 4):
                    w = ls[k:k + 4]; durs.append(it[w].max() * ((nefc[w].max() + 15) // 16) * C16)
                # the cohort's waves share the chip with the other cohorts': a third of the SIMD slots
                worst = max(worst, makespan(durs, 1024 // G + 1))
            res.setdefault((C32ratio, T), []).append((worst / F * 1e6, (nefc > T).mean()))
print(f"S24 {nenv} envs: window kernel duration per cohort launch (model; measured today 345 us)")
for (r, T), v in res.items():
    a = np.array(v).mean(0)
    print(f"  32-row cost {r:.2f} x 16-row, rows > {T if T < 1 << 19 else 'inf':>4}: {a[0]:6.0f} us, {100 * a[1]:5.1f} % of the envs in 32-row mode -> step ~{77 + 14 + a[0]:.0f} us -> {nenv / (77 + 14 + a[0]) :.2f} M env-steps/s")
