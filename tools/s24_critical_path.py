"""S24: what bounds the window chain — the serial Gauss-Seidel chain of the SLOWEST environment ("commit the measured
critical-path length ... that is the bound").  A wavefront of the window kernel carries four envs and runs max(sweeps) x max(windows)
window-sweeps of ~130 instructions at one instruction per ~9 clocks (one wave per SIMD); a cohort's next step waits for its slowest
wave.   python tools/s24_critical_path.py [nenv] [settle] [config: s24 | s24d]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mujoco_sim_amd as ms
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 400
m = ms.scene("s24"); e = ms.Engine(m, nenv); e.load_s24(); e.set_cohorts(3)
e.step(settle); e.synchronize()
CLK_PER_WS = 1170.0      # clocks per window-sweep: 130 instructions x 9 clocks (profiles/r02m_valu_issue_bench.txt, r04a_s24_summary.md)
acc = []
for rep in range(20):
    e.step(1); e.synchronize()
    st = e.get_stats()
    nefc, it = st[:, 1], st[:, 2]
    nwin = (nefc + 15) // 16
    work = it * nwin                                    # window-sweeps of the env on its own
    G = 3
    worst_wave = 0; mean_wave = []
    for g in range(G):
        g0, g1 = nenv * g // G, nenv * (g + 1) // G
        order = g0 + np.argsort(-(work[g0:g1]), kind="stable")         # longest job first, as mjh_order_kernel dispatches
        for k in range(0, len(order), 4):
            idx = order[k:k + 4]
            w = it[idx].max() * nwin[idx].max()
            worst_wave = max(worst_wave, w); mean_wave.append(w)
    acc.append((work.max(), work.mean(), worst_wave, np.mean(mean_wave), nefc.max(), nefc.mean(), it.mean(), (it >= 100).mean(), nwin.max()))
a = np.array(acc).mean(0)
print(f"S24 {nenv} envs, 20 steps after {settle}: rows mean {a[5]:.1f} max {a[4]:.0f} (windows max {a[8]:.1f}); sweeps mean {a[6]:.1f}, at the 100-sweep cap {100 * a[7]:.1f} % of the envs")
print(f"window-sweeps per env-step: mean {a[1]:.0f}, slowest env {a[0]:.0f}; per wavefront (4 envs, longest job first): mean {a[3]:.0f}, slowest {a[2]:.0f}")
for f in (2.4e9, 2.1e9):
    print(f"at {CLK_PER_WS:.0f} clocks per window-sweep and {f / 1e9:.1f} GHz: slowest wave {a[2] * CLK_PER_WS / f * 1e6:.0f} us, mean wave {a[3] * CLK_PER_WS / f * 1e6:.0f} us "
          f"-> a cohort's step cannot be shorter than assemble (77 us) + {a[2] * CLK_PER_WS / f * 1e6:.0f} us, whatever the number of cohorts: {nenv / (77e-6 + a[2] * CLK_PER_WS / f) / 1e6:.1f} M env-steps/s")
print(f"SIMD fill of the window kernel if every wave lived as long as the slowest: {a[3] / a[2]:.2f}")
