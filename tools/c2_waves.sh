for w in 1 2 4; do MJH_SOLVE_WAVES=$w timeout 400 python bench.py --config c2 --steps 60 --no-cpu-baseline --no-second-window 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves $w:', round(d['value']), round(d['ms_per_step'],2), d['config']['mean_ncon'])"; done
