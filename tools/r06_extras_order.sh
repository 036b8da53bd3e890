cd ${GRAFT_REPO_ROOT:-/root/repo}
show() { python -c "
import json,sys
r=json.loads(open('$1').read().strip().splitlines()[-1])
print('$2','S24',round(r['value']/1e6,3),'inv',round(r['value_with_inverse']/1e6,3),'literal',round(r['literal_loop']['value']/1e6,3),'|',' '.join(f\"{k} {v['value']/1e6:.3f}\" for k,v in r['configs'].items()))"; }
for rep in 1 2; do
python bench.py --gpus 1 --steps 20 --warmup 5 > /tmp/a.json 2>/dev/null; show /tmp/a.json first
BENCH_EXTRAS_LAST=1 python bench.py --gpus 1 --steps 20 --warmup 5 > /tmp/b.json 2>/dev/null; show /tmp/b.json last
done
