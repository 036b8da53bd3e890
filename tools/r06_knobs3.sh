cd ${GRAFT_REPO_ROOT:-/root/repo}
q() { L=$1; shift; bash tools/s24_quick.sh "$L" --steps 100 --warmup 20 "$@"; }
q default
for c in 2 4; do q "cohorts $c" --cohorts $c; done
for t in 80 88 104; do MJH_WINDOW64=$t q "win64 $t"; done
for oe in 16 64; do MJH_ORDER_EVERY=$oe q "order-every $oe"; done
q default2
qd() { L=$1; shift; bash tools/s24_quick.sh "$L" --config s24d --steps 200 --warmup 20 "$@"; }
qd "s24d default"
for c in 2 4; do qd "s24d cohorts $c" --cohorts $c; done
for t in 176 184 200 208; do MJH_WINDOW64=$t qd "s24d win64 $t"; done
qd "s24d default2"
