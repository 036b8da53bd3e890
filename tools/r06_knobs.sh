#!/bin/bash
# round 6: the cheap knobs once more on this round's boxes (each line one bench process; same call)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for q in 4 8 16; do GPU_MAX_HW_QUEUES=$q tools/s24_quick.sh s24_q$q; GPU_MAX_HW_QUEUES=$q tools/s24_quick.sh s24d_q$q --config s24d --steps 200; done
for oe in 8 16 32 64; do MJH_ORDER_EVERY=$oe tools/s24_quick.sh s24_oe$oe; MJH_ORDER_EVERY=$oe tools/s24_quick.sh s24d_oe$oe --config s24d --steps 200; done
for c in 2 3 4; do tools/s24_quick.sh s24_c$c --cohorts $c; tools/s24_quick.sh s24d_c$c --config s24d --steps 200 --cohorts $c; done
for t in 88 96 104; do MJH_WINDOW64=$t tools/s24_quick.sh s24_w64_$t; done
for t in 192 208 224; do MJH_WINDOW64=$t tools/s24_quick.sh s24d_w64_$t --config s24d --steps 200; done
for mc in 80 88 96; do tools/s24_quick.sh s24d_cap$mc --config s24d --steps 200 --maxcon $mc; done
for ts in 1 5 20; do tools/s24_quick.sh s24_ts$ts --timing-stride $ts; done
