for w in 208 192 176 160 144 128; do MJH_WINDOW64=$w tools/s24_quick.sh s24d_w64_$w --config s24d; done
MJH_WINDOW64=192 tools/s24_quick.sh s24d_w64_192_c3 --config s24d --cohorts 3
MJH_WINDOW64=160 tools/s24_quick.sh s24d_w64_160_c3 --config s24d --cohorts 3
