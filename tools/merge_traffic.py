"""profiles/hbm_traffic.json (what bench.py's roofline leg cites) from the per-config records of tools/profile_config.sh:
python tools/merge_traffic.py profiles/r04e_c2_traffic.json profiles/r04a_s24_traffic.json ...   (later files win per config)"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles", "hbm_traffic.json")
cur = json.load(open(dst)) if os.path.exists(dst) else {}
for p in sys.argv[1:]:
    t = json.load(open(p))
    cur[t["config"]] = t
json.dump(cur, open(dst, "w"), indent=1)
print({k: v.get("tag") for k, v in cur.items()})
