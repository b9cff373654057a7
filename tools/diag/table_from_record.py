"""Prints the robots.py keyword arguments of the table a tools/identify_r6.py `minimal` record ends with (entries that still differ from the
reference point, rounded to 5 significant digits), so that the shipped table is copied from the record and not typed.
usage: python tools/diag/table_from_record.py profiles/r06_laikago_minimal.json"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import identify_r6 as ir

rec = json.load(open(sys.argv[1]))
spec = ir.SPECS[rec["robot"]]
th = rec["minimal"]["theta"]
base = ir.reference_theta(spec)
moved = rec["minimal"]["still_moved"]
out = {}
for k in moved:
    out[k] = float("%.5g" % th[k]) if k in spec["params"] else int(th[k])
if "soft" in moved and th["soft"]:
    out["soft_k"], out["soft_d"] = float("%.5g" % th["soft_k"]), float("%.5g" % th["soft_d"])
print(json.dumps(out, indent=1))
print("# effect of putting each back:", json.dumps({k: round(v["min_J_if_put_back"] - rec["minimal"]["min_J"], 4) for k, v in rec["minimal"]["effect_of_each_moved_entry"].items()}))
