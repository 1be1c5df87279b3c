"""profiles/traffic.json from the PMC summaries of a profiling round (tools/profile_round.sh <tag>, copied to
profiles/<tag>_<workload>_pmc.txt): bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB.
Usage: update_traffic.py <tag> <commit-note>"""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, note = sys.argv[1], sys.argv[2]
KEYS = {"c3": "C3_n1_k0", "c3_unweighted": "C3_n1_k2_unweighted", "c3_unweighted_exact": "C3_n1_k5_unweighted",
        "c3_exact64": "C3_n1_k7", "c4": "C4_n1_k0", "c5": "C5_n1_k0",
        "c5s01": "8192x50000@0.01_n1_k0", "c5s002": "8192x50000@0.002_n1_k0"}
path = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(path))
for w, key in KEYS.items():
    f = os.path.join(ROOT, "profiles", "%s_%s_pmc.txt" % (tag, w))
    if not os.path.exists(f):
        continue
    vals = {}
    for ln in open(f):
        m = re.match(r"(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)", ln)
        if m:
            vals[m.group(1)] = float(m.group(2))
    if len(vals) == 2:
        t[key] = {"bytes": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, "fetch_size_kib": vals["FETCH_SIZE"],
                  "write_size_kib": vals["WRITE_SIZE"], "source": "profiles/%s_%s_pmc.txt" % (tag, w), "commit": note}
        print(key, "%.3f GB" % (t[key]["bytes"] / 1e9))
json.dump(t, open(path, "w"), indent=1)
