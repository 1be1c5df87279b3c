#!/bin/bash
# Per-launch averages of SQ / TCC counters for one kernel of a bench.py run, one rocprofv3 --pmc
# pass per counter set (counters only -- no tracing domains besides the kernel trace that names the
# dispatches).  Run on the GPU box:
#     tools/pmc.sh <out-tag> <kernel-name-substring> "<set1 counters>" "<set2 counters>" ... -- <bench.py args>
# Writes gpurun_out/pmc_<tag>/summary.txt (and the raw csv per pass next to it).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; kern=$2; shift 2
sets=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
out=$R/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "${sets[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pass_$i" -- python3 "$R/bench.py" "$@" > "$out/pass_$i.json" 2> "$out/pass_$i.err" || { echo "pass $i failed"; tail -n 5 "$out/pass_$i.err"; exit 1; }
done
KERN="$kern" OUT="$out" python3 - <<'PY' | tee "$out/summary.txt"
import csv, glob, os
kerns, out = os.environ["KERN"].split("|"), os.environ["OUT"]   # "a|b": the per-launch averages of a and of b, added
for f in sorted(glob.glob(out + "/pass_*/*/*counter_collection.csv")):
    acc = {}
    for r in csv.DictReader(open(f)):
        for kern in kerns:
            if kern in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], {}).setdefault(kern, []).append(float(r["Counter_Value"]))
    for k, per in acc.items():
        total = sum(sum(v) / len(v) for v in per.values())
        print("%-32s %16.1f   (%s)" % (k, total, " + ".join("avg of %d launches of *%s*" % (len(v), kern) for kern, v in per.items())))
PY
