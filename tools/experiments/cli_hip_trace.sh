#!/bin/bash
# HIP API time of ONE run of the frcfrc executable on a generated table (rocprofv3 --hip-trace --stats, no counters):
# which runtime calls the command's "convert" phase (stage A + staging) and passes spend their host time in.
# usage: cli_hip_trace.sh SAMPLES LEAVES DENSITY OUT
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
ns=$1; nl=$2; dens=$3; out=$4
case "$out" in /*) ;; *) out="$PWD/$out";; esac
d=$(mktemp -d /tmp/ffcli.XXXXXX)
python3 - "$R" $ns $nl $dens $d <<'PY'
import sys
sys.path.insert(0, sys.argv[1])
from frackyfrac_amd import synth
ns, nl, dens, d = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), sys.argv[5]
tree, ptr, idx, val = synth.make(ns, nl, dens, 123)
open(d + "/t.tree", "w").write(tree.newick())
open(d + "/t.tab", "w").write(synth.sparse_text(tree, ptr, idx, val))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --stats --output-format csv -d $d/trace -- "$R/frackyfrac_amd/lib/frcfrc" -s -w -t $d/t.tree -i $d/t.tab -o $d/out.txt -stats 2> $d/err.txt
grep seconds $d/err.txt | tail -n 1
ls $d/trace/*/ 
cp $d/trace/*/*hip_api_stats.csv "$out" 2>/dev/null || cp $d/trace/*/*hip_stats.csv "$out"
python3 - "$out" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("   %-40s calls %5s total %9.1f ms avg %9.1f us" % (r["Name"][:40], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
rm -rf $d
