#!/bin/bash
# Per-kernel times of ONE run of the frcfrc command on a generated C4-sized table (rocprofv3 --kernel-trace --stats,
# the executable directly behind `--`): where the device side of the command spends its time -- stage A, staging,
# the pair kernel's passes, the formatter's three kernels.  usage: cli_kernel_trace.sh SAMPLES LEAVES DENSITY OUT
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
rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -- "$R/frackyfrac_amd/lib/frcfrc" -s -w -t $d/t.tree -i $d/t.tab -o $d/out.txt -stats 2> $d/err.txt
tail -n 3 $d/err.txt
cp $d/trace/*/*kernel_stats.csv "$out"
python3 - "$out" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("   %-60s calls %4s total %9.1f us avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
rm -rf $d
