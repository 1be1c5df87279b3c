#!/bin/bash
# Kernel times of the split path (FF_SPARSE_SPLIT=1) under rocprofv3 for one workload: pair_low_kernel next to the
# dense kernel over the remaining rows.  usage: sparse_split_trace.sh WORKLOAD OUT
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
wl=$1; out=$2
case "$out" in /*) ;; *) out="$PWD/$out";; esac
d=$(mktemp -d /tmp/ffsp.XXXXXX)
cd /tmp && export TMPDIR=/tmp
FF_SPARSE_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 "$R/bench.py" --workload $wl --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-live-traffic --no-end-to-end > $d/b.json 2> $d/b.err
cp $d/*/*kernel_stats.csv "$out"
python3 - "$out" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("   %-60s calls %4s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $d
