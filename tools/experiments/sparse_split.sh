#!/bin/bash
# The sparse regime with and without the rare rows split out of the staged matrix (pair_low_kernel), same clock:
# C5's tree and sample count at 1 % and 0.2 % leaf density, C5 itself (5 %), C3.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for wl in 8192x50000@0.01 8192x50000@0.002 C5 C3; do
  for v in 0 1 auto; do
    if [ $v = auto ]; then unset FF_SPARSE_SPLIT; else export FF_SPARSE_SPLIT=$v; fi
    python3 "$R/bench.py" --workload $wl --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --no-live-traffic --no-end-to-end 2>/tmp/ss.err |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$wl split=$v: ms_per_step %.3f kernel_ms %.3f (%s) frac %.3f rows_staged %d' % (d['ms_per_step'], r['kernel_ms'], r['kernel'], r['frac'], r['rows_staged']))" || tail -3 /tmp/ss.err
  done
done
