#!/bin/bash
# The weighted FIXED32 pass with and without the rare rows split out of the staged matrix (pair_low_kernel), same clock:
# the BASELINE configs and the sparse regime, the plan's own choice (auto) beside both forced ways.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for wl in C3 C4 C5 8192x50000@0.01 8192x50000@0.002 3000x10000 6000x10000 12000x10000; do
  for v in 0 1 auto; do
    if [ $v = auto ]; then unset FF_SPARSE_SPLIT; else export FF_SPARSE_SPLIT=$v; fi
    python3 "$R/bench.py" --workload $wl --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --no-live-traffic --no-end-to-end 2>/tmp/ss.err |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$wl split=$v: ms_per_step %.3f kernel_ms %.3f (%s) frac %.3f rare_rows %s of %d' % (d['ms_per_step'], r['kernel_ms'], '+'.join(r.get('kernels', [r['kernel']])), r['frac'], r.get('rare_rows', 0), r['rows_staged']))" || tail -3 /tmp/ss.err
  done
done
