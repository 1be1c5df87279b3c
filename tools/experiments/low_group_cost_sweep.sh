#!/bin/bash
# The split decision's price of 64 updates of a heavy rare row (FF_LOW_GROUP_COST, ps chip-wide; an experiment hook read when
# the plan is staged) swept: the kernel time the decision leads to, per workload.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for wl in ${WORKLOADS:-C3 C4 C5 8192x50000@0.01 8192x50000@0.002 3000x10000 6000x10000}; do
  for v in ${SWEEP:-15 25 35 50 70 100}; do
    export FF_LOW_GROUP_COST=$v
    python3 "$R/bench.py" --workload $wl --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --no-live-traffic --no-end-to-end 2>/tmp/ss.err |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$wl group_cost=$v ps: ms_per_step %.3f kernel_ms %.3f (%s) rare_rows %s of %d' % (d['ms_per_step'], r['kernel_ms'], '+'.join(r.get('kernels', [r['kernel']])), r.get('rare_rows', 0), r['rows_staged']))" || tail -3 /tmp/ss.err
  done
done
# (FF_LOW_GROUP_COST was a staging-time hook for this sweep only -- ff_dev_stage.hip's GROUP_COST; not in the tree.)
