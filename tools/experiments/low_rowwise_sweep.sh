#!/bin/bash
# pair_low_kernel: the updates-per-row threshold from which a bitmap word's rows are taken by groups of lanes
# (FF_LOW_ROWWISE, an experiment hook read at launch) swept over the BASELINE configs and the sparse regime; split forced
# so that the same rows are rare at every setting.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for wl in ${WORKLOADS:-C3 C4 C5 8192x50000@0.01 8192x50000@0.002}; do
  for v in ${SWEEP:-0 2 8 32 1000000}; do
    export FF_LOW_ROWWISE=$v
    python3 "$R/bench.py" --workload $wl --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --no-live-traffic --no-end-to-end 2>/tmp/ss.err |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$wl rowwise_min=$v: ms_per_step %.3f kernel_ms %.3f (%s) rare_rows %s of %d' % (d['ms_per_step'], r['kernel_ms'], '+'.join(r.get('kernels', [r['kernel']])), r.get('rare_rows', 0), r['rows_staged']))" || tail -3 /tmp/ss.err
  done
done
# (FF_LOW_ROWWISE was a launch-time hook for this sweep only: the updates-per-row threshold between the search over a
# word's updates and the walk of its B entries.  The last sweep -- profiles/r05_low_rowwise_sweep.txt: 0 = always the
# walk, 1000000 = always the search -- had the walk ahead at every density, and the search left the kernel with the hook.)
