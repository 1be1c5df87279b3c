#!/bin/bash
# pair_low_kernel with every side of its blocks of pairs forced (FF_LOW_TILE), the split forced (FF_SPARSE_SPLIT=1: the same
# rare rows whatever the side): what the split estimate of ff_dev_stage.hip has to reproduce.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export FF_SPARSE_SPLIT=1
for wl in ${WORKLOADS:-C3 C4 C5 8192x50000@0.01 8192x50000@0.002 6000x10000 11584x10000}; do
  for v in 128 112 96 80 64; do
    export FF_LOW_TILE=$v
    python3 "$R/bench.py" --workload $wl --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --no-live-traffic --no-end-to-end 2>/tmp/ss.err |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$wl side=$v: kernel_ms %.3f rare_rows %s of %d' % (r['kernel_ms'], r.get('rare_rows', 0), r['rows_staged']))" || tail -3 /tmp/ss.err
  done
done
