#!/bin/bash
# C3 unweighted on the matrix cores: the two ways out of pair_common_mfma_kernel, same clock.
#   default               every item owns a private partial tile, reduce_private_kernel sums / finishes all 272 tiles
#   FF_MFMA_PRIVATE_MB=0  a main-round item (its tile's only one) finishes in place through LDS; only the 16 split
#                         tiles of the remainder go through partial tiles + reduce_partials_kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for v in "" "0"; do
  echo "== FF_MFMA_PRIVATE_MB=${v:-default}"
  FF_MFMA_PRIVATE_MB=$v python3 "$R/bench.py" --unweighted --steps 200 --warmup 10 --no-secondary --no-cpu-baseline --no-live-traffic --no-end-to-end 2>/dev/null |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
