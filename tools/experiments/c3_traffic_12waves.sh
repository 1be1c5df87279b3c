#!/bin/bash
# Round 4, VERDICT item 7 (one attempt): C3 weighted on the 12-wave kernel with its first round XCD-sliced
# (FF_XCD_SLICES = 2 / 4 / 8) against the plain thirds the plan takes by itself: kernel time (bench.py, HIP events)
# and fabric traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes through tools/pmc.sh).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for s in unset 2 4 8; do
  if [ "$s" = unset ]; then unset FF_XCD_SLICES; else export FF_XCD_SLICES=$s; fi
  export FF_WAVES_PER_WG=12
  ms=$(python3 "$R/bench.py" --steps 100 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; o=json.loads(sys.stdin.read()); print('%.4f %.4f' % (o['ms_per_step'], o['roofline']['kernel_ms']))")
  "$R/tools/pmc.sh" "x7_$s" pair_sad_kernel "FETCH_SIZE" "WRITE_SIZE" -- --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > /tmp/x7_$s.txt 2>&1
  f=$(grep FETCH_SIZE /tmp/x7_$s.txt | awk '{print $2}'); w=$(grep WRITE_SIZE /tmp/x7_$s.txt | awk '{print $2}')
  python3 -c "print('FF_XCD_SLICES=%s: ms_per_step kernel_ms = %s | FETCH_SIZE %s KiB WRITE_SIZE %s KiB -> %.2f GB per launch' % ('$s', '$ms', '$f', '$w', (2*float('$f')+float('$w'))*1024/1e9))"
done
