#!/bin/bash
# What a run of the frcfrc executable costs before and after its work: the reference's three-sample golden case, ten
# runs, wall clock per run (process start, HIP context, exit).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
G=$R/tests/golden
t0=$(date +%s.%N)
for k in 1 2 3 4 5 6 7 8 9 10; do "$R/frackyfrac_amd/lib/frcfrc" -w -t $G/wtd.tree -i $G/wtd.dense -o /tmp/ff_fixed_cost.out 2>/tmp/ff_fixed_cost.err; done
t1=$(date +%s.%N)
echo "golden wtd, 10 runs: $(python3 -c "print('%.3f s per run' % (($t1 - $t0) / 10))"); last run said: $(grep Took /tmp/ff_fixed_cost.err)"
