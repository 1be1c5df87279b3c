#!/bin/bash
# Re-profiles the bench workloads at HEAD on the GPU box and leaves the summaries under
# gpurun_out/prof_<tag>/ (copy what is to be judged into profiles/):
#   <w>_kernel_stats.csv   rocprofv3 --kernel-trace --stats -- python3 bench.py <args of w>
#   <w>_bench.json         the bench line printed under that profiler run
#   <w>_pmc.txt            per-launch averages of FETCH_SIZE / WRITE_SIZE (separate --pmc passes) and,
#                          for the two dominant kernels, the SQ counters
# usage: tools/profile_round.sh <tag> [workload ...]      (default: all)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
out=$R/gpurun_out/prof_$tag
mkdir -p "$out"
declare -A ARGS KERN
E2E="--no-end-to-end"   # (the bench's own child processes -- the frcfrc runs, the live counter passes -- stay out of a profile)
ARGS[c3]="--no-cpu-baseline --no-secondary $E2E";                         KERN[c3]="pair_sad_kernel|pair_low_kernel"
ARGS[c3_unweighted]="--unweighted --no-cpu-baseline --no-secondary $E2E --steps 50"; KERN[c3_unweighted]=pair_common_mfma
ARGS[c3_unweighted_lognormal]="--unweighted --lengths lognormal --no-cpu-baseline --no-secondary $E2E --steps 50"; KERN[c3_unweighted_lognormal]=pair_common_mfma
ARGS[c3_unweighted_exact]="--unweighted --lengths lognormal --precision auto --no-cpu-baseline --no-secondary $E2E --steps 10"; KERN[c3_unweighted_exact]=pair_exact_unw
ARGS[c3_exact64]="--precision exact64 --steps 5 --no-cpu-baseline --no-secondary $E2E"; KERN[c3_exact64]=pair_exact64
ARGS[c2]="--workload C2 --unweighted --no-cpu-baseline --no-secondary $E2E --steps 50"; KERN[c2]=pair_common_small
ARGS[c4]="--workload C4 --steps 5 --no-cpu-baseline --no-secondary $E2E";  KERN[c4]="pair_sad_kernel|pair_low_kernel"
ARGS[c5]="--workload C5 --steps 5 --no-cpu-baseline --no-secondary $E2E";  KERN[c5]="pair_sad_kernel|pair_low_kernel"
# the sparse regime (bench.py SPARSE_REGIME): C5's tree and sample count at 1 % and 0.2 % leaf density
ARGS[c5s01]="--workload 8192x50000@0.01 --steps 5 --no-cpu-baseline --no-secondary $E2E";   KERN[c5s01]="pair_sad_kernel|pair_low_kernel"
ARGS[c5s002]="--workload 8192x50000@0.002 --steps 5 --no-cpu-baseline --no-secondary $E2E"; KERN[c5s002]="pair_sad_kernel|pair_low_kernel"
list=("$@"); [ ${#list[@]} -eq 0 ] && list=(c3 c3_unweighted c3_unweighted_lognormal c3_unweighted_exact c3_exact64 c2 c4 c5 c5s01 c5s002)
cd /tmp && export TMPDIR=/tmp
for w in "${list[@]}"; do
  echo "== $w: kernel trace"
  rm -rf "$out/$w.trace"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$w.trace" -- python3 "$R/bench.py" ${ARGS[$w]} > "$out/${w}_bench.json" 2> "$out/${w}_bench.err" || { echo "trace of $w failed"; tail -n 5 "$out/${w}_bench.err"; exit 1; }
  cp "$out/$w.trace"/*/*kernel_stats.csv "$out/${w}_kernel_stats.csv"
  if [ "$w" != c2 ]; then
    sets=("FETCH_SIZE" "WRITE_SIZE")
    if [ "$w" = c3 ] || [ "$w" = c3_unweighted ] || [ "$w" = c3_unweighted_lognormal ] || [ "$w" = c3_unweighted_exact ]; then
      sets+=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE")
      [ "$w" = c3_unweighted ] || [ "$w" = c3_unweighted_lognormal ] && sets+=("SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD TCC_HIT_sum TCC_MISS_sum")
      [ "$w" = c3 ] && sets+=("SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM TCC_HIT_sum TCC_MISS_sum")
      [ "$w" = c3_unweighted_exact ] && sets+=("SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES TCC_HIT_sum TCC_MISS_sum")
    fi
    steps=3; [ "$w" = c4 ] || [ "$w" = c5 ] || [ "$w" = c5s01 ] || [ "$w" = c5s002 ] || [ "$w" = c3_exact64 ] && steps=2
    a=$(echo "${ARGS[$w]}" | sed -E 's/--steps [0-9]+//')
    "$R/tools/pmc.sh" "${tag}_$w" "${KERN[$w]}" "${sets[@]}" -- $a --steps $steps --warmup 1 > "$out/${w}_pmc.txt" || { echo "pmc of $w failed"; exit 1; }
    cat "$out/${w}_pmc.txt"
  fi
  python3 - "$out/${w}_kernel_stats.csv" "$out/${w}_bench.json" <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:4]:
    print("   %-70s calls %5s avg %10.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print("   bench: ms_per_step %.4f kernel_ms %.4f frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
except Exception as e:
    print("   (no bench line: %s)" % e)
PY
done
