#!/bin/bash
# Counters of pair_low_kernel alone at C3 (what its batches of 64 updates wait for): tools/pmc.sh passes.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
"$R/tools/pmc.sh" r05_low pair_low_kernel "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" -- --no-cpu-baseline --no-secondary --no-end-to-end --steps 3 --warmup 1
