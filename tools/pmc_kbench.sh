#!/bin/bash
# SQ / TCC counter passes of a tools/kbench.py command (one rocprofv3 run per counter set, --kernel-trace only).
# usage: tools/pmc_kbench.sh <outdir> <kbench args...>
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
sets=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU"
 "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for s in "${sets[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $s --kernel-trace --output-format csv -d "$out/pass$i" -- python3 tools/kbench.py --steps 2 "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/pass$i.log"; }
  echo "pass $i done"
done
python3 tools/pmc_summary.py "$out/pmc.json" "$out"/pass* > "$out/pmc.txt"
