#!/bin/bash
# Hardware-counter passes over one frame of the default bench (each set in its own rocprofv3 run, no other tracing).
# usage (on the GPU box, from the repo root):  bash tools/pmc_trace.sh <outdir>
set -e
out=${1:-gpurun_out/pmc}
mkdir -p "$out"
root=$(pwd)
export TMPDIR=/tmp
sets=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS"
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
)
i=0
for s in "${sets[@]}"; do
  cd /tmp
  timeout -k 10 300 rocprofv3 --pmc $s -d "$root/$out/set$i" -o pmc --output-format csv -- python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$root/$out/set$i.log" 2>&1
  cd "$root"
  i=$((i+1))
done
python3 tools/pmc_summary.py "$out" k_wf > "$out/summary.txt"
cat "$out/summary.txt"
