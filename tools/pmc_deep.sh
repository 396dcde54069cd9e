#!/bin/bash
# The wider hardware-counter set behind profiles/r04_pool_counters_deep.txt: six rocprofv3 --pmc passes (no other tracing) over one frame
# of the default bench, summarised per kernel.  usage (on the GPU box, from the repo root):  bash tools/pmc_deep.sh <outdir> [kernel substring]
set +e
out=${1:-gpurun_out/pmc_deep}; kern=${2:-k_wf_trace_pool}
root=$(pwd); mkdir -p "$root/$out"
export TMPDIR=/tmp
sets=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
 "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH"
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_WAVES SQ_ACCUM_PREV"
)
i=0
for s in "${sets[@]}"; do
  cd /tmp
  timeout -k 10 300 rocprofv3 --pmc $s -d "$root/$out/set$i" -o pmc --output-format csv -- python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$root/$out/set$i.log" 2>&1
  cd "$root"
  i=$((i+1))
done
python3 tools/pmc_summary.py "$out" "$kern" | tee "$out/summary.txt"
