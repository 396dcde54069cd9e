#!/bin/bash
# Voting engine against the pool engine (MCPT_TRACE_ENGINE=pool) on one box, and the three builders of the culling hierarchy:
#   bash tools/engine_compare.sh <tag>   -> gpurun_out/<tag>/engine_compare.txt, builder_compare.txt, pool_kernel_stats.csv, pool_pmc.txt
tag=${1:-cmp}
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
one() { # engine build label args
  MCPT_PRINT_DIAG=1 MCPT_TRACE_ENGINE=$1 timeout -k 10 400 python3 bench.py --no-cpu-baseline --build $2 $4 > $out/j_$3.json 2> $out/j_$3.err
  python3 - <<PY
import json
d=json.load(open('$out/j_$3.json')); r=d['roofline']
print('%-34s %9.2f ms/frame  %7.3f ms/launch  %6.2f nodes + %5.2f triangles per ray  %8.1f Mrays/s' % ('$3', d['ms_per_step'], r['avg_launch_ms'], r['dom_nodes_per_ray'], r['dom_tris_per_ray'], d['value']))
PY
}
{
echo "build $(python3 -c 'import montecarlopathtracing_amd as M; print(M.build_id())')"
for e in vote pool; do
  one $e default ${e}_cornell-box "--steps 5 --warmup 2"
  one $e default ${e}_veach-mis_spp100 "--scene veach-mis --spp 100 --steps 4"
  one $e default ${e}_interior "--scene interior --steps 3"
  one $e default ${e}_synthetic10m_spp16 "--scene synthetic --spp 16 --steps 3"
  one $e default ${e}_cornell-box_one_eighth "--steps 10 --warmup 2 --sim-world 8"
done
grep -h "deferred" $out/j_pool_*.err | sort | uniq -c | sort -rn | head -3
} > $out/engine_compare.txt 2>&1
{
echo "build $(python3 -c 'import montecarlopathtracing_amd as M; print(M.build_id())')"
for b in host device_fast device_sah; do
  one vote $b ${b}_cornell-box "--steps 3 --warmup 1"
  one vote $b ${b}_veach-mis_spp100 "--scene veach-mis --spp 100 --steps 3"
  one vote $b ${b}_interior "--scene interior --steps 2"
done
for b in default device_fast device_sah; do
  one vote $b ${b}_synthetic10m_spp16 "--scene synthetic --spp 16 --steps 3"
  grep -h "device create" $out/j_${b}_synthetic10m_spp16.err | tail -6
done
} > $out/builder_compare.txt 2>&1
cd /tmp
MCPT_TRACE_ENGINE=pool timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/pool_stats -o p --output-format csv -- python3 $root/bench.py --no-cpu-baseline > $out/pool_stats.log 2>&1
MCPT_TRACE_ENGINE=pool timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d $out/pool_pmc/set0 -o pmc --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pool_pmc0.log 2>&1
MCPT_TRACE_ENGINE=pool timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE TCP_GATE_EN1_sum -d $out/pool_pmc/set1 -o pmc --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pool_pmc1.log 2>&1
cd $root
python3 tools/pmc_summary.py $out/pool_pmc k_wf_trace_pool > $out/pool_pmc.txt
cp $out/pool_stats/*kernel_stats.csv $out/pool_kernel_stats.csv 2>/dev/null || cp $out/pool_stats/*/*kernel_stats.csv $out/pool_kernel_stats.csv
cat $out/engine_compare.txt $out/builder_compare.txt $out/pool_pmc.txt
