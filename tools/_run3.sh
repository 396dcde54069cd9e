#!/bin/bash
# round 4, call 3: suite on the current build, path-pool diagnostics, config 5 with / without prefetch + its PMC passes
root=$(pwd); out=$root/gpurun_out/c3; mkdir -p $out
export TMPDIR=/tmp
V=$root/montecarlopathtracing_amd/csrc/variants
echo "== tests"; date
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?"; tail -n 3 $out/tests.log
echo "== bench"; date
bench() { # label args
  MCPT_PRINT_DIAG=1 timeout -k 10 400 python bench.py --no-cpu-baseline $2 > $out/b_$1.json 2> $out/b_$1.err || { echo "$1 FAILED"; tail -n 5 $out/b_$1.err; return; }
  python - $1 $out <<'PY'
import json,sys
v,out=sys.argv[1],sys.argv[2]
d=json.load(open('%s/b_%s.json'%(out,v))); r=d['roofline']
print('%-22s ms/frame %.3f  trace avg ms %.3f  launches %d  nodes/ray %.2f frac %.3f' % (v, d['ms_per_step'], r['avg_launch_ms'], r['launches'], d['nodes_per_ray'], r['frac']))
PY
}
bench default_1 "--steps 5 --warmup 2"
bench default_2 "--steps 5 --warmup 2"
bench eighth_1 "--steps 10 --warmup 2 --sim-world 8"
export MCPT_LIB=$V/libmcpt_diag.so
bench diag_eighth "--steps 2 --warmup 1 --sim-world 8"; grep "path pool" $out/b_diag_eighth.err | tail -n 6
MCPT_FINISH_PATHS=134000000 bench diag_allpaths "--steps 1 --warmup 1 --spp 64"; grep "path pool" $out/b_diag_allpaths.err | tail -n 6
unset MCPT_LIB
echo "== config 5"; date
bench syn_prefetch "--scene synthetic --spp 16 --steps 3"
MCPT_PREFETCH_MIN_TRIS=1000000000 bench syn_plain "--scene synthetic --spp 16 --steps 3"
bench interior "--scene interior --steps 3"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  MCPT_PREFETCH_MIN_TRIS=1000000000 timeout -k 10 400 rocprofv3 --pmc $c -d $out/pmc_syn/$n -o pmc --output-format csv -- python3 $root/bench.py --scene synthetic --spp 16 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_syn_$n.log 2>&1
done
cd $root
python3 tools/pmc_summary.py $out/pmc_syn k_ > $out/pmc_syn_summary.txt 2>&1; grep -A6 "k_wf_trace<\|k_wf_logic" $out/pmc_syn_summary.txt | head -60
date
