#!/bin/bash
# round 4, call 4: can a logic block share a CU with the pool engine's workgroup?  (two frames in flight, logic grid of one block per CU)
root=$(pwd); out=$root/gpurun_out/c4; mkdir -p $out
export TMPDIR=/tmp
V=$root/montecarlopathtracing_amd/csrc/variants
bench() { # label args
  MCPT_PRINT_DIAG=1 timeout -k 10 400 python bench.py --no-cpu-baseline $2 > $out/b_$1.json 2> $out/b_$1.err || { echo "$1 FAILED"; tail -n 5 $out/b_$1.err; return; }
  python - $1 $out <<'PY'
import json,sys
v,out=sys.argv[1],sys.argv[2]
d=json.load(open('%s/b_%s.json'%(out,v))); r=d['roofline']
print('%-30s ms/frame %.3f  trace avg ms %.3f  launches %d' % (v, d['ms_per_step'], r['avg_launch_ms'], r['launches']))
PY
}
bench default "--steps 6 --warmup 2"
bench default_pipe "--steps 6 --warmup 2 --pipeline"
MCPT_LOGIC_GRID=256 bench default_pipe_g256 "--steps 6 --warmup 2 --pipeline"
export MCPT_LIB=$V/libmcpt_nolicm_all.so
bench nolicm "--steps 6 --warmup 2"
bench nolicm_pipe "--steps 6 --warmup 2 --pipeline"
MCPT_LOGIC_GRID=256 bench nolicm_pipe_g256 "--steps 6 --warmup 2 --pipeline"
MCPT_LOGIC_GRID=256 bench nolicm_g256 "--steps 6 --warmup 2"
MCPT_LOGIC_GRID=512 bench nolicm_pipe_g512 "--steps 6 --warmup 2 --pipeline"
MCPT_LOGIC_GRID=256 MCPT_FINISH_PATHS=1500000 bench nolicm_pipe_g256_t15 "--steps 6 --warmup 2 --pipeline"
cd /tmp
MCPT_LOGIC_GRID=256 timeout -k 10 300 rocprofv3 --kernel-trace -d $out/kt -o kt --output-format csv -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --pipeline > $out/kt.log 2>&1
cd $root
unset MCPT_LIB
bench syn_plain "--scene synthetic --spp 16 --steps 3"
bench interior "--scene interior --steps 3"
python3 - $out/kt/kt_kernel_trace.csv <<'PY' > $out/overlap.txt 2>&1
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void mcpt::","").replace("mcpt::","")) for r in rows))
t0 = ev[0][0]
# overlap between k_wf_trace_pool and k_wf_logic<false>
tr = [(s,e) for s,e,n in ev if n.startswith("k_wf_trace_pool")]
lg = [(s,e) for s,e,n in ev if n.startswith("k_wf_logic")]
ov = 0
for s,e in lg:
    for s2,e2 in tr:
        lo, hi = max(s,s2), min(e,e2)
        if hi > lo: ov += hi - lo
print("trace total %.2f ms, logic total %.2f ms, overlap %.2f ms, span %.2f ms" % (sum(e-s for s,e in tr)/1e6, sum(e-s for s,e in lg)/1e6, ov/1e6, (ev[-1][1]-t0)/1e6))
for s,e,n in ev[-120:-60]:
    print("%10.1f us  dur %8.1f  %s" % ((s-t0)/1e3, (e-s)/1e3, n[:40]))
PY
head -n 40 $out/overlap.txt
date
