#!/bin/bash
# Everything the round's measurement table is made of, on the GPU box from the repo root:
#   bash tools/final_profile.sh <tag>      -> gpurun_out/<tag>/...
# bench line (with CPU baseline), rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE and the SQ / TCP sets in
# separate --pmc passes, kernel timelines at N = 1 and at one eighth of the frame, per-rank share at N = 2, 4, 8 (--sim-world),
# the other configs (3, 4 and 5 at their own size).
set +e
tag=${1:-final}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py > $out/bench.json 2> $out/bench.err
# the dominant kernel of that run (the engine the library picked for the scene)
K=$(python3 -c "import json; print(json.load(open('$out/bench.json'))['roofline']['kernel'])")
cut -c1-300 $out/bench.json
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o p --output-format csv -- python3 $root/bench.py --no-cpu-baseline > $out/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/kt8 -o kt --output-format csv -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sim-world 8 > $out/kt8.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c -d $out/pmc/$c -o pmc --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_$c.log 2>&1
done
cd $root
python3 tools/pmc_to_traffic.py $out/pmc $K "python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (cornell-box 1280x720 SPP 256)" $out/hbm_traffic.json > /dev/null
python3 tools/pmc_summary.py $out/pmc k_wf > $out/pmc_hbm_traffic.txt
bash tools/pmc_trace.sh gpurun_out/$tag/pmc_sq > $out/pmc_sq.log 2>&1
python3 tools/pmc_issue.py $out/pmc_sq $K $out/issue_utilisation.json "python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (cornell-box 1280x720 SPP 256)"
python3 tools/timeline.py $out/kt8/kt_kernel_trace.csv 1 > $out/timeline_one_eighth.txt
for n in 2 4 8; do
  timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sim-world $n > $out/sim_world_$n.json 2>> $out/bench.err
  python3 -c "import json; d=json.load(open('$out/sim_world_$n.json')); print('sim-world $n', round(d['ms_per_step'],3), 'ms')"
done
# every rank's share of the 8-way partition, rendered alone (the slowest one is what an 8-GPU frame waits for, before the exchange)
for r in 0 1 2 3 4 5 6 7; do
  timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sim-world 8 --sim-rank $r > $out/sim8_rank$r.json 2>> $out/bench.err
  python3 -c "import json; d=json.load(open('$out/sim8_rank$r.json')); print('sim-world 8 rank $r', round(d['ms_per_step'],3), 'ms')"
done
timeout -k 10 200 python3 bench.py --scene veach-mis --spp 100 --steps 5 --no-cpu-baseline > $out/veach_mis_spp100.json 2>> $out/bench.err
timeout -k 10 300 python3 bench.py --scene interior --steps 3 --no-cpu-baseline > $out/interior_spp256.json 2>> $out/bench.err
timeout -k 10 400 python3 bench.py --scene synthetic --spp 16 --steps 3 --no-cpu-baseline > $out/synthetic10m_spp16.json 2>> $out/bench.err
timeout -k 10 600 python3 bench.py --scene synthetic --width 3840 --height 2160 --spp 1024 --steps 1 --warmup 0 --no-cpu-baseline > $out/synthetic10m_3840x2160_spp1024.json 2>> $out/bench.err
# FETCH_SIZE / WRITE_SIZE (and the L2 hit / miss counts) of the other profiled workloads, one frame each, program directly after "--":
# bench.py quotes them as roofline.traffic for those commands (same build-id rule as the headline's)
other_pmc() { # tag, kernel, bench arguments, command text
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    timeout -k 10 400 rocprofv3 --pmc $c -d $out/pmc_$1/$n -o pmc --output-format csv -- python3 $root/bench.py $3 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_$1_$n.log 2>&1
  done
  cd $root
  python3 tools/pmc_to_traffic.py $out/pmc_$1 $2 "python3 bench.py $3 --steps 1 --warmup 0 --no-cpu-baseline" $out/$1_hbm_traffic.json > /dev/null
  python3 tools/pmc_summary.py $out/pmc_$1 k_wf > $out/$1_pmc_hbm_traffic.txt
}
KV=$(python3 -c "import json; print(json.load(open('$out/veach_mis_spp100.json'))['roofline']['kernel'])")
KI=$(python3 -c "import json; print(json.load(open('$out/interior_spp256.json'))['roofline']['kernel'])")
KS=$(python3 -c "import json; print(json.load(open('$out/synthetic10m_spp16.json'))['roofline']['kernel'])")
other_pmc veach_mis $KV "--scene veach-mis --spp 100"
other_pmc interior $KI "--scene interior"
other_pmc synthetic10m $KS "--scene synthetic --spp 16"
# lanes per phase, iterations and the pre-test's share from the in-kernel counters (diagnostic build: variants/libmcpt_diag.so, tools/build_variant.sh diag "-DMCPT_TRACE_DIAG -DMCPT_POOL_DEBUG")
if [ -f montecarlopathtracing_amd/csrc/variants/libmcpt_diag.so ]; then
  MCPT_LIB=montecarlopathtracing_amd/csrc/variants/libmcpt_diag.so MCPT_PRINT_DIAG=1 timeout -k 10 200 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/diag.json 2> $out/diag.err
  grep -E "trace diag|k_wf_trace:|deferred|logic diag|finish diag|^pool " $out/diag.err | tail -24 > $out/trace_phases.txt
  cat $out/trace_phases.txt
fi
for f in veach_mis_spp100 interior_spp256 synthetic10m_spp16 synthetic10m_3840x2160_spp1024; do python3 -c "import json; d=json.load(open('$out/$f.json')); print('$f', round(d['ms_per_step'],2), 'ms', round(d['value'],1), 'Mrays/s', round(d['nodes_per_ray'],1), round(d['tris_per_ray'],1))"; done
