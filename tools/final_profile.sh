#!/bin/bash
# Everything the round's measurement table is made of, on the GPU box from the repo root:
#   bash tools/final_profile.sh <tag>      -> gpurun_out/<tag>/...
# bench line (with CPU baseline), rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE in separate --pmc passes,
# per-rank share at N = 2, 4, 8 (--sim-world), the other configs.
set -e
tag=${1:-final}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py > $out/bench.json 2> $out/bench.err
cat $out/bench.json | cut -c1-300
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o p --output-format csv -- python3 $root/bench.py --no-cpu-baseline > $out/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c -d $out/pmc/$c -o pmc --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_$c.log 2>&1
done
cd $root
python3 tools/pmc_to_traffic.py $out/pmc k_wf_trace "python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (cornell-box 1280x720 SPP 256)" $out/hbm_traffic.json > /dev/null
python3 tools/pmc_summary.py $out/pmc k_wf > $out/pmc_hbm_traffic.txt
for n in 2 4 8; do
  timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sim-world $n > $out/sim_world_$n.json 2>> $out/bench.err
  python3 -c "import json; d=json.load(open('$out/sim_world_$n.json')); print('sim-world $n', round(d['ms_per_step'],3), 'ms')"
done
timeout -k 10 200 python3 bench.py --scene veach-mis --spp 100 --steps 5 --no-cpu-baseline > $out/veach_mis_spp100.json 2>> $out/bench.err
timeout -k 10 300 python3 bench.py --scene interior --steps 3 --no-cpu-baseline > $out/interior_spp256.json 2>> $out/bench.err
timeout -k 10 400 python3 bench.py --scene synthetic --spp 16 --steps 3 --no-cpu-baseline > $out/synthetic10m_spp16.json 2>> $out/bench.err
for f in veach_mis_spp100 interior_spp256 synthetic10m_spp16; do python3 -c "import json; d=json.load(open('$out/$f.json')); print('$f', round(d['ms_per_step'],2), 'ms', round(d['value'],1), 'Mrays/s', round(d['nodes_per_ray'],1), round(d['tris_per_ray'],1))"; done
