#!/bin/bash
# Frame time for several values of one environment knob of libmcpt (MCPT_FINISH_PATHS, MCPT_TRACE_MAX_CHUNK, MCPT_TRACE_MIN_CHUNK,
# MCPT_TRACE_BLOCK_RAYS, MCPT_LOGIC_GRID, MCPT_WORKSPACE_GB, MCPT_FAST_LEAF, MCPT_FAST_CT ...), on the GPU box from the repo root:
#   bash tools/knob_sweep.sh MCPT_FINISH_PATHS "200000 500000 1000000" [extra bench.py arguments, e.g. --sim-world 8]
knob=$1; values=$2; shift 2
for v in $values; do
  echo "$knob=$v"
  env $knob=$v timeout -k 10 150 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' ', round(d['ms_per_step'],3), 'ms/frame,', round(d['roofline']['avg_launch_ms'],3), 'ms per trace launch,', round(d['nodes_per_ray'],2), 'nodes and', round(d['tris_per_ray'],2), 'triangles per ray')"
done
