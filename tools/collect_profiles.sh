#!/bin/bash
# copies what is judged from gpurun_out/<tag>/ (scratch) into profiles/ (tracked):  bash tools/collect_profiles.sh <tag> <prefix>
set -e
t=gpurun_out/$1; p=profiles/$2
cp $t/bench.json ${p}_bench.json
cp $t/stats/p_kernel_stats.csv ${p}_kernel_stats.csv
cp $t/hbm_traffic.json ${p}_hbm_traffic.json
cp $t/issue_utilisation.json ${p}_issue_utilisation.json
cp $t/pmc_hbm_traffic.txt ${p}_pmc_hbm_traffic.txt
cp $t/pmc_sq/summary.txt ${p}_pmc_sq_tcp.txt
cp $t/timeline_one_eighth.txt ${p}_timeline_one_eighth_frame.txt
[ -f $t/trace_phases.txt ] && cp $t/trace_phases.txt ${p}_trace_phases.txt
# the other profiled workloads' traffic (bench.py quotes profiles/rNN_<tag>_hbm_traffic.json for their commands)
rr=${p%_final}
for w in veach_mis interior synthetic10m; do
  [ -f $t/${w}_hbm_traffic.json ] && cp $t/${w}_hbm_traffic.json ${rr}_${w}_hbm_traffic.json && cp $t/${w}_pmc_hbm_traffic.txt ${rr}_${w}_pmc_hbm_traffic.txt
done
python3 - $t ${p}_other_configs.json <<'PY'
import json, sys
t, out = sys.argv[1:3]
rec = {}
for name in ("sim_world_2", "sim_world_4", "sim_world_8", "sim8_rank0", "sim8_rank1", "sim8_rank2", "sim8_rank3", "sim8_rank4", "sim8_rank5", "sim8_rank6", "sim8_rank7", "veach_mis_spp100", "interior_spp256", "synthetic10m_spp16", "synthetic10m_3840x2160_spp1024"):
    try:
        d = json.load(open("%s/%s.json" % (t, name)))
    except Exception as e:
        rec[name] = {"error": str(e)}
        continue
    rec[name] = {k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "rays_per_frame", "nodes_per_ray", "tris_per_ray", "config")}
    rec[name]["roofline"] = {k: d["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "launches", "record_bytes_rate_GBs")}
    rec[name]["build_id"] = d.get("build_id")
json.dump(rec, open(out, "w"), indent=1)
PY
ls -la ${p}_*
