for leaf in 4 2 1; do for ct in 1 0.5 2; do
echo "leaf=$leaf ct=$ct"; MCPT_FAST_LEAF=$leaf MCPT_FAST_CT=$ct timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['nodes_per_ray'],2), round(d['tris_per_ray'],2), round(d['roofline']['avg_launch_ms'],3))"
done; done
