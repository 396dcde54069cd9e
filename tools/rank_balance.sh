# per-rank share of the frame at N = 8, every rank rendered alone on this GPU: how even is the tile partition?
for r in 0 1 2 3 4 5 6 7; do
  timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --sim-world 8 --sim-rank $r 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rank $r', round(d['ms_per_step'],3), 'ms', int(d['rays_per_frame']), 'rays')"
done
