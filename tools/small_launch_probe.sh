# per-rank frame time at one eighth of the frame (and the whole frame) for several hand-over points to the finishing kernel
for r in 200000 800000 1600000 3200000 6400000; do
  echo "finish below $r paths"; MCPT_FINISH_PATHS=$r timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --sim-world 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"
done
for r in 200000 1600000 6400000 25000000; do
  echo "N=1 finish below $r paths"; MCPT_FINISH_PATHS=$r timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"
done
