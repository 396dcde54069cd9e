# frame time for several largest claim sizes of the persistent trace kernel's queue (tail of a big launch vs atomics)
for r in 2048 1024 512 256; do
  echo "max chunk $r"; MCPT_TRACE_MAX_CHUNK=$r timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3))"
done
