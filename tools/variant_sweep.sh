# bench every tuning variant built under montecarlopathtracing_amd/csrc/variants/ (MCPT_LIB selects the library)
run() { timeout -k 10 150 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3))"; }
echo default; run
for f in montecarlopathtracing_amd/csrc/variants/libmcpt_*.so; do echo $f; MCPT_LIB=$PWD/$f run; done
