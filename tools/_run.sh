mkdir -p gpurun_out/r2n
export TMPDIR=/tmp
root=$PWD
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $root/gpurun_out/r2n/kt8 -o kt --output-format csv -- python3 $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --sim-world 8 > $root/gpurun_out/r2n/kt8.log 2>&1
cd $root
tail -1 gpurun_out/r2n/kt8.log | cut -c1-200
