mkdir -p gpurun_out/r2q
export TMPDIR=/tmp
root=$PWD
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/r2q/c -o p --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > $root/gpurun_out/r2q/c.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/r2q/v -o p --output-format csv -- python3 $root/bench.py --scene veach-mis --spp 100 --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > $root/gpurun_out/r2q/v.log 2>&1
cd $root
for s in c v; do echo $s; cut -d, -f1-4 gpurun_out/r2q/$s/p_kernel_stats.csv | head -9 | cut -c1-150; done
