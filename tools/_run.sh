mkdir -p gpurun_out/r2l
for v in diagshare diagnoshare; do
MCPT_PRINT_DIAG=1 MCPT_LIB=$PWD/montecarlopathtracing_amd/csrc/variants/libmcpt_$v.so timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2l/$v.json 2> gpurun_out/r2l/$v.err
echo $v; grep "trace diag" gpurun_out/r2l/$v.err | tail -1
done
