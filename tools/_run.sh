bash tools/_bench_variants.sh gpurun_out/r2e default nocull26 cull26
MCPT_LIB=$PWD/montecarlopathtracing_amd/csrc/variants/libmcpt_cull26.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "closest_hit or bulk or overflow or edge or pipelines or wavefront_iterations or image_matches" > gpurun_out/r2e/tests.log 2>&1 || { tail -40 gpurun_out/r2e/tests.log; exit 1; }
tail -3 gpurun_out/r2e/tests.log
