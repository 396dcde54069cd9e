mkdir -p gpurun_out/r3a
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipelines or wavefront_iterations or image_matches or edge or partition or chunked or multi_device or pipelined or published" > gpurun_out/r3a/tests.log 2>&1 || { tail -30 gpurun_out/r3a/tests.log | cut -c1-250; exit 1; }
tail -3 gpurun_out/r3a/tests.log
bash tools/_bench_variants.sh gpurun_out/r3a default
