mkdir -p gpurun_out/r2f
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi_device or cpp_drop_in or render_scene_outputs" > gpurun_out/r2f/tests.log 2>&1 || { tail -60 gpurun_out/r2f/tests.log; exit 1; }
tail -3 gpurun_out/r2f/tests.log
