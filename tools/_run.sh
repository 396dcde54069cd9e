mkdir -p gpurun_out/r3y
python -m pytest tests -q -m gpu > gpurun_out/r3y/tests.log 2>&1 || { tail -30 gpurun_out/r3y/tests.log | cut -c1-250; exit 1; }
tail -3 gpurun_out/r3y/tests.log
