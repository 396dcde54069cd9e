mkdir -p gpurun_out/r2x
python -m pytest tests -q -m gpu --durations=3 > gpurun_out/r2x/tests.log 2>&1; rc=$?
tail -8 gpurun_out/r2x/tests.log | cut -c1-220
exit $rc
