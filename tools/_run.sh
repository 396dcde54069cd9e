mkdir -p gpurun_out/r2o
python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r2o/tests.log 2>&1; rc=$?
tail -16 gpurun_out/r2o/tests.log | cut -c1-250
exit $rc
