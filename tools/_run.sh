mkdir -p gpurun_out/r2h
python -m pytest tests -q -m gpu --durations=12 > gpurun_out/r2h/tests.log 2>&1; rc=$?
tail -32 gpurun_out/r2h/tests.log
exit $rc
