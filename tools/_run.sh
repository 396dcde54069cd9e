mkdir -p gpurun_out/r3f
python -m pytest tests -q -m gpu > gpurun_out/r3f/tests.log 2>&1; rc=$?
tail -4 gpurun_out/r3f/tests.log | cut -c1-220
[ $rc -eq 0 ] && bash tools/final_profile.sh r02_final4 2>&1 | tail -12
exit $rc
