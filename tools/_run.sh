mkdir -p gpurun_out/r3n
python -m pytest tests -q -m gpu -x > gpurun_out/r3n/tests.log 2>&1; rc=$?
tail -15 gpurun_out/r3n/tests.log | cut -c1-250
bash tools/_bench_variants.sh gpurun_out/r3n prev sq default
exit $rc
