mkdir -p gpurun_out/r3s
python -m pytest tests -q -m gpu -x > gpurun_out/r3s/tests.log 2>&1 || { tail -30 gpurun_out/r3s/tests.log | cut -c1-250; exit 1; }
tail -3 gpurun_out/r3s/tests.log
timeout -k 10 600 python tools/flip_probe.py > gpurun_out/r3s/flips.txt 2>&1 && cat gpurun_out/r3s/flips.txt | cut -c1-300
