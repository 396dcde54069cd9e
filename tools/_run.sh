mkdir -p gpurun_out/r2g
python tools/flip_probe.py 40000 > gpurun_out/r2g/flips.txt 2>&1; cat gpurun_out/r2g/flips.txt | head -8
