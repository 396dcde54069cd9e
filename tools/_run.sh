mkdir -p gpurun_out/r2r
timeout -k 10 120 ./tools/probes/gather_probe > gpurun_out/r2r/gather.txt 2>&1; cat gpurun_out/r2r/gather.txt
