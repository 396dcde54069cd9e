mkdir -p gpurun_out/r3z
export MCPT_BENCH_SHARE_GPU=1
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3z/two_ranks.json 2> gpurun_out/r3z/two_ranks.err || { tail -20 gpurun_out/r3z/two_ranks.err; exit 1; }
cut -c1-600 gpurun_out/r3z/two_ranks.json
