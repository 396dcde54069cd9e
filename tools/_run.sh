mkdir -p gpurun_out/r3v
python -m pytest tests -q -m gpu > gpurun_out/r3v/tests.log 2>&1 || { tail -30 gpurun_out/r3v/tests.log | cut -c1-250; exit 1; }
tail -3 gpurun_out/r3v/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 300 python bench.py > gpurun_out/r3v/bench.json 2> gpurun_out/r3v/bench.err && cut -c1-400 gpurun_out/r3v/bench.json
