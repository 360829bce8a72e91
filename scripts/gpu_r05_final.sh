#!/bin/bash
# round 5, final tree: the -m gpu suite, smoke, the driver's bench command, the default bench command
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05final; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "gpu tests rc $?"; tail -3 $O/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; python scripts/print_bench.py $O/bench_steps20.json | head -30
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python scripts/print_bench.py $O/bench_default.json | head -12
