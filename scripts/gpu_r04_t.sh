#!/bin/bash
# round 4, GPU call T: the committed tree once more - GPU suite, smoke, the driver's bench command
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04t; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_steps20.json 2> $O/bench_line_steps20.err; echo "bench rc $?"
python scripts/print_bench.py $O/bench_line_steps20.json 2>/dev/null | head -3
