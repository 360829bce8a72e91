#!/bin/bash
# round 4, GPU call Q: the final tree - the whole GPU suite, the bench line with the driver's arguments and with the default ones
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04q; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line_steps20.json 2> $O/bench_line_steps20.err; echo "bench steps20 rc $?"
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench default rc $?"
python scripts/print_bench.py $O/bench_line_steps20.json 2>/dev/null | head -60
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
