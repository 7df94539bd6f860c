#!/bin/bash
# end-of-round run: the default bench line, the GPU suite, smoke
mkdir -p gpurun_out/r5a
timeout 1500 python bench.py > gpurun_out/r5a/bench_line.json 2> gpurun_out/r5a/bench_err.txt; echo "bench rc=$?"
tail -c 600 gpurun_out/r5a/bench_err.txt
python scripts/show_bench_line.py gpurun_out/r5a/bench_line.json | cut -c1-400
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
