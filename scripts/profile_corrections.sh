R=${GRAFT_REPO_ROOT:-/root/repo}
python $R/scripts/bench_corrections.py 2>&1 | grep -v "amdgpu.ids\|warning limit"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01_corr -o c -- python $R/scripts/bench_corrections.py > $R/gpurun_out/r01_corr.log 2>&1
cd $R && python scripts/rocpd_summary.py gpurun_out/r01_corr/c_results.db | cut -c1-150 | head -10
python -m pytest tests/test_udf_gpu.py -m gpu -x -q -k corr 2>&1 | tail -3
