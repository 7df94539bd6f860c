R=${GRAFT_REPO_ROOT:-/root/repo}
python $R/scripts/bench_crystallinity.py 2>&1 | grep -v "amdgpu.ids"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01_cryst -o c -- python $R/scripts/bench_crystallinity.py > $R/gpurun_out/r01_cryst.log 2>&1
cd $R && python scripts/rocpd_summary.py gpurun_out/r01_cryst/c_results.db | cut -c1-150 | head -14
