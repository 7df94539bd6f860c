# CrystallinityUDF (row f3): the whole job through run_udf, the kernel alone per pixel type, the hipFFT route
# beside it, rocprofv3 kernel trace and SQ counters.  Usage (GPU box, repo root): bash scripts/profile_crystallinity.sh <tag>
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${tag}_crystallinity.txt
(
echo "== whole job (scripts/bench_crystallinity.py: 65 536 frames of 256 x 256 uint16 resident in HBM, Context.run_udf)"
python $R/scripts/bench_crystallinity.py 2>&1 | grep -v "amdgpu.ids"
LTMI_FFT_FUSED=0 python $R/scripts/bench_crystallinity.py 2>&1 | grep "Mframes" | sed 's/^/LTMI_FFT_FUSED=0: /'
echo "== kernel alone (scripts/bench_cryst_kernel.py: 16 384 frames, HIP events, median of 10)"
for d in uint8 int8 uint16 int16 uint32 int32 float32; do DTYPE=$d python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s"; done
MASK=0 python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s"
for r in 24 48 63 64 70 71 100 128; do RAD_OUT=$r python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s" | sed "s/^/rad_out $r: /"; done
LTMI_CRYST_WAVES=8 python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s" | sed 's/^/8 waves: /'
LTMI_FFT_FUSED=0 python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s"
echo "== 128 x 128 frames (k_cryst_fused128; 65 536 frames)"
for r in 16 32 48 64; do SIG=128 N=65536 RAD_OUT=$r python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s" | sed "s/^/rad_out $r: /"; done
for d in uint8 float32; do SIG=128 N=65536 RAD_OUT=32 DTYPE=$d python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s"; done
SIG=128 N=65536 RAD_OUT=32 MASK=0 python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s"
SIG=128 N=65536 RAD_OUT=32 LTMI_FFT_FUSED=0 python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s"
SIG=128 SCAN=512 python $R/scripts/bench_crystallinity.py 2>&1 | grep "Mframes"
SIG=128 SCAN=512 LTMI_FFT_FUSED=0 python $R/scripts/bench_crystallinity.py 2>&1 | grep "Mframes" | sed 's/^/LTMI_FFT_FUSED=0: /'
echo "== 512 x 512 frames (k_cryst_rows512 + k_cryst_cols512; 4 096 frames, 1 024 per pass of the workspace)"
for r in 64 128 256; do SIG=512 N=4096 RAD_OUT=$r python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s" | sed "s/^/rad_out $r: /"; done
for d in uint8 float32; do SIG=512 N=4096 RAD_OUT=128 DTYPE=$d python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s"; done
SIG=512 N=4096 RAD_OUT=128 LTMI_FFT_FUSED=0 python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s"
SIG=512 SCAN=128 python $R/scripts/bench_crystallinity.py 2>&1 | grep "Mframes"
SIG=512 SCAN=128 LTMI_FFT_FUSED=0 python $R/scripts/bench_crystallinity.py 2>&1 | grep "Mframes" | sed 's/^/LTMI_FFT_FUSED=0: /'
echo "== 1024 x 1024 frames (the same two kernels, four 256-point transforms per 1024 points; 1 024 frames)"
for r in 128 256 512; do SIG=1024 N=1024 RAD_OUT=$r python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s" | sed "s/^/rad_out $r: /"; done
SIG=1024 N=1024 RAD_OUT=256 DTYPE=float32 python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s"
SIG=1024 N=1024 RAD_OUT=256 LTMI_FFT_FUSED=0 python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s"
SIG=1024 SCAN=64 python $R/scripts/bench_crystallinity.py 2>&1 | grep "Mframes"
SIG=1024 SCAN=64 LTMI_FFT_FUSED=0 python $R/scripts/bench_crystallinity.py 2>&1 | grep "Mframes" | sed 's/^/LTMI_FFT_FUSED=0: /'
echo "== corrected frames (dark + gain + 50 dead pixels through the conversion pass, then the same kernels)"
for sg in 128 256 512; do for fu in 1 0; do CORR=1 SIG=$sg N=$((sg == 512 ? 4096 : 16384)) LTMI_FFT_FUSED=$fu python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s" | sed "s/^/sig $sg LTMI_FFT_FUSED=$fu: /"; done; done
echo "== timing-only ablations of k_cryst_fused<uint16,mask> (LTMI_CRYST_ABLATE; results are garbage)"
for a in 1 2 3 4 5; do LTMI_CRYST_ABLATE=$a python $R/scripts/bench_cryst_kernel.py 2>&1 | grep "frames/s" | sed "s/^/ablation $a: /"; done
) > $O 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_cryst -o c -- python $R/scripts/bench_crystallinity.py > $R/gpurun_out/${tag}_cryst.log 2>&1
cd $R
(echo "== rocprofv3 --kernel-trace --stats -- python scripts/bench_crystallinity.py"
 python scripts/rocpd_summary.py gpurun_out/${tag}_cryst/c_results.db | cut -c1-150 | head -12) >> $O 2>&1
rm -rf gpurun_out/${tag}_cryst
bash scripts/profile_cryst_sq.sh > /dev/null 2>&1
(echo "== SQ counters of k_cryst_fused<uint16,mask> (scripts/profile_cryst_sq.sh: one rocprofv3 --pmc pass per block)"; cat gpurun_out/cryst_sq.txt) >> $O
cat $O
