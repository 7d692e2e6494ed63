cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_nolap
XZAMD_NO_OVERLAP=1 XZAMD_SPAN_AUTO=1 timeout 250 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_nolap/trace -o b --output-format csv -- python tools/prof_case.py 1368 6 2 > gpurun_out/prof_nolap/log.txt 2>&1
find gpurun_out/prof_nolap/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_nolap/kernel_stats.csv \;
find gpurun_out/prof_nolap/trace -type f -size +2M -delete
tail -3 gpurun_out/prof_nolap/log.txt
