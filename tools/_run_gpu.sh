cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3c
for v in "" tools/variants/lib_nofb.so; do
EDGL_LIB_PATH=$v rocprofv3 --kernel-trace --stats -d gpurun_out/r3c/p -o s -- python tools/strip_bench.py > gpurun_out/r3c/log.txt 2>&1
python tools/kstats.py gpurun_out/r3c/p/s_results.db 4 | grep strip_kernel | cut -c1-120; rm -rf gpurun_out/r3c/p
done
