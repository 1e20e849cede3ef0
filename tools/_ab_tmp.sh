cd $GRAFT_REPO_ROOT
export EDGL_BENCH_SPIN_MS=60
for v in new old; do
if [ $v = old ]; then export EDGL_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_tnold.so; else unset EDGL_LIB_PATH; fi
OUT=$GRAFT_REPO_ROOT/gpurun_out/kt_$v; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $OUT -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 600 --warmup 50 --no-cpu-baseline --no-extras > $OUT/log.txt 2>&1)
echo "== $v"; python tools/kstats.py $OUT/k_results.db 650 | cut -c1-140 | head -24
grep -h '"metric"' $OUT/log.txt | cut -c1-200
rm -f $OUT/k_results.db
done
