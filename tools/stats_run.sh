cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stats_now; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-concurrent --no-kernel-timing > $O/stats_bench.json 2> $O/stats.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs cat | cut -d, -f1-4 | head -16
tail -c 300 $O/stats_bench.json
