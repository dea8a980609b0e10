#!/bin/bash
# on the GPU box: `rocprofv3 --kernel-trace --stats` of the bench command itself (headline only: no sub-records, no CPU baseline) ->
# gpurun_out/prof_bench_<tag>_kernel_stats.csv (tools/prof_summary.py) + rocprofv3's own stats csv + the bench line
tag=${1:-r05}; root=$(pwd); export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench_$tag -o b -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --no_subs --no_cpu_baseline > $root/gpurun_out/prof_bench_$tag.json 2> $root/gpurun_out/prof_bench_$tag.err)
st=$(find /tmp/prof_bench_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$st" ] && cp "$st" $root/gpurun_out/prof_bench_${tag}_rocprof_kernel_stats.csv
head -6 $root/gpurun_out/prof_bench_${tag}_rocprof_kernel_stats.csv | cut -c1-200
tail -c 300 $root/gpurun_out/prof_bench_$tag.json
