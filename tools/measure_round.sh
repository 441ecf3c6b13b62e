#!/bin/bash
# Round measurements on the GPU box: bench lines for every workload, rocprofv3 kernel stats of the default bench command,
# (the kernel-stats run passes --no-others: the default command times four more workloads behind its headline region, whose launches
#  would be averaged into the same kernel names)
# PMC passes of the fused tile kernel at the cfg2b / cfg3 launch sizes, small-batch latency.  Results under gpurun_out/<tag>/;
# tools/pmc_summary.py + a copy into profiles/ follow on the build host.
#   tools/measure_round.sh r2
tag=${1:-r3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
python bench.py > $out/bench_cfg2b.json 2> $out/bench_cfg2b.err; echo "bench cfg2b rc=$?"
for w in cfg2 cfg3 cfg4 cfg5s; do python bench.py --workload $w --steps 10 --warmup 2 --cpu-frames 0 > $out/bench_$w.json 2> $out/bench_$w.err; echo "bench $w rc=$?"; done
for w in cfg2b cfg3; do python bench.py --workload $w --steps 10 --warmup 2 --cpu-frames 0 --no-others --tie-rule reference_queue > $out/bench_tie_$w.json 2> $out/bench_tie_$w.err; echo "bench $w (tie_rule reference_queue) rc=$?"; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 bench.py --steps 10 --warmup 2 --cpu-frames 0 --no-others > $out/stats.log 2>&1; echo "stats rc=$?"
bash tools/pmc_traffic.sh cfg2b --frames 4096
bash tools/pmc_traffic.sh cfg3 --model SMILy_Mouse_static_joints --frames 256 --views 18 --radius 4.0
python tools/latency_probe.py 2>&1 | grep "B=" > $out/latency.txt; cat $out/latency.txt
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
head -5 $out/kernel_stats.csv | cut -c1-150
