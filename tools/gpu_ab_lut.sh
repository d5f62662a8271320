#!/bin/bash
# kernel durations of the eager bench command (rocprofv3 kernel stats) -- every command bounded, nothing reads stdin
OUT=gpurun_out/${1:-r04l}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench --output-format csv -- python bench.py --steps 48 --warmup 5 --no-cpu-baseline --graph off > $OUT/bench_prof.json 2> $OUT/prof.err < /dev/null; echo "prof rc=$?"
find $OUT/prof -name '*kernel_trace*' -delete; find $OUT/prof -name '*domain_stats*' -delete
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then head -16 "$f" | cut -c1-180; fi
