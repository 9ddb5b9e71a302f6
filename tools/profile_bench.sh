#!/bin/bash
# Profiles ONE bench command three times (kernel trace + the two PMC passes, separate runs as MI355X_MICROARCH.md prescribes) and
# writes profiles/<tag>_k4_profile.json (tools/prof_summary.py) plus the kernel-stats CSV.  Run on the GPU box:
#     bash tools/profile_bench.sh r02
set -e
tag=${1:-r02}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out/trace $out/fetch $out/write
args="bench.py --no-cpu-baseline --steps 150 --warmup 15"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 $args > $out/trace_bench.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 $args > $out/fetch_bench.json 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 $args > $out/write_bench.json 2> $out/write.err
python3 tools/prof_summary.py $tag $out/trace $out/fetch $out/write
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) profiles/${tag}_bench_kernel_stats.csv
# (gpurun merges gpurun_out/ back; profiles/ is part of the repository snapshot and does not travel back by itself)
cp profiles/${tag}_k4_profile.json profiles/${tag}_bench_kernel_stats.csv $out/
