#!/bin/bash
# The TITAN-side part of tools/profile_round.sh plus the default bench line (everything lands under gpurun_out/prof_$1/).
tag=${1:-r04}
out=gpurun_out/prof_$tag; mkdir -p $out
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python bench.py --kernel-times > $out/bench.json 2> $out/kernel_times.txt
echo bench done
python bench.py --config titan --patches 4096 --ragged --steps 16 --warmup 8 > $out/titan_bench.json 2> $out/titan_bench.err
rocprofv3 --kernel-trace --stats -d $out/titan_stats -o s --output-format csv -- python3 bench.py --config titan --patches 4096 --ragged --steps 16 --warmup 8 --no-cpu-baseline > $out/titan_stats.log 2>&1
find $out/titan_stats -name "*kernel_stats.csv" -exec cp {} $out/titan_kernel_stats.csv \;
echo titan done
bash tools/pmc.sh ${tag}_dense tools/dense_microbench.py
bash tools/pmc_hbm.sh ${tag}_dense tools/dense_microbench.py
python tools/pmc_summary.py gpurun_out/pmc_${tag}_dense dense_attn > $out/pmc_dense_attn.txt
echo profile set done
