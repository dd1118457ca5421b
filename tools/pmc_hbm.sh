#!/bin/bash
# HBM traffic counters (separate passes: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2): tools/pmc_hbm.sh <tag> <script> [args]
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/pmc_$tag; mkdir -p $out
rocprofv3 --pmc FETCH_SIZE -d $out/f -o f --output-format csv -- python3 "$@" > $out/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/w -o w --output-format csv -- python3 "$@" > $out/w.log 2>&1
echo done
