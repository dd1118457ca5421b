#!/bin/bash
# usage: tools/pmc.sh <tag> <python script and args>: one counter group per rocprofv3 run, no tracing domains
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/pmc_$tag; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $out/a -o a --output-format csv -- python3 "$@" > $out/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM -d $out/b -o b --output-format csv -- python3 "$@" > $out/b.log 2>&1
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVES -d $out/c -o c --output-format csv -- python3 "$@" > $out/c.log 2>&1
echo done
