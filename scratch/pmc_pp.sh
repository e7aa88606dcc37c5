#!/bin/bash
# usage: scratch/pmc_pp.sh <tag> <SP_CONV_PP value> cin cout hw B   -> gpurun_out/pmc_<tag>/pass{1,2,3} + summary csv
tag=$1; pp=$2; shift; shift
export SP_CONV_PP=$pp
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"
P3="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES"
mkdir -p /root/repo/gpurun_out/pmc_$tag
i=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /root/repo/gpurun_out/pmc_$tag/pass$i -- python3 /root/repo/scratch/many_conv.py "$@" > /root/repo/gpurun_out/pmc_$tag/log$i.txt 2>&1
  i=$((i+1))
done
python3 /root/repo/scratch/pmc_reduce.py /root/repo/gpurun_out/pmc_$tag conv3x3 > /root/repo/gpurun_out/pmc_$tag/summary.txt
rm -rf /root/repo/gpurun_out/pmc_$tag/pass*
cat /root/repo/gpurun_out/pmc_$tag/summary.txt
