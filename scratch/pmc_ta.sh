#!/bin/bash
# TA / TCP / TD counters of one conv launch: usage scratch/pmc_ta.sh <tag> cin cout hw k B
tag=$1; shift
mkdir -p /root/repo/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
P1="TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE TA_BUFFER_READ_LDS_WAVEFRONTS_sum"
P2="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUSY_avr"
P3="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum"
P4="TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
i=1
for P in "$P1" "$P2" "$P3" "$P4"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /root/repo/gpurun_out/pmc_$tag/pass$i -- python3 /root/repo/scratch/one_conv.py "$@" > /root/repo/gpurun_out/pmc_$tag/log$i.txt 2>&1
  i=$((i+1))
done
