#!/bin/bash
# Counter study of one workload's kernels next to a pure stream of the same bytes (tools/pmc_probe.py): one rocprofv3 --pmc
# pass per counter group (never combined with tracing), csv output under gpurun_out/$OUT/<group>; summary:
#   python3 tools/pmc_study_summary.py gpurun_out/$OUT > gpurun_out/$OUT/summary.txt
# usage (on the GPU box): WORKLOAD=c3 OUT=r05pmc [ONLY="tcc_hit tcc_ea"] bash tools/pmc_study.sh      (every pass under its own timeout)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${OUT:-pmc_study}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$OUT
run() { [ -n "$ONLY" ] && [[ " $ONLY " != *" $1 "* ]] && return; timeout -k 10 ${PASS_TIMEOUT:-150} rocprofv3 --pmc $2 --output-format csv -d $R/gpurun_out/$OUT/$1 -- python3 $R/tools/pmc_probe.py > $R/gpurun_out/$OUT/$1.log 2>&1; echo "$1 rc=$?"; }
run sq_wait   "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
run sq_level  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
run sq_fifo   "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES"
run tcp_lat   "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
run tcp_stall "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"
run tcp_tlb   "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"
run tcp_wr    "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
# (a pass with TA_* and TD_* counters together aborted inside rocprofv3 on this image -- signal 6 -- and left the run hanging: TA alone, under the per-pass timeout)
run ta        "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
run tcc_hit   "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_STREAMING_REQ_sum"
run tcc_ea    "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"
run tcc_wr    "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum"
run tcc_stall "TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_BUSY_sum"
run grbm      "GRBM_GUI_ACTIVE GRBM_TA_BUSY"
