#!/bin/bash
# Which unit of the compute units a workload's kernels keep busy (GPU box): five rocprofv3 --pmc passes over a single-stream
# 64-pass step, summed per kernel by profiles/summarize_units.py.   tools/pmc_units.sh <workload> -> gpurun_out/units_<workload>/
set -u
W="${1:-c4}"
OUT=gpurun_out/units_$W
mkdir -p $OUT
pass() {
  V4R_PIPES=1 timeout -s KILL 240 rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-side --spp 64 > $OUT/$1.log 2>&1 || exit 1
}
pass active "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" &&
pass detail "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" &&
pass l1a "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TAGRAM0_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE" &&
pass l1b "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum GRBM_GUI_ACTIVE" &&
pass l1c "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE" &&
python3 profiles/summarize_units.py $OUT > $OUT/units.json && cat $OUT/units.json
