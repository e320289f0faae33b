#!/bin/bash
mkdir -p gpurun_out/pmc2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/$n -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc2/$n.log 2>&1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA
run sq3 SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_VALU_MFMA_MOPS_F16
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
ls $R/gpurun_out/pmc2
