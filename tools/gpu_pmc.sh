#!/bin/bash
# profile evidence for profiles/: kernel-trace stats, then PMC groups in separate passes (never combined with sys/hip traces)
# usage: tools/gpu_pmc.sh [OUTNAME [bench.py args...]]   (default: pmc_final, the default C1 workload)
NAME=${1:-pmc_final}; shift || true
EXTRA="$@"
mkdir -p gpurun_out/$NAME
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$NAME
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-two-streams $EXTRA > $O/stats.log 2>&1
run() { n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-p24 --no-f32 --no-two-streams --spinup-ms 0 $EXTRA > $O/$n.log 2>&1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
ls $O
