#!/bin/bash
# profile evidence for the BACKWARD (profiles/r03_bwd_*): kernel-trace stats of tools/bench_bwd.py, then PMC groups in separate
# passes (never combined with sys/hip traces). usage: tools/gpu_bwd_pmc.sh [OUTNAME]   env BATCH / DISTINCT / SIZE / NF as bench_bwd.py;
# ONLY="stats atom" restricts the passes (default: all)
NAME=${1:-pmc_bwd}
mkdir -p gpurun_out/$NAME
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$NAME
want() { [ -z "$ONLY" ] || [[ " $ONLY " == *" $1 "* ]]; }
want stats && ITERS=20 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/bench_bwd.py > $O/stats.log 2>&1
run() { n=$1; shift
  want $n || return 0
  ITERS=3 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python3 $R/tools/bench_bwd.py > $O/$n.log 2>&1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run atom TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_EA0_WRREQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
tail -2 $O/stats.log
ls $O
