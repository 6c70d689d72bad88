#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box:   bash profiles/collect.sh r01c ["trace fetch write sq bench"]
# (run through gpurun; outputs land in gpurun_out/<tag>_*; then `python profiles/summarize.py <tag>` turns them into profiles/<tag>_*).
# Counters are collected in their own passes with --kernel-trace only (never combined with sys/hip/hsa traces).
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
# the DRIVER's command (bench.py --gpus 1 --steps 20 --warmup 5) without the legs that add other kernels / minutes to a trace: the profile must
# reproduce the driver's bench line (round-4 review: the 5-step profile sat 8 % below it -- one cold launch in five, a colder clock)
CMD=${CMD:-"python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extractors --no-small-batches --no-score-fwd"}      # e.g. CMD="python $ROOT/bench.py --lmax 3 --steps 5 --warmup 1 --no-cpu-baseline --no-extractors --no-small-batches" for another workload
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=${2:-"trace fetch write sq sq2 sq3 tcp bench"}     # each PMC pass costs ~3.5 min of box time
has() { [[ " $PASSES " == *" $1 "* ]]; }
has trace && rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -- $CMD > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.log
pass() {   # name, counters...
    local name=$1; shift
    has $name && rocprofv3 --kernel-trace --pmc "$@" -d $OUT/${TAG}_pmc_$name -- $CMD > /dev/null 2> $OUT/${TAG}_pmc_$name.log
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD
pass sq3 SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_SALU
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
has bench && python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo done
