#!/bin/bash
# SQ / TCP / TCC counter passes of bench.py on the GPU box -> gpurun_out/<tag>/ (summaries are copied to profiles/ by hand).
# usage (through gpurun): bash tools/collect_pmc.sh r04pmc [cfg2|m15|cfg3|cfg4 ...]
# Every pass is its own rocprofv3 run with --kernel-trace only beside --pmc (MI355X_MICROARCH.md, "rocprofv3 PMC slots":
# 8 SQ slots, 4 TCC slots per pass); the program itself follows "--".
set -o pipefail
tag=${1:-r04pmc}; shift
what=${*:-cfg2 m15 cfg3 cfg4}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
SQB="SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA"
SQC="SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_FMA_F64"
TCP="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
TCC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
run() {  # name, counters, bench args
    local name=$1 ctr=$2; shift 2
    echo "[pmc] $name"
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/$name -o t -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-e2e --no-extra "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
    find $O/$name -name "*kernel_trace.csv" -size +8M -delete
}
for w in $what; do
    case $w in
    cfg2) A=""; P="sqA sqB sqC tcp tcc" ;;
    m15)  A="--neighbors 15"; P="sqA sqB tcp tcc" ;;
    cfg3) A="--contigs 500000 --dim 140 --bins 128"; P="sqA sqB" ;;
    cfg4) A="--contigs 1000000 --dim 146 --bins 200"; P="sqA sqB" ;;
    *) echo "unknown $w"; exit 2 ;;
    esac
    for p in $P; do
        case $p in sqA) C=$SQA ;; sqB) C=$SQB ;; sqC) C=$SQC ;; tcp) C=$TCP ;; tcc) C=$TCC ;; esac
        run ${w}_$p "$C" $A || exit 1
    done
    python3 $R/tools/pmc_report.py $O/${w}_ > /dev/null 2>&1
done
for w in $what; do
    echo "==== $w" >> $O/summary.txt
    for d in $O/${w}_*/; do python3 $R/tools/pmc_report.py $d >> $O/summary.txt 2>&1; done
done
echo done
