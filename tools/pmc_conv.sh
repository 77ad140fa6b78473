#!/bin/bash
# PMC evidence for the wide conv kernels at their dominant shapes: separate rocprofv3 passes (SQ: 8 slots; FETCH_SIZE 3 TCC slots,
# WRITE_SIZE 2).  Usage (GPU box): bash tools/pmc_conv.sh r02   -> gpurun_out/pmc_conv_r02/summary.txt
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_conv_$R
rm -rf $O; mkdir -p $O
for w in dgrad_actbwd dgrad_plain dgrad_acc dgrad_nt2 fwd_pro1 fwd_pro0; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq_$w -- python tools/replay_conv.py $w 10 > /dev/null 2> $O/sq_$w.err
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq2_$w -- python tools/replay_conv.py $w 10 > /dev/null 2> $O/sq2_$w.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$w -- python tools/replay_conv.py $w 10 > /dev/null 2> $O/fetch_$w.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$w -- python tools/replay_conv.py $w 10 > /dev/null 2> $O/write_$w.err
done
python tools/pmc_conv_summary.py $O > $O/summary.txt
cat $O/summary.txt
