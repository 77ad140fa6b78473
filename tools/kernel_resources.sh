#!/bin/bash
# VGPRs / spills / scratch / occupancy of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage), demangled:
#   bash tools/kernel_resources.sh ms_conv_inst_a.hip [extra hipcc flags]
cd "$(dirname "$0")/../maxstyle_amd/csrc"
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -ffp-contract=off -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/kr.o 2>&1 \
 | grep -E "remark: +(Function Name|VGPRs:|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill)" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - - - - \
 | sed -e 's/Function Name: //' | while read -r name rest; do echo "$(echo $name | c++filt | sed 's/(ms::ConvArgs)//' | cut -c1-100) | $rest"; done
