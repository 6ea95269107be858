#!/bin/bash
# Counter passes for the per-tier bootstrap kernels (run on the GPU box through gpurun; rocprofv3 gets the program itself
# after `--`, each --pmc group in its own pass, no tracing domains beside it).  usage: tools/pmc_collect.sh <tag> [tiers...]
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for group in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
             "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $group -d $out/pass$i -o pass$i -- python3 tools/pmc_tiers.py "$@" > $out/pass$i.log 2>&1 || { tail -5 $out/pass$i.log; exit 1; }
  echo "pass $i ($group) done"
done
python3 tools/pmc_table.py $out/table.txt pbs_kernel "$out/pass*/*/*.db" "$out/pass*/*.db"
