#!/bin/bash
# usage: tools/pmc_stages.sh <variant> <stage...> -- SQ counters of k_scan stopped after stage k (one PMC pass each)
v=$1; shift
cp build/variants/$v.so readsb_amd/libreadsb_amd.so || exit 1
for k in "$@"; do
  RSB_STAGE_LIMIT=$k bash tools/pmc.sh st$k SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES || exit 1
done
