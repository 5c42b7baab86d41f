#!/bin/bash
# A/B of prebuilt library variants (build/variants/*.so): stage ablation of the scan kernel for each.
# Run on the GPU box from the repo root; the box's copy of the tree is scratch, so overwriting the library there is fine.
for v in "$@"; do
  cp build/variants/$v.so readsb_amd/libreadsb_amd.so || exit 1
  echo "== $v"
  bash tools/ablate.sh || exit 1
done
