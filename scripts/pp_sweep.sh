#!/bin/bash
# rebuilds the engine with different strip widths / workgroup sizes of the point x point gather and times config 4
set -e
cd "$GRAFT_REPO_ROOT"
for cfg in "1664 256" "1664 512" "3328 512" "832 256"; do
  set -- $cfg
  touch bundle-adjustment_amd/csrc/ba_kernels.h
  make -C bundle-adjustment_amd/csrc -j8 EXTRA="-DJAICOV_PP_CW=$1 -DJAICOV_PP_NT=$2" > /dev/null 2>&1
  echo "== PP_CW=$1 PP_NT=$2"
  python scripts/stage_times.py cfg4 2>&1 | grep -E "it2|it3"
done
