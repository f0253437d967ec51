#!/bin/bash
# k_owner's phases by JN_OWNER_DBG (1 no list loads, 2 no stores, 4 empty; results wrong, timing only), one slot under rocprofv3
export JN_STEREO_LIB=${JN_STEREO_LIB:-${GRAFT_REPO_ROOT:-$(pwd)}/jackal_navigation_amd/libjn_stereo_hooks.so}   # the switches used below exist in the hooks build only (csrc/hooks.h)
for d in ${1:-0 1 2 3 4}; do
  echo "JN_OWNER_DBG=$d: $(JN_OWNER_DBG=$d bash scripts/prof.sh owner_dbg$d 2>&1 | grep -E "k_owner|k_dense_row" | tr '\n' ' ')"
done
