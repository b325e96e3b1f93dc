#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_14; mkdir -p $O
tools/bin/gather_bw 2>&1 | tee $O/gather_bw_all.txt
