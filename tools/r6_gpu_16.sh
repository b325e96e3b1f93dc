#!/bin/bash
# round 6, call 16: one awaited 2^20 proof with its accumulations handed over as the sorts are enqueued (the B2 accumulation starts at
# ~0.35 ms instead of ~0.78 ms: the host is still enqueuing calcH) against the shipped order; resident witness and host buffer
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_16; mkdir -p $O
for r in 1 2 3; do
  ( python3 tools/sync_single.py 20 16 | head -2 | sed 's/^/shipped: /' )
  ( export ZKR_EXP_EARLY=1; python3 tools/sync_single.py 20 16 | head -2 | sed 's/^/early:   /' )
done 2>&1 | tee $O/lone_proof_early.txt
( export ZKR_EXP_EARLY=1; python3 tools/rate_only.py 20 40 pipelined_early ) 2>&1 | grep '^\[' | tee -a $O/lone_proof_early.txt
( python3 tools/rate_only.py 20 40 pipelined_shipped ) 2>&1 | grep '^\[' | tee -a $O/lone_proof_early.txt
( export ZKR_EXP_EARLY=1; python3 tools/sync_single.py 22 8 | head -2 | sed 's/^/early 2^22:   /' ) 2>&1 | tee -a $O/lone_proof_early.txt
( python3 tools/sync_single.py 22 8 | head -2 | sed 's/^/shipped 2^22: /' ) 2>&1 | tee -a $O/lone_proof_early.txt
