#!/bin/bash
# round 6, call 9: do random 64-byte gathers keep their rate when the table is 17 GB (one G1 table with a level per BIT, 256 x 64 B per
# point at 2^20) or 68 GB (four of them) instead of 0.87 GB?  (translation reach: the question behind width-w NAF digits over bit-level tables)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_09; mkdir -p $O
for cfg in "64 832 0 80" "64 17408 0 80" "64 69632 0 80" "128 832 0 80" "128 17408 0 80" "64 832 0 0" "64 17408 0 0" "64 69632 0 0"; do tools/bin/gather_bw $cfg; done 2>&1 | tee $O/gather_table_size.txt
