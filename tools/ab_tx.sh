# tx circuit (BatchProcessTx(2, 6), fused batches of 128 proofs) under knob settings, same box: bash tools/ab_tx.sh <rounds> "name:ENV=val,..." ...
N=$1; shift
for r in $(seq 1 $N); do
  for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    ( IFS=,; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; unset IFS
      echo -n "$name round $r: "; TX_COUNTS=128 python tools/tx_profile.py 2>/dev/null | grep "^128 proofs" | cut -c1-200 )
  done
done
