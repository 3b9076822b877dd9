for a in "4096 4 50000" "4096 64 2800000"; do
  echo "== default, $a"; timeout 120 python scripts/dbg_din.py $a 2>&1 | tail -8
  echo "== ROCPRIM_USE_ATOMIC_BLOCK_ID=0, $a"; ROCPRIM_USE_ATOMIC_BLOCK_ID=0 timeout 120 python scripts/dbg_din.py $a 2>&1 | tail -8
done
