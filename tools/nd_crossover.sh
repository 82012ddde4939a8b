#!/bin/bash
# band (two-ended / four-stream) against nested dissection per grid: tools/nd_crossover.sh [small]
cd "$GRAFT_REPO_ROOT"
export C2_WARM=2 C2_REPS=10
if [ "$1" = "small" ]; then
  GRIDS=("2 64 1000000" "2 48 500000" "2 90 1000000" "3 16 100000" "3 20 200000" "4 8 100000" "4 10 300000" "2 128 2000000")
else
  GRIDS=("3 24 300000" "3 32 1000000" "3 40 1000000" "3 48 2000000" "4 12 1000000" "4 16 2000000" "2 64 1000000" "2 256 4000000")
fi
for g in "${GRIDS[@]}"; do
  for nd in 0 1; do echo "== grid $g ND=$nd"; SPLPAK_ND=$nd python tools/c2_profile.py $g 2>&1 | grep "ms per fit"; done
done
