#!/bin/bash
# band (two-ended / four-stream) against nested dissection per grid: tools/nd_crossover.sh
cd "$GRAFT_REPO_ROOT"
export C2_WARM=1 C2_REPS=3
for g in "3 24 300000" "3 32 1000000" "3 40 1000000" "3 48 2000000" "4 12 1000000" "4 16 2000000" "2 64 1000000" "2 256 4000000"; do
  for nd in 0 1; do echo "== grid $g ND=$nd"; SPLPAK_ND=$nd python tools/c2_profile.py $g 2>&1 | grep "ms per fit"; done
done
