#!/bin/bash
# C3 factorisation time by schedule cut (round 5): tools/r05_c3_cut.sh
cd "$GRAFT_REPO_ROOT"
for cfg in "X=1" "SPLPAK_ND_CUT=1" "SPLPAK_ND_CUT=2" "SPLPAK_ND_CUT=3" "SPLPAK_ND_CUT=4" "SPLPAK_ND_SQUARE=1" "X=2"; do
  echo "== $cfg"; env $cfg python tools/nd_repeat.py 3 64 10000000 5 2>&1 | grep "^fit"
done
