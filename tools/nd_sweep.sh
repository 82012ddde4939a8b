#!/bin/bash
# Factorisation-time sweep of the nested-dissection switches on one grid: tools/nd_sweep.sh [ndim nodes ndata]
cd "$GRAFT_REPO_ROOT"
export C2_WARM=1 C2_REPS=3
A="${1:-3} ${2:-64} ${3:-10000000}"
for cfg in "X=1" "SPLPAK_ND_NO_RAMP=1" "SPLPAK_ND_NO_ROOT_LOOKAHEAD=1" "SPLPAK_NO_PANEL_CU=1" "SPLPAK_ND_NO_RAMP=1 SPLPAK_ND_NO_ROOT_LOOKAHEAD=1 SPLPAK_NO_PANEL_CU=1" "SPLPAK_ND_PIN_ROUNDS=1" "SPLPAK_ND_RES_CUS=4" "SPLPAK_ND_KB=2"; do
  echo "== $cfg"; env $cfg python tools/c2_profile.py $A 2>&1 | grep "ms per fit"
done
