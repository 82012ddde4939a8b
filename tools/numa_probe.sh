#!/bin/bash
# where is GPU 0 attached, and does the fit time depend on the NUMA node the host thread runs on?
for d in /sys/class/drm/card*/device; do [ -f $d/numa_node ] && echo "$d numa_node=$(cat $d/numa_node) $(cat $d/uevent 2>/dev/null | grep PCI_SLOT_NAME)"; done | head -12
lscpu | grep -E "NUMA|Socket|^CPU\(s\)"
echo "allowed cpus: $(taskset -pc $$ | cut -d: -f2)"
rocm-smi --showtoponuma 2>/dev/null | grep -E "GPU\[0\]" | head -4
