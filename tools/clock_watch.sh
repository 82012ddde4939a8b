#!/bin/bash
# samples sclk / power of GPU 0 while a command runs:  tools/clock_watch.sh <logfile> <command...>
log=$1; shift
"$@" &
pid=$!
while kill -0 $pid 2>/dev/null; do
    echo "$(date +%s.%N) $(rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power|fclk|mclk' | tr -s ' ' | tr '\n' '|')" >> $log
    sleep 0.25
done
wait $pid
