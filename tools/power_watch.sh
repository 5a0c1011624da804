#!/bin/bash
# Samples the package power, the shader clock and the busy figures rocm-smi reports while a command runs (the command after --), 4 samples per second, into the file given first:
#   tools/power_watch.sh out.txt -- python3 tools/unet_eval.py sdxl 128 8 200
# (ordinary user: read-only queries).  Evidence for DESIGN section 7: the SDXL evaluation runs at the board's power limit with the shader clock pulled down.
out=$1; shift; shift
( while true; do rocm-smi --showpower --showclocks --showuse --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' '; echo; sleep 0.25; done ) > "$out" &
W=$!
"$@"; rc=$?
kill $W 2>/dev/null; wait $W 2>/dev/null
exit $rc
