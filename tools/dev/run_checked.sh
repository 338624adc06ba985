#!/bin/bash
# Runs a development command on the GPU box and FAILS when the GPU runtime reported a fault, whatever the command's own exit
# status was (round 2: a memory access fault inside tools/dev/k2f_debug.py left an exit status of 0 behind).
#   usage: tools/dev/run_checked.sh <log file> <command ...>
log=$1; shift
"$@" > "$log" 2>&1
rc=$?
if grep -q -E "Memory access fault|HSA_STATUS_ERROR|hipErrorIllegalAddress|Queue .* aborting" "$log"; then
  echo "run_checked: the GPU runtime reported a fault (see $log)" >&2
  [ $rc -eq 0 ] && rc=70
fi
exit $rc
