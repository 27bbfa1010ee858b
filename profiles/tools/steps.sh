#!/bin/bash
# Run GPU steps one after another on the gpurun box: "name|timeout_s|command" per argument.  Output of each step goes to
# gpurun_out/<name>.log.  A step that fails is reported and the next one still runs, but a step that was KILLED at its
# time limit (or by a signal) ends the session: nothing else is started on a GPU that may be wedged.
mkdir -p gpurun_out
for spec in "$@"; do
    name="${spec%%|*}"; rest="${spec#*|}"; tmo="${rest%%|*}"; cmd="${rest#*|}"
    echo "=== $name (limit ${tmo}s): $cmd"
    start=$(date +%s)
    timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "=== $name rc=$rc in $(( $(date +%s) - start ))s"
    tail -n 6 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then
        echo "=== $name was killed (rc=$rc): stopping the session"
        exit $rc
    fi
done
exit 0
