#!/bin/bash
# Runs a command repeatedly; when a run exceeds 15 s, prints its host stacks (gdb) and kills it.
for i in $(seq 1 ${RUNS:-40}); do
  "$@" > /tmp/hang_out.txt 2>&1 &
  pid=$!
  for t in $(seq 1 30); do
    sleep 0.5
    if ! kill -0 $pid 2>/dev/null; then break; fi
  done
  if kill -0 $pid 2>/dev/null; then
    echo "=== run $i hangs (pid $pid); last output:"; tail -3 /tmp/hang_out.txt
    if [ -x /opt/rocm/bin/rocgdb ]; then /opt/rocm/bin/rocgdb -p $pid -batch -ex "thread apply all bt 14" 2>&1 | grep -v "^\[New\|warning\|Reading\|Loaded\|debuginfo" | head -90; fi
    kill -9 $pid; wait $pid 2>/dev/null
    break
  fi
  wait $pid
done
echo probe done
