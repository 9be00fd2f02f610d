#!/bin/bash
# Runs the command under rocgdb until a run hangs (15 s), then interrupts it and prints the host stacks.
for i in $(seq 1 ${RUNS:-30}); do
  /opt/rocm/bin/rocgdb -q -batch -ex "set pagination off" -ex run -ex "thread apply all bt 16" --args "$@" > /tmp/gdb_out.txt 2>&1 &
  pid=$!
  hung=1
  for t in $(seq 1 40); do
    sleep 0.5
    if ! kill -0 $pid 2>/dev/null; then hung=0; break; fi
  done
  if [ $hung = 1 ]; then
    echo "=== run $i hangs"
    # interrupt the inferior: gdb then runs the bt command
    pkill -INT -P $pid 2>/dev/null
    sleep 8
    grep -v "^\[New\|^\[Thread\|warning\|Reading\|Loaded\|debuginfo\|^  cp\|^step\|inserted" /tmp/gdb_out.txt | tail -120
    kill -9 $pid 2>/dev/null
    break
  fi
done
echo probe2 done
