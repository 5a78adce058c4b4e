#!/bin/bash
# Which SDMA engine carries the input (engineType=2) and the result (engineType=1) copies of valign_hip_align_host?
# Three fresh processes; AMD_LOG_LEVEL=4 lines of the runtime's copy path, counted.  (developer tool, GPU box)
for i in 1 2 3; do
  AMD_LOG_LEVEL=4 python3 tools/align_copy_log.py > /dev/null 2> /tmp/copy_log.err
  echo "== process $i"
  sed -n "/=== TRACED CALL/,/=== END/p" /tmp/copy_log.err | grep "HSA Copy" | sed -e "s/.*HSA Copy/HSA Copy/" -e "s/dst=.*forceSDMA/forceSDMA/" -e "s/wait_event.*//" | sort | uniq -c
done
rm -f /tmp/copy_log.err
