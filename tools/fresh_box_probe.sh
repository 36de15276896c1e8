#!/bin/bash
# Runs ONCE per GPU box (marker in /tmp), before anything else has touched the GPU there: the handle-churn scenario with the graph
# epoch guard OFF and taps ON, first on the round-2 build that once reproduced the corruption (tools/_bisect/vT = f889722 + taps),
# then at HEAD.  The round-2 corruption was only ever seen in the first GPU process of a freshly acquired box.
mkdir -p gpurun_out/r03
if [ -e /tmp/opd_fresh_probe_done ]; then echo "fresh_box_probe: box already probed"; exit 0; fi
touch /tmp/opd_fresh_probe_done
log=gpurun_out/r03/fresh_probe_$(date +%H%M%S).log
echo "fresh box: weight cache entries $(ls /tmp/opd_weights 2>/dev/null | wc -l), uptime $(cut -d' ' -f1 /proc/uptime)" > $log
if [ -d tools/_bisect/f889722 ]; then   # the exact round-2 binary (guard patched out, no taps) as the box's FIRST GPU process
  echo "== f889722 as built in round 2, guard off" >> $log
  (cd tools/_bisect/f889722 && timeout -k 10 240 python ../poison_taps.py 2>&1 | grep -v -E "Warn|amdgpu.ids" | cut -c1-400) >> $log 2>&1
fi
if [ -d tools/_bisect/vT ]; then
  echo "== f889722 + taps" >> $log
  (cd tools/_bisect/vT && timeout -k 10 240 python ../poison_taps.py taps 2>&1 | grep -v -E "Warn|amdgpu.ids" | cut -c1-400) >> $log 2>&1
fi
echo "== HEAD" >> $log
timeout -k 10 240 python tools/graph_churn_probe.py handles_taps 2>&1 | grep -v -E "Warn|amdgpu.ids" | cut -c1-400 >> $log 2>&1
grep -E "diff|differing|after churn" $log | head -12
exit 0
