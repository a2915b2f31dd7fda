#!/bin/bash
# usage: par.sh "ENV=.." : three concurrent instances of the probe, 10 runs each; prints the count of Y outcomes
V="$1"
for i in 1 2 3; do env $V python scripts/probes/nondet_probe.py 10 2>&1 | grep "^run" | awk '{print $7}' > /tmp/par_$i.txt & done
wait
cat /tmp/par_1.txt /tmp/par_2.txt /tmp/par_3.txt | awk -v v="$V" '{n++; if ($1+0 > 1e-5) y++} END {print v, ": Y", y+0, "of", n}'
