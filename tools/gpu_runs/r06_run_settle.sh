#!/bin/bash
# the headline's 20 timed steps after 5, 50 and 200 untimed ones, alternating on one box: how much of a short run's ms_per_step is the chip's clocks still rising
B="python3 bench.py --steps 20 --no-extras --cpu-seconds 0 --e2e 0"
for rep in 1 2 3; do
for w in 5 50 200; do
	echo -n "warmup $w: "; $B --warmup $w 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['kernels_ms'])"
done
done
