#!/bin/bash
# round 6 experiments: (a) the pitch of single=1's transposed buffer (a multiple of 8 values, or of a 128-byte line), (b) genes per workgroup of normvar's moments pass
export TMPDIR=/tmp
O=gpurun_out/r06l
mkdir -p $O
for rep in 1 2 3; do
	for al in 8 32; do
		echo "ldye multiple of $al (rep $rep): $(NRM_S1_LDYE_ALIGN=$al python bench.py --workload de_c4_single1 --steps 50 --warmup 5 --no-extras --cpu-seconds 0 --e2e 0 2>&1 | grep '^{"workload_detail' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), d['kernels_ms'])")" >> $O/ldye.txt
	done
done
cat $O/ldye.txt
for rep in 1 2; do
	for g in 1 2 4; do
		echo "NV_G=$g (rep $rep): $(python tools/with_lib.py tools/exp/nrm_normvar_NV_G_$g.so bench.py --workload normvar_c2 --steps 20 --warmup 3 --no-extras --cpu-seconds 0 --e2e 0 2>&1 | grep '^{"workload_detail' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), d['kernels_ms'])")" >> $O/nvg.txt
	done
done
cat $O/nvg.txt
python -m pytest tests/test_gpu_round2.py -q -x -k "bench_self" > $O/t_bench.log 2>&1; echo "rc=$?" >> $O/t_bench.log; tail -n 5 $O/t_bench.log
