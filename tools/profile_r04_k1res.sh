#!/bin/bash
# Round-4 profiles, second part: K1 with the rows resident on chip (opt-in, NRM_K1=res) beside the default kernel on the configs[3] rows --
# kernel time, HBM-side counters, the per-phase time stamps -- plus binnet's phase timing and the Infinity-Cache probe.
export TMPDIR=/tmp
O=gpurun_out/r04prof
mkdir -p $O
B="python3 bench.py --cpu-seconds 0 --e2e 0 --no-extras"
NRM_K1=res timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/de_c4_res_stats -o res -- $B --workload de_c4 --steps 5 --warmup 2 > /dev/null 2> $O/res_stats.err
f=$(find $O/de_c4_res_stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/r04_de_c4_k1res_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
	NRM_K1=res timeout 240 rocprofv3 --pmc $c --output-format csv -d $O/de_c4_res_$c -o pmc -- $B --workload de_c4 --steps 2 --warmup 1 > /dev/null 2> $O/res_$c.err
done
python3 tools/pmc_summary.py $O/de_c4_res_FETCH_SIZE $O/de_c4_res_WRITE_SIZE > $O/r04_pmc_de_c4_k1res.json
NRM_K1=res timeout 120 python3 tools/k1_phases.py 16000 50000 f32 5 > $O/r04_k1res_phases.txt 2>&1
NRM_K1=res timeout 120 python3 tools/k1_phases.py 3840 500000 f64 3 >> $O/r04_k1res_phases.txt 2>&1
NRM_K1=res NRM_K1_ABLATE=1 timeout 120 python3 tools/k1_phases.py 16000 50000 f32 5 >> $O/r04_k1res_phases.txt 2>&1
timeout 120 python3 tools/time_binnet.py > $O/r04_binnet_time.txt 2>&1
timeout 60 ./tools/mall_probe > $O/r04_mall_probe.txt 2>&1
timeout 300 python3 tools/time_chunks.py 3840 500000 1 8 16 > $O/r04_k2_chunks_c5.txt 2>&1
cat $O/r04_k1res_phases.txt $O/r04_binnet_time.txt $O/r04_mall_probe.txt | grep -v amdgpu
