#!/bin/bash
# Round-4 profiles (run on the GPU box from the repo root): per-kernel stats and HBM-side counters of the bench workloads, including the
# ones added this round (single=4 at configs[3] size, binnet on a 30 000^2 P-value matrix, configs[1] on the fp64 matrix cores).
# Counters are collected in their own passes (no trace options besides --kernel-trace), as gpurun requires.
export TMPDIR=/tmp
O=gpurun_out/r04prof
mkdir -p $O
B="python3 bench.py --cpu-seconds 0 --e2e 0 --no-extras"
declare -A W=( [c2]="--steps 20 --warmup 3" [de_c3]="--workload de_c3 --steps 20 --warmup 3" [de_c4]="--workload de_c4 --steps 5 --warmup 2" [coex_c5]="--workload coex_c5 --steps 3 --warmup 1" \
	[de_c4_single4]="--workload de_c4_single4 --steps 5 --warmup 2" [binnet_c5]="--workload binnet_c5 --steps 5 --warmup 2" [c2_f64]="--workload coex_c2_f64 --steps 10 --warmup 3" )
for w in c2 de_c3 de_c4 coex_c5 de_c4_single4 binnet_c5 c2_f64; do
	rocprofv3 --kernel-trace --stats --output-format csv -d $O/${w}_stats -o $w -- $B ${W[$w]} > $O/${w}_stats.json 2> $O/${w}_stats.err
	f=$(find $O/${w}_stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/r04_${w}_kernel_stats.csv
done
for w in c2 de_c4 coex_c5 de_c4_single4 binnet_c5 de_c3; do
	for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
		rocprofv3 --pmc $c --output-format csv -d $O/${w}_$c -o pmc -- $B ${W[$w]} --steps 3 --warmup 1 > /dev/null 2> $O/${w}_$c.err
	done
	python3 tools/pmc_summary.py $O/${w}_FETCH_SIZE $O/${w}_WRITE_SIZE $O/${w}_GRBM_GUI_ACTIVE > $O/r04_pmc_$w.json
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/c2_SQ -o pmc -- $B --steps 3 --warmup 1 > /dev/null 2> $O/c2_SQ.err
python3 tools/pmc_summary.py $O/c2_FETCH_SIZE $O/c2_WRITE_SIZE $O/c2_GRBM_GUI_ACTIVE $O/c2_SQ > $O/r04_pmc_c2.json
# K1 with the rows resident on chip (opt-in): its kernel time and traffic on the configs[3] rows beside the default kernel's
NRM_K1=res rocprofv3 --kernel-trace --stats --output-format csv -d $O/de_c4_res_stats -o res -- $B --workload de_c4 --steps 5 --warmup 2 > /dev/null 2> $O/res_stats.err
f=$(find $O/de_c4_res_stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/r04_de_c4_k1res_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
	NRM_K1=res rocprofv3 --pmc $c --output-format csv -d $O/de_c4_res_$c -o pmc -- $B --workload de_c4 --steps 3 --warmup 1 > /dev/null 2> $O/res_$c.err
done
python3 tools/pmc_summary.py $O/de_c4_res_FETCH_SIZE $O/de_c4_res_WRITE_SIZE > $O/r04_pmc_de_c4_k1res.json
python3 tools/k1_phases.py 16000 50000 f32 5 > $O/r04_k1res_phases.txt 2>&1
NRM_K1=res python3 tools/k1_phases.py 3840 500000 f64 3 >> $O/r04_k1res_phases.txt 2>&1
python3 tools/time_binnet.py > $O/r04_binnet_time.txt 2>&1
./tools/mall_probe > $O/r04_mall_probe.txt 2>&1
ls $O/*.json $O/*.csv $O/*.txt
