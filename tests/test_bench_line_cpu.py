"""bench.py's contract line (no GPU): built from a recorded run's records it stays under 4 KB -- the driver keeps the tail of stdout, and
the round-4 line (14 KB) lost its first extra workloads there --, carries the contract's keys, one summary row per extra workload and
the scaling series; HBM traffic is attached only from the kernel(s) a step actually ran."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _recorded():
	d = json.load(open(os.path.join(ROOT, 'profiles', 'r04_bench_default.json')))
	return d, d.pop('extra_workloads')


def test_contract_line_is_short_and_complete():
	head, extras = _recorded()
	head['scaling_series'] = dict(workload='coex_c5: BASELINE configs[4] per-rank slice, 3750 gene rows per rank x 500000 cells fp64', value=9.2e7, unit='tests/s', ms_per_step=76.2,
								  ranks=1, tests_per_step=7029375, value_per_rank=9.2e7, roofline_frac=0.45, exchange_ms_not_hidden=0.0)
	head['config']['scaling_series'] = 'top-level `scaling_series` = BASELINE configs[4] per-rank slice, the same workload at every N (the headline itself at N > 1)'
	extras['de_c4']['cold_ms'] = 3.1
	extras['broken'] = dict(error='RuntimeError: ' + 'x' * 500)
	line = bench.contract_line(head, extras, 1, head.get('end_to_end_pcie'))
	text = json.dumps(line)
	assert len(text) < 4096, len(text)
	for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline',
			  'scaling_series', 'extra_workloads'):
		assert k in line, k
	assert line['vs_baseline'] is None and line['config']['workload'].startswith('norm.coex') and 'model' not in line['config']
	for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
		assert k in line['roofline'], k
	for k in ('value', 'unit', 'cores', 'kind', 'sample'):
		assert k in line['cpu_baseline'], k
	assert set(line['extra_workloads']) == set(extras) and line['extra_workloads']['de_c4']['cold_ms'] == 3.1 and isinstance(line['extra_workloads']['broken'], str)
	assert line['scaling_series']['ranks'] == 1 and line['scaling_series']['workload'].startswith('coex_c5')


def test_traffic_comes_from_the_kernels_that_ran(tmp_path, monkeypatch):
	pm = {'k_de_sparse<float, true, true>': dict(hbm_bytes_per_launch=3.4e9, effective_clock_ghz=2.1), 'k_s1_stream<float, 5, 8, true, true>': dict(hbm_bytes_per_launch=3.2e9),
		  'k_gram_i8<6>': dict(hbm_bytes_per_launch=1.96e10)}
	f = tmp_path / 'pmc.json'
	f.write_text(json.dumps(pm))
	monkeypatch.setitem(bench.PMC_FILES, 'w', [str(f)])
	roof = dict(kernel='k_de_sparse')
	bench.pmc_traffic('w', roof, kernels=['k_de_sparse', 'k_s1_stream'])
	assert roof['traffic'] == 3.4e9 + 3.2e9 and 'k_de_sparse' in roof['traffic_source'] and 'k_s1_stream' in roof['traffic_source'] and 'k_gram_i8' not in roof['traffic_source']
	roof = dict(kernel='k_de_sparse')
	bench.pmc_traffic('w', roof)
	assert roof['traffic'] == 3.4e9
	roof = dict(kernel='k_fused_new', traffic=1.0, traffic_source='stale')
	bench.pmc_traffic('w', roof)  # no counters for this kernel on file: null, never another kernel's
	assert roof['traffic'] is None and 'traffic_source' not in roof
