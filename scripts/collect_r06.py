"""gpurun_out/r06/* (written by scripts/final_run_r06.sh on the GPU box) -> profiles/r06_*; prints the numbers DESIGN quotes"""
import json, os, shutil
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "r06")
dst = os.path.join(src, "..", "..", "profiles")
names = {'bench_50m_n1.json': 'r06_bench_50m_n1.json', 'bench_50m_n1_genome.json': 'r06_bench_50m_n1_genome.json',
         'bench_50m_n1_genome_1x.json': 'r06_bench_50m_n1_genome_1x.json', 'bench_50m_n1_genome_3x.json': 'r06_bench_50m_n1_genome_3x.json',
         'bench_50m_n1_k21.json': 'r06_bench_50m_n1_k21.json', 'r06_kernel_stats.txt': 'r06_kernel_stats.txt', 'r06_pmc.json': 'r06_pmc.json',
         'r06_k21_kernel_stats.txt': 'r06_k21_kernel_stats.txt', 'r06_k21_pmc.json': 'r06_k21_pmc.json',
         'r06_sq_counters.json': 'r06_sq_counters.json', 'r06_sq_counters_l1_ring.json': 'r06_sq_counters_l1_ring.json',
         'configs.json': 'r06_configs.json', 'reference_loop.json': 'r06_reference_loop.json',
         'virtual8_genome.json': 'r06_virtual_ranks8_genome.json', 'gpu_tests.txt': 'r06_gpu_tests.txt',
         'coverage_scan.txt': 'r06_coverage_scan.txt'}
for a, b in names.items():
    assert os.path.getsize(os.path.join(src, a)) > 0, a
    shutil.copy(os.path.join(src, a), os.path.join(dst, b))
for f in ('bench_50m_n1', 'bench_50m_n1_genome', 'bench_50m_n1_genome_1x', 'bench_50m_n1_genome_3x', 'bench_50m_n1_k21'):
    d = json.load(open(os.path.join(src, f + '.json')))
    print(f, d['value'], d['ms_per_step'], d['parity_fullsize'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline'].get('traffic_source'))
d = json.load(open(os.path.join(src, 'bench_50m_n1.json')))
e = d['extra']['config5_kmer_index']
print('config5', e['build_ms'], e['parity'], '1e9 pairs', e['synthetic_1e9_pairs']['build_ms'], e['synthetic_1e9_pairs']['parity'])
print('host_fed', d.get('host_fed', {}).get('value'), 'cpu', d['cpu_baseline']['value'])
print('source hash', json.load(open(os.path.join(src, 'r06_pmc.json'))).get('_source_hash'))
print(open(os.path.join(src, 'coverage_scan.txt')).read())
