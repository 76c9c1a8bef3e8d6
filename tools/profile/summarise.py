"""Prints the numbers the docs quote from a prof_r05.sh output directory: python tools/profile/summarise.py gpurun_out/r05_e/"""
import csv, json, sys
O = sys.argv[1].rstrip("/") + "/"
def j(n): return json.load(open(O + n))
b = j('bench.json'); print('bench', round(b['value']), b['ms_per_step'], [round(x, 2) for x in b['config'].get('windows_ms')], 'single', round(b['single_attack']['value']), b['single_attack']['ms_per_step'])
print('roofline', {k: b['roofline'].get(k) for k in ('achieved', 'frac', 'solo_launch_ms', 'solo_frac', 'solo_launch_ms_end', 'solo_drift', 'traffic', 'traffic_stale', 'avg_launch_ms')})
e = b['secondary'].get('end_to_end')
if e: print('e2e static', {k: e['static'].get(k) for k in e['static'] if 'per_s' in k or k == 'wall_s'}, 'dynamic', {k: e['dynamic'].get(k) for k in e['dynamic'] if 'per_s' in k or k == 'wall_s'}, e.get('dynamic_over_static'), min(e['iterations_per_attack']), max(e['iterations_per_attack']))
for k, v in b['secondary'].items():
    if isinstance(v, dict) and 'value' in v: print(k, round(v['value']), round(v['single_attack']['value']), v['single_attack']['ms_per_step'], v['single_attack'].get('kernel_launch_ms'))
d = j('bench_driver_args.json'); print('driver', round(d['value']), round(d['single_attack']['value']), [round(x, 2) for x in d['config']['windows_ms']])
for n in ('bench_faithful', 'bench_gmm_csi', 'bench_realistic', 'iv_bench', 'iv_osi_b201_bench'):
    x = j(n + '.json'); s = x.get('single_attack'); print(n, round(x['value']), round(s['value']), s['ms_per_step'])
for n in ('gmm_1attack', 'gmm_3attacks', 'iv_sv_1attack', 'iv_osi_b201_1attack'):
    print(n)
    for r in list(csv.DictReader(open(O + n + '_kernel_stats.csv')))[:19]:
        if int(r['Calls']) > 10: print('   %-40s %7.1f' % (r['Name'].split('(')[0][-40:], float(r['AverageNs']) / 1e3))
print(b['cpu_baseline'])
t = j('traffic.json'); print({k: t[k] for k in t if k in ('commit',)})
