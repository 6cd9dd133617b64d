import json, shutil, os, csv
g='gpurun_out/'
m={'final_bench_grid8.json':'r06_bench_grid8.json','final_bench_grid8_single_lane.json':'r06_bench_grid8_single_lane.json',
'profiles_new/kernel_trace_summary.csv':'r06_bench_grid8_kernel_trace_summary.csv','profiles_new/kernel_stats.csv':'r06_bench_grid8_rocprofv3_kernel_stats.csv',
'profiles_new/bench_under_rocprof.json':'r06_bench_grid8_under_rocprof.json','profiles_new/pmc_counters_summary.csv':'r06_pmc_counters_summary.csv',
'profiles_new/hbm_traffic_and_mfma_util.json':'r06_hbm_traffic_and_mfma_util.json','profiles_new/batch1_hbm_traffic.json':'r06_batch1_hbm_traffic.json',
'final_bench_cascade.json':'r06_bench_cascade.json','final_bench_cascade_sync.json':'r06_bench_cascade_synchronous.json','final_bench_cascade_fp16.json':'r06_bench_cascade_fp16.json',
'final_bench_tiles.json':'r06_bench_tiles.json','final_bench_grid8_fp16.json':'r06_bench_grid8_fp16.json','final_bench_grid8_fp32.json':'r06_bench_grid8_fp32.json',
'final_bench_grid32_n1.json':'r06_bench_grid32_n1.json','final_per_op_batch64.txt':'r06_per_op_batch64.txt','final_per_op_batch1.txt':'r06_per_op_batch1.txt',
'final_ttft_ttst.json':'r06_ttft_ttst_latency.json','b1_timeline.txt':'r06_batch1_timeline.txt','final_sb_layers.txt':'r06_sb_layer_battery.txt',
'final_tests.txt':'r06_gpu_tests.txt','batch_sweep.txt':'r06_batch_sweep.txt','final_decoder_forward_batch4.txt':'r06_decoder_forward_batch4.txt','attn_profile.txt':'r06_attention_mfma_utilisation.txt'}
for a,b in m.items():
    if os.path.exists(g+a) and os.path.getsize(g+a)>0: shutil.copy(g+a,'profiles/'+b)
    else: print('MISSING',a)
open('profiles/r06_wide_tile_and_two_lanes_ab.txt','w').write('''Round 6, final build: the two structural changes of the round against each other on ONE box, interleaved, two rounds (tools/r06_final.sh -> tools/ab.sh bench):
`python bench.py` (BASELINE configs[2]) under engine options; then the cascade (configs[4] shapes) without the wide tile / without the few-cout flavour / with the defaults.
  glds_wide=0,dual_stream=0 = the round-5 configuration (one sampler lane, conv_glds / conv_sb only)
  glds_wide=0               = two lanes only          dual_stream=0 = wide tile only          (empty) = the defaults          glds_wide_min_wgs=1024 = wide tile without the 16x16 level

'''+open(g+'final_ab_wide_dual.txt').read()+'\ncascade:\n'+open(g+'final_ab_cascade.txt').read())
d=json.loads(open('profiles/r06_bench_grid8.json').read().strip().splitlines()[-1]); r=d['roofline']
print('grid8', d['value'], d['ms_per_step'], 'frac', r['achieved'], r['frac'], r['avg_launch_us'], 'single', r['single_lane'], 'e2e', r['end_to_end_achieved'], r['end_to_end_frac'], 'traffic', r.get('traffic'), r.get('traffic_over_algorithmic'), r.get('traffic_over_algorithmic_strict'), r['library_build_id'], 'share', r['share_of_unet_kernel_time'])
print('lat', d['latency_single_tile_ms'], 'anchor', d['strong_scaling_anchor']['value'], d['strong_scaling_anchor']['ms_per_64_window_batch'], 'cpu', d['cpu_baseline']['value'], d['roofline']['small_batch_kernel'])
pj=json.load(open('profiles/r06_hbm_traffic_and_mfma_util.json')); print(pj['csrc_sha16'], pj['library_build_id'], pj.get('seam_sha16'))
tot=0;n=0
for k,v in pj['kernels'].items():
    print(k[:80], v.get('dispatches'), v.get('mfma_util'), v.get('l2_hit_rate'), v.get('hbm_read_bytes_per_launch'), v.get('hbm_write_bytes_per_launch'), v.get('fabric_read_requests_per_launch'), v.get('wait_any_share'), v.get('valu_per_mfma'))
    if 'conv_glds' in k and v.get('mfma_util'): tot+=v['dispatches']*v['mfma_util']; n+=v['dispatches']
print('family mfma busy weighted', tot/n, n)
rows=[x for x in csv.DictReader(open('profiles/r06_bench_grid8_kernel_trace_summary.csv')) if 'conv_glds_kernel' in x['kernel']]
calls=sum(int(x['calls']) for x in rows); t=sum(float(x['total_us']) for x in rows)
print('trace', calls, t, t/calls, 199.127893693e9/(t/calls*1e-6)/1e12)
for f in ['r06_bench_cascade.json','r06_bench_cascade_fp16.json','r06_bench_cascade_synchronous.json','r06_bench_tiles.json','r06_bench_grid8_fp16.json','r06_bench_grid8_fp32.json','r06_bench_grid32_n1.json','r06_bench_grid8_single_lane.json']:
    d=json.loads(open('profiles/'+f).read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print(f, d['value'], d['ms_per_step'], {k:r[k] for k in r if k in ('achieved','frac','end_to_end_frac','hbm_gbps_decoder_512x512','one_request_conv_kernel_ms','avg_launch_us','one_request_kernel_ms_by_resolution')})
print(open('profiles/r06_ttft_ttst_latency.json').read()[:120])
print(open('profiles/r06_batch1_hbm_traffic.json').read()[300:])
print(open('profiles/r06_batch1_timeline.txt').read()[:200])
print(open('profiles/r06_per_op_batch64.txt').readline())
print(open('profiles/r06_batch_sweep.txt').read()[-900:])
print(open('profiles/r06_wide_tile_and_two_lanes_ab.txt').read()[-700:])
print(open('profiles/r06_gpu_tests.txt').read()[-200:])
