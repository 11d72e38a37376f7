# round 5, closing set under the final source hash (after k_order): GPU suite + smoke, traffic / kernel stats (profile_round), 15x15 traffic, bench records
python -m pytest tests -m gpu -x -q > gpurun_out/r5zj_pytest.log 2>&1; tail -3 gpurun_out/r5zj_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python tools/profile_round.py r5zj > gpurun_out/r5zj_profile_round.log 2>&1; tail -1 gpurun_out/r5zj_profile_round.log
rm -rf gpurun_out/profiles_r5zj/raw_*
cd /tmp; export TMPDIR=/tmp
B="python /root/repo/bench.py --board 15 --games 1024 --blocks 10 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe --no-trained-net"
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c -d /tmp/r5zj15_$c -o p -- $B --sims 20 --steps 1 --warmup 0 > /dev/null 2>&1
python /root/repo/tools/rocpd_summary.py pmc $(find /tmp/r5zj15_$c -name "*.db" | head -1) 2>&1 | grep -i "boardh\|layer16h" | head -4
done > /root/repo/gpurun_out/r5zj_pmc_15x15.txt 2>&1; cat /root/repo/gpurun_out/r5zj_pmc_15x15.txt
cd /root/repo
python - <<'P'
import json, re
t = json.load(open('gpurun_out/profiles_r5zj/r5zj_traffic.json'))
v = {}
for l in open('gpurun_out/r5zj_pmc_15x15.txt'):
    m = re.search(r'k_boardh<15, 2>.*?(FETCH_SIZE|WRITE_SIZE)\s+\d+\s+([0-9.e+]+)', l)
    if m: v[m.group(1)] = float(m.group(2))
if len(v) == 2:
    t['other_workloads'] = [{"kernel": "k_boardh<15, 2>", "workload": {"board": 15, "games": 1024, "blocks": 10, "planes": 128, "note": "BASELINE configs[4] per-GPU shape; one launch = conv1 + 20 trunk convs of 1024 boards"},
                             "fetch_size_kib": v['FETCH_SIZE'], "write_size_kib": v['WRITE_SIZE'], "fetch_correction": 2.0, "hbm_bytes_per_launch": (2 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024,
                             "source": "profiles/r5zj_pmc_15x15.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, same box and sources as the record above; tools/exp/r5zj.sh)"}]
json.dump(t, open('gpurun_out/profiles_r5zj/r5zj_traffic.json', 'w'), indent=1)
print('traffic json:', t['csrc_sha16'], t['hbm_bytes_per_launch'], [w['hbm_bytes_per_launch'] for w in t.get('other_workloads', [])])
P
mkdir -p profiles_tmp && cp gpurun_out/profiles_r5zj/r5zj_traffic.json profiles/r5zj_traffic.json
python bench.py > gpurun_out/r5zj_bench.json 2> gpurun_out/r5zj_bench.err
python -c "
import json
d=json.load(open('gpurun_out/r5zj_bench.json')); t=d['trained_net']
print('bench:', round(d['value']), d['roofline']['frac'], d['roofline']['traffic'], '| trained', round(t['value']), 'static', round(t['static_rows']['value']), 'tree us', t['roofline_tree']['avg_launch_ms']*1e3, '| wide', round(d['wide_board']['value']), d['wide_board']['roofline'].get('traffic'), '| single', d['single_game']['value'], '| cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
"
