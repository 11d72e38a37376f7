#!/bin/bash
# round 6, closing set under the final source hash: GPU suite + smoke, traffic / kernel stats (profile_round), 15x15 traffic, the two-product
# kernels' counters (MFMA pipe busy, clock, bytes), forward times by batch size with two and three products, and the bench record
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r6z_pytest.log 2>&1; tail -3 gpurun_out/r6z_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python tools/profile_round.py r6z > gpurun_out/r6z_profile_round.log 2>&1; tail -1 gpurun_out/r6z_profile_round.log
rm -rf gpurun_out/profiles_r6z/raw_*
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --board 15 --games 1024 --blocks 10 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe --no-trained-net --no-fp16-grid"
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c -d /tmp/r6z15_$c -o p -- $B --sims 20 --steps 1 --warmup 0 > /dev/null 2>&1
python $R/tools/rocpd_summary.py pmc $(find /tmp/r6z15_$c -name "*.db" | head -1) 2>&1 | grep -i "boardh\|layer16h" | head -4
done > $R/gpurun_out/r6z_pmc_15x15.txt 2>&1; cat $R/gpurun_out/r6z_pmc_15x15.txt
# the two-product kernels: trained checkpoint on the fp16 grid, 4096 boards (resident trunk) / random-init 15x15, 1024 boards (board-resident)
{
for spec in "4096 4 9 0 --weights $R/profiles/r4_trained_9x9_4block.pt" "4096 4 9 0 --weights $R/profiles/r4_trained_9x9_4block.pt --fp16-grid" "1024 10 15 0" "1024 10 15 0 --fp16-grid"; do
  echo "### tools/time_net.py $spec"
  python $R/tools/time_net.py $spec 2>/dev/null
  for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
    rm -rf /tmp/r6zw; rocprofv3 --kernel-trace --pmc $set -d /tmp/r6zw -o p -- python $R/tools/time_net.py $spec > /dev/null 2>&1
    python $R/tools/rocpd_summary.py pmc $(find /tmp/r6zw -name "*.db" | head -1) 2>&1 | grep -i "k_trunk16h\|k_boardh" | head -4
  done
done
} > $R/gpurun_out/r6z_two_product_kernels_pmc.txt 2>&1
cd $R
{
echo "# forward time by batch size, 9x9 / 4 blocks, random-init weights as they are (3 products) and rounded to the fp16 grid (2 products); tools/time_net.py"
for n in 64 128 256 512 768 1024 2048 3072 4096; do
  python tools/time_net.py $n 4 9 0 2>/dev/null
  python tools/time_net.py $n 4 9 0 --fp16-grid 2>/dev/null
done
echo "# 15x15 / 10 blocks"
for n in 32 64 256 1024; do
  python tools/time_net.py $n 10 15 0 2>/dev/null
  python tools/time_net.py $n 10 15 0 --fp16-grid 2>/dev/null
done
echo "# searches of a few games (tools/time_single_game.py: tree step + per-board / row kernels), three and two products"
for g in 1 8 32 128; do
  python tools/time_single_game.py --games $g --moves 6 2>/dev/null
  python tools/time_single_game.py --games $g --moves 6 --fp16-grid 2>/dev/null
done
} > gpurun_out/r6z_forward_by_batch_two_vs_three_products.txt 2>&1
python - <<'P'
import json, re
t = json.load(open('gpurun_out/profiles_r6z/r6z_traffic.json'))
v = {}
for l in open('gpurun_out/r6z_pmc_15x15.txt'):
    m = re.search(r'k_boardh<15, 2>.*?(FETCH_SIZE|WRITE_SIZE)\s+\d+\s+([0-9.e+]+)', l)
    if m: v[m.group(1)] = float(m.group(2))
if len(v) == 2:
    t['other_workloads'] = [{"kernel": "k_boardh<15, 2>", "workload": {"board": 15, "games": 1024, "blocks": 10, "planes": 128, "note": "BASELINE configs[4] per-GPU shape; one launch = conv1 + 20 trunk convs + the heads' 1x1 convs of 1024 boards"},
                             "fetch_size_kib": v['FETCH_SIZE'], "write_size_kib": v['WRITE_SIZE'], "fetch_correction": 2.0, "hbm_bytes_per_launch": (2 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024,
                             "source": "profiles/r6z_pmc_15x15.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, same box and sources as the record above; tools/exp/r6z.sh)"}]
json.dump(t, open('gpurun_out/profiles_r6z/r6z_traffic.json', 'w'), indent=1)
print('traffic json:', t['csrc_sha16'], t['hbm_bytes_per_launch'], [w['hbm_bytes_per_launch'] for w in t.get('other_workloads', [])])
P
cp gpurun_out/profiles_r6z/r6z_traffic.json profiles/r6z_traffic.json
python bench.py > gpurun_out/r6z_bench.json 2> gpurun_out/r6z_bench.err
python -c "
import json
d=json.load(open('gpurun_out/r6z_bench.json')); t=d['trained_net']; g=d['trained_net_fp16grid']
print('bench:', round(d['value']), d['roofline']['frac'], d['roofline']['traffic'], '| trained', round(t['value']), 'static', round(t['static_rows']['value']), '| fp16grid', round(g['value']), g['roofline']['frac'], g.get('vs_trained_net'), '| wide', round(d['wide_board']['value']), d['wide_board']['roofline'].get('traffic'), d['wide_board'].get('fp16grid',{}).get('value'), '| single', d['single_game']['value'], '| cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
"
