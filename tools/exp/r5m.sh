# round 5 checkpoint: full GPU suite + smoke, the profile set of tools/profile_round.py (kernel stats, PMC, traffic keyed on the source hash)
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r5m_gputests.log; tail -3 gpurun_out/r5m_gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python tools/profile_round.py r5m > /dev/null 2>&1
ls gpurun_out/profiles_r5m/
for f in kernel_stats pmc traffic single_game_kernel_stats; do ls gpurun_out/profiles_r5m/r5m_$f.* >/dev/null 2>&1 && cp gpurun_out/profiles_r5m/r5m_$f.* gpurun_out/; done
rm -rf gpurun_out/profiles_r5m/raw_*
cat gpurun_out/r5m_traffic.json | head -40
