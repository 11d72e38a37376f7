# round 5: where does k_boardh<15>'s time go? PMC passes (MFMA busy, LDS, waits) on the configs[4] per-GPU shape, beside k_layer16h<15> (AO_BOARDK=0)
python tools/profile_deep.py r5j --filter k_boardh,k_layer16h --bench-args "--board 15 --games 1024 --blocks 10 --no-trained-net" > /dev/null
cp gpurun_out/profiles_r5j/r5j_pmc_deep.txt gpurun_out/r5j_pmc_deep_boardh.txt
AO_BOARDK=0 python tools/profile_deep.py r5j0 --filter k_boardh,k_layer16h --bench-args "--board 15 --games 1024 --blocks 10 --no-trained-net" > /dev/null
cp gpurun_out/profiles_r5j0/r5j0_pmc_deep.txt gpurun_out/r5j_pmc_deep_layer16h.txt
rm -rf gpurun_out/profiles_r5j gpurun_out/profiles_r5j0
cat gpurun_out/r5j_pmc_deep_boardh.txt | grep -v "^#"
grep "k_layer16h<15, 4, 4, 0>" gpurun_out/r5j_pmc_deep_layer16h.txt
