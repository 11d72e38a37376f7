# round 5 (review item 5b): where do the non-MFMA cycles of the medium- and small-batch conv kernels go? one rocprofv3 PMC pass per counter group on the
# bare forward (tools/time_net.py): k_layer16hk<9, 4> at 1024 boards, k_row16hk<9> at 256 boards, k_layer16h<9> at 2048 boards for comparison
python tools/profile_deep.py r5n_1024 --filter k_layer16hk,k_row16hk,k_layer16h --cmd "python /root/repo/tools/time_net.py 1024 4 9 0" > /dev/null 2>&1
python tools/profile_deep.py r5n_256 --filter k_layer16hk,k_row16hk,k_layer16h --cmd "python /root/repo/tools/time_net.py 256 4 9 0" > /dev/null 2>&1
python tools/profile_deep.py r5n_2048 --filter k_layer16hk,k_row16hk,k_layer16h --cmd "python /root/repo/tools/time_net.py 2048 4 9 0" > /dev/null 2>&1
for t in 1024 256 2048; do cp gpurun_out/profiles_r5n_$t/r5n_${t}_pmc_deep.txt gpurun_out/; rm -rf gpurun_out/profiles_r5n_$t; done
grep -h "0>(Lay\|hk<9" gpurun_out/r5n_1024_pmc_deep.txt | grep -v ", 2>(\|, 1>(" | head -40
