# round 5: timing knock-outs of k_boardh<15> (wrong results on purpose): what does each part of the launch cost? 1024 boards, 10 blocks
python tools/time_net.py 1024 10 15 0 2>/dev/null | sed 's/^/product       : /'
for ko in 1 2 3 4 5; do
  : # (libomok_hip_bko$ko.so built in the container: AO_BUILD_TAG=bko$ko AO_EXTRA_FLAGS="-DAO_BKO=$ko -DAO_WRONG_RESULTS_OK" python -m alpha_omok_amd.build)
  AO_LIB_TAG=bko$ko python tools/time_net.py 1024 10 15 0 2>/dev/null | sed "s/^/AO_BKO=$ko      : /"
done
python tools/time_net.py 1024 10 15 0 2>/dev/null | sed 's/^/product again : /'
