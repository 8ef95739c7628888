E=mg-cfd-app-plain_amd/csrc/build/exp
for e in 1 0 1 0; do MGCFD_LIB=$E/libmgcfd_hip_nofor.so MGCFD_FREE_SECOND_TILE=$e timeout -k 10 120 python3 tools/exp/time_flux.py 67 500 free 2>&1 | grep median | sed "s/^/second_tile=$e /"; done
