E=mg-cfd-app-plain_amd/csrc/build/exp
for v in cur plainst ord5 cur; do MGCFD_LIB=$E/libmgcfd_hip_$v.so timeout -k 10 200 python3 tools/exp/time_flux.py 134 100 exact,free 2>&1 | grep median; done
