export BORE_LIB_PATH=$PWD/bore_amd/csrc/libbore_hip_w12.so
for w in 0 1; do BORE_LBFGSB_W12=$w timeout -k 10 200 python tools/cfg_restarts.py cfg2 256 5 2>&1 | grep -v amdgpu.ids | sed "s/^/W12=$w /"; done
for w in 0 1; do BORE_LBFGSB_W12=$w timeout -k 10 200 python tools/cfg_restarts.py cfg2 64 5 2>&1 | grep -v amdgpu.ids | sed "s/^/W12=$w /"; done
