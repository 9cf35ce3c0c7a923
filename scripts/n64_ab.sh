# Round-4 GPU call: 64-channel tiles of the tap-reuse kernel, four waves along the pixels vs the 4 x 2 form (profiles/r04za_n64_*.txt).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine.py tests/test_hovernet_post.py -m gpu -q -x 2>&1 | tail -4
SH="1024,64,64,64 1024,64,64,56 32,64,64,256 8,64,64,512 32,256,64,164 1024,128,128,32"
echo "== 4 waves along M"; timeout 300 python scripts/conv_probe.py $SH 2>&1 | grep "^n=" | tee gpurun_out/r04za_n64_4waves.txt
echo "== 8 waves (4 x 2)"; TIA_DEV=1 TIA_CONV_N64_8WAVES=1 timeout 300 python scripts/conv_probe.py $SH 2>&1 | grep "^n=" | tee gpurun_out/r04za_n64_8waves.txt
timeout 300 python scripts/perf_trunk.py 1024 256 2>&1 | grep -v "amdgpu\|No local"
timeout 300 python scripts/perf_trunk.py 1024 224 2>&1 | grep -v "amdgpu\|No local"
