# Round-4 GPU call: strided 3x3 / 7x7-map layers on the gathering LDS-DMA ring (profiles/r04z_ring_probe.txt), trunk and segmentation forwards.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_engine.py -m gpu -q -x -k "mfma_conv or mfma_resnet" 2>&1 | tail -5
SH="1024,64,128,56,3,2 1024,128,256,28,3,2 1024,256,512,14,3,2 1024,512,512,7,3,1 1024,64,128,64,3,2 1024,128,256,32,3,2 1024,256,512,16,3,2 32,512,512,64,3,2"
echo "== gather ring"; timeout 300 python scripts/conv_probe.py $SH 2>&1 | grep "^n=" | tee gpurun_out/r04z_ring_probe.txt
timeout 300 python scripts/perf_trunk.py 1024 256 2>&1 | grep -v "amdgpu\|No local"
timeout 300 python scripts/perf_trunk.py 1024 224 2>&1 | grep -v "amdgpu\|No local"
timeout 300 python scripts/perf_hovernet_layers.py hovernet 32 2>&1 | grep -v "amdgpu\|No local" | head -1
timeout 300 python scripts/perf_hovernet_layers.py unet 8 2>&1 | grep -v "amdgpu\|No local" | head -1
