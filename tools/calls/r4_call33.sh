cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "deep_ktiles" 2>&1 | tail -5
for v in 0 1; do echo "== BK64=$v"; VBG_PAIR_BK64=$v python tools/step_plane_profile.py 2>/dev/null; done
