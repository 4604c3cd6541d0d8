# round 5, call 5: lr = 0 probe of the stock DDP route (do the gradients of steps 2, 3 equal step 1's?), BatchNorm folding kernels
cd /root/repo
for h in 1 0; do echo "VBG_HOME=$h"; VBG_HOME=$h timeout 300 python tools/ddp_stock_probe.py 2>&1 | grep "^step" | cut -c1-900; done
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -rf -x -k "batchnorm" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -rf -x 2>&1 | tail -5
bash tools/run_ab.sh VBG_BN_FOLD 2>&1 | grep "VBG_BN_FOLD="
