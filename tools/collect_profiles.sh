# every rocprofv3 collection the committed profiles/ are made from, on one box:  gpurun -- bash tools/collect_profiles.sh ; python tools/make_profiles.py r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-amp-leg --no-h2d-leg"
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/bench_line.err | grep "^{" > gpurun_out/bench_line.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_e -o e -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-amp-leg --no-h2d-leg > gpurun_out/prof_e.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_amp -o amp -- python bench.py --amp --steps 10 --warmup 3 --no-cpu-baseline --no-h2d-leg > gpurun_out/prof_amp.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- $B > gpurun_out/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- $B > gpurun_out/pmc_w.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_m -o m -- $B > gpurun_out/pmc_m.log 2>&1
timeout 600 python tools/gemm_bench.py > gpurun_out/gemm_shapes.txt 2>&1
timeout 600 python tools/gemm_bench.py --amp > gpurun_out/gemm_shapes_amp.txt 2>&1
timeout 600 python tools/plane_gemm_bench.py > gpurun_out/plane_gemm_shapes.txt 2>&1
timeout 600 python tools/conv3_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/conv3_shapes.txt
timeout 600 python tools/conv3_forms_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/conv3_forms.txt
timeout 600 python tools/conv3_pw_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/conv3_pw.txt
timeout 600 python tools/step_conv3_profile.py 2>/dev/null > gpurun_out/step_conv3.txt
timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/attn_shapes.txt
timeout 300 python tools/step_plane_profile.py 2>/dev/null > gpurun_out/step_plane.txt
timeout 300 python tools/plane_pair_bench.py 30 2>&1 | grep -v amdgpu.ids > gpurun_out/plane_pair_shapes.txt
timeout 600 python bench.py --amp --steps 20 --warmup 5 --no-cpu-baseline --no-h2d-leg 2>/dev/null | grep "^{" > gpurun_out/bench_amp_line.json
rm -f gpurun_out/prof_amp/amp_kernel_trace.csv
ls -la gpurun_out/prof_e gpurun_out/pmc_f gpurun_out/pmc_m | head -30; cut -c1-400 gpurun_out/bench_line.json
# exploratory shapes and legs of the same build
for s in cfg3 cfg4 cfg5; do timeout 600 python bench.py --shape $s --steps 6 --warmup 2 --no-cpu-baseline --no-h2d-leg 2>/dev/null | grep "^{" > gpurun_out/bench_$s.json; done
timeout 300 python tools/infer_latency.py > gpurun_out/infer_latency.txt 2>&1
