# round 6, call 27: three workgroups per CU for the 64-filter pipelined convolution kernel (__launch_bounds__(256, 3): 140 VGPRs, no spills)
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c27
for v in base lb3; do
  echo "== $v"; L=""; [ $v = lb3 ] && L=$PWD/build/ab/libvbg_lb3.so
  VBG_BENCH_HASH=1 VBG_LIB_PATH=$L timeout 600 python tools/conv3_pw_bench.py 2>&1 | grep "PW  \|PW bn64\|hash" | grep -v forced | grep "64->64\|32x32\|16x16\|hash B8 128x128 64"
done > ${R}_shapes.txt
run() { VBG_LIB_PATH=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a ${R}_ab.txt; }
rm -f ${R}_ab.txt
for i in 1 2 3; do run base ""; run lb3 $PWD/build/ab/libvbg_lb3.so; done
cat ${R}_shapes.txt | cut -c1-150
