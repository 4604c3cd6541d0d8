# round 6, call 9: conv3 pipelined kernel -- activation pieces to LDS through inline-asm ds_write (no compiler vmcnt(0) in front of them) and an
# 8-stage filter ring on 64-filter tiles; four builds (build/ab/libvbg_{base,asmw,nsb8,both}.so): shapes, bit-identity, the step (A/B x 2)
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c09
for v in base asmw nsb8 both; do
  echo "== $v"; VBG_BENCH_HASH=1 VBG_LIB_PATH=$PWD/build/ab/libvbg_$v.so timeout 600 python tools/conv3_pw_bench.py 2>&1 | grep "PW  \|PW bn64\|hash" | grep -v forced
done > ${R}_shapes.txt
run() { VBG_LIB_PATH=$PWD/build/ab/libvbg_$1.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['last_loss'] if 'last_loss' in d else d['config']['last_loss'])" | tee -a ${R}_ab.txt; }
rm -f ${R}_ab.txt
for i in 1 2; do for v in base asmw nsb8 both; do run $v; done; done
grep "hash" ${R}_shapes.txt | sort | uniq -c | awk '{print $1}' | sort | uniq -c
grep -v hash ${R}_shapes.txt | cut -c1-120
