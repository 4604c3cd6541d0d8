# round 6, call 20: which of the library's streams share a hardware queue?  pool-stream offsets (VBG_STREAM_SKIP), A/B x 2
cd /root/repo
mkdir -p gpurun_out
R=gpurun_out/r6c20
run() { env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg --no-single-stream-pass 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a ${R}_ab.txt; }
rm -f ${R}_ab.txt
for i in 1 2; do run VBG_NOP=1; run VBG_STREAM_SKIP=heads=1; run VBG_STREAM_SKIP=heads=2; run VBG_STREAM_SKIP=heads=3; run VBG_STREAM_SKIP=side=1; run VBG_STREAM_SKIP=side=2; run VBG_STREAM_SKIP=side=3; done
