cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/r6c55_infer.txt
for rep in 1 2; do
for cfg in "0 0" "1 0" "1 99999999"; do
  set -- $cfg
  echo "VBG_OVERLAP_NOGRAD=$1 VBG_ENCODER_FIRST_PIXELS=$2" | tee -a gpurun_out/r6c55_infer.txt
  VBG_OVERLAP_NOGRAD=$1 VBG_ENCODER_FIRST_PIXELS=$2 VBG_INFER_BATCHES=1,2,8 timeout 300 python tools/infer_latency.py 2>/dev/null | tee -a gpurun_out/r6c55_infer.txt
done
done
