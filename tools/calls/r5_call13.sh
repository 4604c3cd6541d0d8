# round 5, call 13: host profile of single-document inference; inference latency
cd /root/repo
python tools/infer_host_profile.py 2>/dev/null | head -75 | cut -c1-200
python tools/infer_latency.py 2>/dev/null
