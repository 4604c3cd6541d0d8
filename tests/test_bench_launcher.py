"""`python bench.py --gpus N` must START without a launcher around it (the driver's command form is exactly that):
bench.py re-executes itself under torch.distributed.run the way the reference is started (readme.md:87, pipeline/distributed_utils.py:
74-98).  CPU-only: the hidden --launch-check mode exercises the launcher, the rendezvous on 127.0.0.1 and one all-reduce over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    e.update({"VBG_DIST_BACKEND": "gloo", "GLOO_SOCKET_IFNAME": "lo"}, **(env or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"] + extra, capture_output=True, text=True,
                       timeout=240, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE JSON line: only rank 0 writes to stdout
    return json.loads(lines[0]), r.stderr


@pytest.mark.timeout(300)
def test_bench_gpus2_launches_itself_under_torchrun():
    out, err = _run([])
    assert out["launch_check"] and out["self_launched"]
    assert out["world_size"] == 2 == out["gpus_arg"]
    assert out["rank_sum"] == 3.0                     # ranks 0 and 1 both took part in the all-reduce
    assert out["syncbn_comm"] == "shared"             # one communicator is the default
    assert "torch.distributed.run" in err


@pytest.mark.timeout(400)
def test_two_communicator_probe_falls_back_to_shared():
    # the opt-in second communicator is probed first; a probe that ends like a stalled one (exit code 17) must turn the timed run to `shared`
    out, err = _run(["--syncbn-comm", "direct"], env={"VBG_PROBE_IN_CHECK": "1", "VBG_PROBE_FAIL": "1"})
    assert out["syncbn_comm"] == "shared" and out["comm_fallback"]
    assert "falling back to --syncbn-comm shared" in err
    # ... and a probe that passes leaves the choice alone
    out, _ = _run(["--syncbn-comm", "direct"], env={"VBG_PROBE_IN_CHECK": "1"})
    assert out["syncbn_comm"] == "direct" and not out["comm_fallback"]
