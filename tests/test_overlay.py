"""The drop-in recipe of INTEGRATION.md section 1: `vibertgrid-pytorch_amd/` on PYTHONPATH beside the reference checkout.

The reference's entry points import `model.ViBERTgrid_net` next to `pipeline.train_val_utils` / `pipeline.distributed_utils`
(train_SROIE.py:13-26): `model.*`, `pipeline.transform` and `pipeline.custom_loss` must resolve to THIS package, everything
else under `pipeline.` / `model.` to the reference's directories (which have no `__init__.py`).  The test builds a stub
"reference" tree with the same shape (marker modules, no reference source), runs an entry-point stand-in the way the reference is
launched (`python train_stub.py` from the checkout: the script's directory is sys.path[0], i.e. IN FRONT of PYTHONPATH) and with the
two possible PYTHONPATH orders, and asserts which file every import came from."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vibertgrid-pytorch_amd")

ENTRY = textwrap.dedent("""
    import json, sys
    import torch                                             # the entry points import torch before the model (train_SROIE.py:8)
    from model.ViBERTgrid_net import ViBERTgridNet          # train_SROIE.py:13
    from pipeline.train_val_utils import train_one_epoch     # train_SROIE.py:14-20
    from pipeline.distributed_utils import setup_seed        # train_SROIE.py:21-26
    import pipeline.transform, pipeline.custom_loss, pipeline.criteria
    import model.BERTgrid_generator, model.ResNetFPN_ViBERTgrid, model.grid_roi_align, model.crf
    import model.field_type_classification_head, model.semantic_segmentation_head, model.ref_only_helper
    import model, pipeline
    mods = ["model.ViBERTgrid_net", "model.BERTgrid_generator", "model.ResNetFPN_ViBERTgrid", "model.grid_roi_align", "model.crf",
            "model.field_type_classification_head", "model.semantic_segmentation_head", "model.ref_only_helper",
            "pipeline.transform", "pipeline.custom_loss", "pipeline.train_val_utils", "pipeline.distributed_utils", "pipeline.criteria"]
    out = {m: sys.modules[m].__file__ for m in mods}
    out["train_one_epoch"] = train_one_epoch()               # the reference-side function really is the reference's
    out["net_module_file"] = sys.modules[ViBERTgridNet.__module__].__file__
    print("RESOLVED " + json.dumps(out))
""")


def _stub_reference(top):
    """a tree shaped like the reference checkout: `pipeline/` and `model/` WITHOUT `__init__.py`, every module a marker"""
    for d in ("pipeline", "model", "data", "utils"):
        os.makedirs(os.path.join(top, d))
    marker = "WHO = 'stub-reference'\n"
    for f in ("train_val_utils", "distributed_utils", "criteria", "transform", "custom_loss"):
        body = marker
        if f == "train_val_utils":           # imports its neighbours the way the reference does (pipeline/train_val_utils.py:13-18)
            body += "from pipeline.distributed_utils import setup_seed\nfrom pipeline.criteria import WHO as _c\n" \
                    "def train_one_epoch():\n    return 'stub-reference train_one_epoch'\n"
        if f == "distributed_utils":
            body += "def setup_seed(seed=0):\n    return seed\n"
        open(os.path.join(top, "pipeline", f + ".py"), "w").write(body)
    for f in ("ViBERTgrid_net", "BERTgrid_generator", "ResNetFPN_ViBERTgrid", "grid_roi_align", "crf",
              "field_type_classification_head", "semantic_segmentation_head", "ref_only_helper"):
        body = marker + ("class ViBERTgridNet:\n    pass\n" if f == "ViBERTgrid_net" else "")
        open(os.path.join(top, "model", f + ".py"), "w").write(body)
    open(os.path.join(top, "train_stub.py"), "w").write(ENTRY)
    return top


def _run(cmd, cwd, pythonpath):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(pythonpath))
    r = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESOLVED ")]
    assert line, r.stdout[-2000:]
    return json.loads(line[-1][len("RESOLVED "):])


OURS = ["model.ViBERTgrid_net", "model.BERTgrid_generator", "model.ResNetFPN_ViBERTgrid", "model.grid_roi_align", "model.crf",
        "model.field_type_classification_head", "model.semantic_segmentation_head", "pipeline.transform", "pipeline.custom_loss"]
THEIRS = ["model.ref_only_helper", "pipeline.train_val_utils", "pipeline.distributed_utils", "pipeline.criteria"]


@pytest.mark.parametrize("launch", ["script_in_checkout", "overlay_first", "overlay_last"])
def test_overlay_import_resolution(tmp_path, launch):
    if not os.path.exists(os.path.join(PKG, "libvbg.so")):
        pytest.skip("libvbg.so not built (model.* imports load it)")
    ref = _stub_reference(str(tmp_path / "ViBERTgrid-PyTorch"))
    if launch == "script_in_checkout":       # `python train_SROIE.py` from the checkout, PYTHONPATH as INTEGRATION.md says
        res = _run([sys.executable, "train_stub.py"], ref, [PKG, ref])
    elif launch == "overlay_first":
        res = _run([sys.executable, "-c", ENTRY], str(tmp_path), [PKG, ref])
    else:                                    # even with the reference in front: the regular package wins over the namespace portion
        res = _run([sys.executable, "-c", ENTRY], str(tmp_path), [ref, PKG])
    for m in OURS:
        assert os.path.realpath(res[m]).startswith(os.path.realpath(PKG) + os.sep), (m, res[m])
    for m in THEIRS:
        assert os.path.realpath(res[m]).startswith(os.path.realpath(ref) + os.sep), (m, res[m])
    assert res["train_one_epoch"] == "stub-reference train_one_epoch"
    assert os.path.realpath(res["net_module_file"]) == os.path.realpath(os.path.join(PKG, "model", "ViBERTgrid_net.py"))


def test_overlay_against_the_real_checkout():
    """in the build container the real checkout is there: `pipeline.distributed_utils` (torch only) must come from it, the model from
    here, launched from inside the checkout like `python train_SROIE.py`.  Skipped where /root/reference does not exist (GPU box)."""
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "pipeline")) or not os.path.exists(os.path.join(PKG, "libvbg.so")):
        pytest.skip("no reference checkout here")
    code = ("import json, sys, torch\n"
            "sys.path.insert(0, %r)\n"                       # what `python train_SROIE.py` does with the script's directory
            "from model.ViBERTgrid_net import ViBERTgridNet\n"
            "from pipeline.distributed_utils import init_distributed_mode, setup_seed, is_main_process, save_on_master\n"
            "import pipeline.transform, pipeline.custom_loss\n"
            "print('RESOLVED ' + json.dumps({m: sys.modules[m].__file__ for m in ['model.ViBERTgrid_net', 'pipeline.distributed_utils', "
            "'pipeline.transform', 'pipeline.custom_loss']}))\n" % ref)
    res = _run([sys.executable, "-c", code], "/tmp", [PKG, ref])
    assert res["pipeline.distributed_utils"].startswith(ref + os.sep)
    for m in ("model.ViBERTgrid_net", "pipeline.transform", "pipeline.custom_loss"):
        assert os.path.realpath(res[m]).startswith(os.path.realpath(PKG) + os.sep), (m, res[m])
