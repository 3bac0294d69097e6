"""The argument contract of ``build_model`` (north star: main.py / main_multi.py run unchanged): namespaces produced by
the REFERENCE's own ``get_args_parser()`` on the command line of every shipped configs/training/*.sh
(tests/golden/args.json, tools/gen_golden_args.py) build every configuration here, and models/config.py's hand-kept
namespaces equal them field by field."""
import json
import os
from types import SimpleNamespace

import pytest
import torch


@pytest.fixture(scope="module")
def fixture(golden_dir):
    return json.load(open(os.path.join(golden_dir, "args.json")))


def test_parsers_and_scripts_are_the_expected_ones(fixture):
    assert set(fixture["parsers"]) == {"main.py", "main_multi.py"}
    assert set(fixture["configs"]) == {"Baseline.sh", "LateFusion.sh", "Encoder_CrossFusion.sh", "Backbone_CrossFusion.sh",
                                       "TransVOD++.sh", "TransVOD++_withdepth.sh"}
    assert fixture["configs"]["TransVOD++.sh"]["script"] == "main_multi.py"
    assert fixture["parsers"]["main.py"]["num_feature_levels"] == 4          # the parser default; every script passes 1


def test_config_module_equals_the_reference_namespaces(fixture):
    from models import config
    paths = {"dformer_weights", "device"}                                      # the caller's business
    for script, rec in fixture["configs"].items():
        ref = rec["namespace"]
        mine = vars(config.config_args(script))
        assert set(mine) - set(ref) <= {"depth_type"}, f"{script}: fields build_model reads that the parser does not produce"
        for key, val in mine.items():
            if key in paths or key not in ref:
                continue
            assert ref[key] == val, f"{script}: {key} is {ref[key]!r} in the reference's namespace, {val!r} here"
    # the parser defaults the module's base layer relies on (flags no script passes)
    for script in ("main.py", "main_multi.py"):
        d = fixture["parsers"][script]
        for key in ("enc_layers", "dec_layers", "dim_feedforward", "hidden_dim", "nheads", "dec_n_points", "enc_n_points",
                    "dpth_n_points", "two_stage", "aux_loss", "position_embedding", "masks", "lr_backbone"):
            assert d[key] == config._DEFAULTS[key], (script, key)


@pytest.mark.parametrize("script", ["Baseline.sh", "LateFusion.sh", "Encoder_CrossFusion.sh", "Backbone_CrossFusion.sh",
                                    "TransVOD++.sh", "TransVOD++_withdepth.sh"])
def test_build_model_takes_the_reference_namespace(fixture, script):
    """``build_model(Namespace(**what the reference's parser produced))`` - every field of the real namespace present,
    none added - builds the configuration (weights files aside: the paths of --dformer_weights / --resume are not opened
    by build_model when absent)."""
    from models import build_model
    ns = dict(fixture["configs"][script]["namespace"])
    ns.update(device="cpu", dformer_weights=None)
    model, criterion, post = build_model(SimpleNamespace(**ns))
    assert "bbox" in post
    keys = set(model.state_dict())
    assert any(k.startswith("backbone.0.body.layer4") for k in keys)
    depth = ns["use_depth"] and ns["fusion_type"] in ("LateFusion", "Encoder_CrossFusion")
    assert any(k.startswith("input_proj_depth") for k in keys) == depth
    assert any("temporal_query_layer1" in k for k in keys) == (ns["dataset_file"] == "vid_multi_plusplus")
    if ns["dataset_file"] == "vid_multi_plusplus":
        assert model.transformer.num_ref_frames == 4
    n = sum(p.numel() for p in model.parameters())
    assert n > 3e7
