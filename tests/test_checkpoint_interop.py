"""State-dict interoperability with the reference's checkpoints (key names, wrappers the
reference's get_state_dict strips: stream_helper.py:49-56), without a GPU."""
import os

import torch

from vcm_ts_amd import stream as S
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.intra import IntraNoAR
from vcm_ts_amd.params import dmc_spec, intra_spec, seeded_state_dict


def test_reference_style_checkpoints_load(tmp_path):
    sd = seeded_state_dict(dmc_spec(), seed=3)
    p = os.path.join(tmp_path, "dmc.pth")
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}}, p)
    y_q, mv_q = DMC.get_q_scales_from_ckpt(p)
    assert torch.equal(y_q, sd["y_q_scale"].reshape(-1)) and torch.equal(mv_q, sd["mv_y_q_scale"].reshape(-1))
    m = DMC(seed=0)
    missing, unexpected = m.load_state_dict(S.get_state_dict(p), strict=True)
    assert not missing and not unexpected
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # bare state dict / {"net": ...} layouts and the intra model
    si = seeded_state_dict(intra_spec(), seed=4)
    p2 = os.path.join(tmp_path, "intra.pth")
    torch.save({"net": si}, p2)
    assert torch.equal(IntraNoAR.get_q_scales_from_ckpt(p2), si["q_scale"].reshape(-1))
    im = IntraNoAR()
    im.load_state_dict(S.get_state_dict(p2), strict=True)
    assert torch.equal(im.state_dict()["enc.0.conv1.weight"], si["enc.0.conv1.weight"])
    # step-size accessors (video_model.py:255-261, image_model.py:50-52)
    q = m.get_curr_y_q(m.y_q_scale[1:3])
    assert q.shape == (2, 96, 1, 1) and torch.equal(q, torch.clamp_min(sd["y_q_basic"], 0.5) * sd["y_q_scale"][1:3])
    assert m.get_curr_mv_y_q(0.7).shape == (1, 64, 1, 1) and im.get_curr_q(1.0).shape == (1, 192, 1, 1)
    # the wrapper prefixes the codec's keys with "dmc." (checkpoint.py:37-42, save_dcvc_weights.py:12-15)
    from vcm_ts_amd.dcvc_hem import build_model, make_cfg

    w = build_model(make_cfg())
    assert set(w.state_dict()) == {"dmc." + k for k in dmc_spec()}


def test_update_builds_tables_without_gpu():
    m = DMC()
    m.update()
    assert m._tables["scale"][0].shape == (256, 103) and m._tables["bit_estimator_z"][0].shape[0] == 64
    m.update()  # second call is a no-op unless forced (common_model.py:75-80 / entropy_models.py:122-123)
    ec = m.entropy_coder
    m.update()
    assert m.entropy_coder is ec
    m.update(force=True)
    assert m.entropy_coder is not ec
