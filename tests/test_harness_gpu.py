"""GPU: harness rows H1-H3 against goldens produced by the reference's own driver code."""
import pytest
import torch

import golden_util as gu
from test_engine_gpu import build, rel_err

pytestmark = pytest.mark.gpu


@torch.no_grad()
def test_h2_rollout_video_slots(dev):
    from slotformer_amd.harness import rollout_video_slots
    g = gu.load_golden('harness_h2')
    m, _ = build(gu.C1_ROLL, g, 301, dev, vp=True)
    N, C = 6, 128
    ori = torch.stack([gu.seeded_normal((128, N, C), 301 + 10 + i) for i in range(2)])
    out = rollout_video_slots(m, ori, int(g['frame_offset']))
    assert out.shape == (2, 160, N, C)
    assert torch.equal(out[:, :128].cpu(), ori)
    assert rel_err(out, g['slots']) < 2e-4


@torch.no_grad()
def test_h3_encode_then_rollout(dev):
    from slotformer_amd.harness import encode_then_rollout
    g = gu.load_golden('harness_h3')
    gs = {k[len('savi::'):]: v for k, v in g.items() if k.startswith('savi::')}
    gf = {k[len('sf::'):]: v for k, v in g.items() if k.startswith('sf::')}
    savi, _ = build(gu.C5_SAVI, gs, 501, dev)
    savi.testing = True
    sf, _ = build(gu.C5_ROLL, gf, 506, dev, vp=True)
    out = encode_then_rollout(savi, sf, gu.seeded_img(2, 1, 128), 5)
    assert rel_err(out['pred_slots'], g['pred_slots']) < 2e-4


@torch.no_grad()
def test_h1_extract_video_slots(dev):
    from slotformer_amd.harness import extract_video_slots
    g = gu.load_golden('savi_c1')
    m, _ = build(gu.C1_SAVI, g, 101, dev)
    m.testing = True
    vids = gu.seeded_img(2, 3, 64)
    slots = extract_video_slots(m, vids, batch_size=1)  # one video per call, as the reference does per GPU
    assert slots.dtype == torch.float32 and slots.device.type == 'cpu'
    assert rel_err(slots, g['post_slots']) < 5e-5
    assert rel_err(extract_video_slots(m, vids, batch_size=2), g['post_slots']) < 5e-5
