"""Plan-cache invalidation: derived / packed weight copies are keyed on (data_ptr, _version) of the parameters.
FlatAdam writes the parameters through a raw pointer and must bump the versions; modules must stay copyable / picklable after
a forward (the plans hold ctypes structures with raw pointers)."""
import copy
import io

import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def _slotformer(dev, seed=0):
    from test_engine_gpu import build
    m, _ = build(gu.C2_ROLL, gu.load_golden('roll_c2'), 202 + seed, dev, vp=True)
    return m


def test_flat_adam_invalidates_inference_plans(dev):
    """eval -> FlatAdam steps -> eval must equal a FRESHLY constructed model holding the stepped weights (the fused d=256
    rollout path serves fragment-ordered copies of every weight)."""
    from slotformer_amd import train
    cfg = gu.C2_ROLL
    rd = cfg['rollout_dict']
    m = _slotformer(dev)
    x = gu.seeded_normal((2, rd['history_len'], rd['num_slots'], rd['slot_size']), 3).to(dev)
    m.eval()
    with torch.no_grad():
        before = m.rollouter(x, 4).clone()              # builds the packed inference plan
    opt = train.FlatAdam([p for p in m.rollouter.parameters() if p.requires_grad], lr=1e-2)
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.
    slots = gu.seeded_normal((2, rd['history_len'] + 3, rd['num_slots'], rd['slot_size']), 4).to(dev)
    m.rollout_len = 3
    for _ in range(3):
        opt.zero_grad()
        out = m({'slots': slots})
        m.calc_train_loss({'slots': slots}, out)['slot_recon_loss'].backward()
        opt.step()
    m.eval()
    with torch.no_grad():
        after = m.rollouter(x, 4).clone()
    assert not torch.equal(before, after), 'the optimizer steps did not reach the inference path (stale packed weights)'
    fresh = _slotformer(dev, seed=1)
    fresh.load_state_dict(m.state_dict())
    fresh.eval()
    with torch.no_grad():
        want = fresh.rollouter(x, 4)
    assert torch.equal(after, want), (after - want).abs().max().item()


def test_invalidate_after_data_write(dev):
    from slotformer_amd import engine
    m = _slotformer(dev).eval()
    rd = gu.C2_ROLL['rollout_dict']
    x = gu.seeded_normal((2, rd['history_len'], rd['num_slots'], rd['slot_size']), 3).to(dev)
    with torch.no_grad():
        a = m.rollouter(x, 3).clone()
        m.rollouter.out_proj.weight.data.mul_(0.5)        # bypasses torch's version counter
        engine.invalidate(m)
        b = m.rollouter(x, 3).clone()
    assert not torch.equal(a, b)


def test_modules_stay_copyable_and_picklable_after_forward(dev):
    m = _slotformer(dev).eval()
    rd = gu.C2_ROLL['rollout_dict']
    x = gu.seeded_normal((1, rd['history_len'], rd['num_slots'], rd['slot_size']), 3).to(dev)
    with torch.no_grad():
        a = m.rollouter(x, 2)
        m2 = copy.deepcopy(m)                             # EMA copies, STEVESlotFormer's decoder copy
        assert torch.equal(m2.rollouter(x, 2), a)
        buf = io.BytesIO()
        torch.save(m, buf)                                # whole-module pickling
        buf.seek(0)
        m3 = torch.load(buf, weights_only=False)
        assert torch.equal(m3.rollouter(x, 2), a)
