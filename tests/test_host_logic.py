"""CPU: host-side mirror of the reference interface -- construction API, state-dict contract,
closed forms, no-fallback behaviour, sharding helpers."""
import os
import tempfile

import pytest
import torch

import golden_util as gu


def vp_cfg_with_ckpt(cfg):
    from slotformer_amd.base_slots import build_model as bb
    scfg = gu.savi_cfg(cfg['resolution'][0], cfg['slot_dict']['num_slots'], slot_size=cfg['slot_dict']['slot_size'])
    scfg['dec_dict'] = {k: v for k, v in cfg['dec_dict'].items() if k != 'dec_ckp_path'}
    savi = bb(gu.ParamsView(scfg))
    path = os.path.join(tempfile.mkdtemp(), 'savi.pth')
    torch.save({'state_dict': savi.state_dict()}, path)
    full = {k: (dict(v) if isinstance(v, dict) else v) for k, v in cfg.items()}
    full['dec_dict']['dec_ckp_path'] = path
    return full, savi


@pytest.mark.parametrize('name,cfg', [('savi_c1', gu.C1_SAVI), ('savi_c2', gu.C2_SAVI), ('savi_c5', gu.C5_SAVI)])
def test_savi_state_dict_contract(name, cfg):
    """Same keys, order and shapes as the reference module => reference checkpoints load strict."""
    from slotformer_amd.base_slots import build_model
    m = build_model(gu.ParamsView(cfg))
    g = gu.load_golden(name)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == gu.shapes_from_golden(g)
    assert torch.equal(m.state_dict()['encoder_pos_embedding.grid'], torch.from_numpy(g['closed::encoder_pos_embedding.grid']))


@pytest.mark.parametrize('name,cfg', [('roll_c1', gu.C1_ROLL), ('roll_c2', gu.C2_ROLL), ('roll_c5', gu.C5_ROLL)])
def test_slotformer_state_dict_contract(name, cfg):
    from slotformer_amd.video_prediction import build_model
    full, savi = vp_cfg_with_ckpt(cfg)
    m = build_model(gu.ParamsView(full))
    g = gu.load_golden(name)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == gu.shapes_from_golden(g)
    assert torch.equal(m.state_dict()['rollouter.enc_t_pe'], torch.from_numpy(g['closed::rollouter.enc_t_pe']))
    # frozen decoder copied from the SAVi checkpoint by key prefix (slotformer.py:203-216)
    assert torch.equal(m.decoder[0][0].weight, savi.decoder[0][0].weight)
    assert not any(p.requires_grad for p in m.decoder.parameters())
    assert not m.rollouter.enc_t_pe.requires_grad
    m.train()
    assert not m.decoder.training


def test_steve_state_dict_contract():
    """STEVE carries its image side: `dvae.*` and `trans_decoder.*` keys with the reference's names, order and shapes
    (fixture from the reference's own STEVE class); the encoder-only fixture is the same list minus those keys."""
    from slotformer_amd.base_slots import build_model
    m = build_model(gu.ParamsView(gu.steve_tokens_cfg()))
    g = gu.load_golden('steve_tokens')
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == gu.shapes_from_golden(g)
    assert not any(p.requires_grad for p in m.dvae.parameters())
    m.train()
    assert not m.dvae.training
    cfg = dict(gu.C4_STEVE, dvae_dict=dict(down_factor=4, vocab_size=64, dvae_ckp_path=''),
               dec_dict=dict(dec_type='slate', dec_num_layers=1, dec_num_heads=4, dec_d_model=64),
               loss_dict=dict(use_img_recon_loss=False))
    m = build_model(gu.ParamsView(cfg))
    hot = [(k, tuple(v.shape)) for k, v in m.state_dict().items() if not k.startswith(('dvae.', 'trans_decoder.'))]
    assert hot == gu.shapes_from_golden(gu.load_golden('steve_c4'))


def test_steve_slotformer_state_dict_contract(tmp_path):
    """STEVESlotFormer: `dvae.*`, `decoder.*` (the STEVE Transformer decoder, loaded from the `trans_decoder.*` keys of
    a STEVE checkpoint and frozen), `rollouter.*` -- names, order and shapes of the reference class."""
    from slotformer_amd.base_slots import build_model as bb
    from slotformer_amd.video_prediction import build_model as bv
    steve = bb(gu.ParamsView(gu.steve_tokens_cfg()))
    path = str(tmp_path / 'steve.pth')
    torch.save({'state_dict': steve.state_dict()}, path)
    cfg = gu.steve_slotformer_cfg()
    with pytest.raises(AssertionError):  # 'Please provide pretrained Transformer decoder weight' (steve_slotformer.py:77)
        bv(gu.ParamsView(cfg))
    cfg['dec_dict']['dec_ckp_path'] = path
    m = bv(gu.ParamsView(cfg))
    g = gu.load_golden('steve_slotformer')
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == gu.shapes_from_golden(g)
    assert torch.equal(m.decoder.head.weight, steve.trans_decoder.head.weight)
    assert not any(p.requires_grad for p in m.decoder.parameters())
    m.train()
    assert not m.decoder.training and not m.dvae.training


def test_build_model_errors_match_reference():
    from slotformer_amd.base_slots import build_model as bb
    from slotformer_amd.video_prediction import build_model as bv
    with pytest.raises(NotImplementedError):
        bb(gu.ParamsView(dict(gu.C1_SAVI, model='Nope')))
    with pytest.raises(NotImplementedError):
        bv(gu.ParamsView(dict(gu.C1_ROLL, model='Nope')))
    with pytest.raises(AssertionError):  # 'Please provide pretrained decoder weight' (slotformer.py:201)
        bv(gu.ParamsView(gu.C1_ROLL))


def test_no_cpu_fallback():
    """The product path must fail loudly off-device; nothing routes through the oracle."""
    from slotformer_amd.base_slots import build_model
    m = build_model(gu.ParamsView(gu.C1_SAVI)).eval()
    m.testing = True
    with torch.no_grad(), pytest.raises(RuntimeError, match='no CPU fallback'):
        m({'img': gu.seeded_img(1, 1, 64)})
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m({'img': gu.seeded_img(1, 1, 64)})  # grad enabled: the training path (row N1) is HIP-only as well
    steve = build_model(gu.ParamsView(gu.steve_tokens_cfg())).eval()
    steve.testing = True
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        steve({'img': gu.seeded_img(1, 1, 64)})  # STEVE's training path is HIP-only too
    # the dVAE's own (Gumbel-softmax) training forward is HIP-only as well
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        steve.dvae.train()
        steve.dvae({'img': gu.seeded_img(1, 1, 64)[:, 0]})
    import slotformer_amd
    src_dir = os.path.dirname(slotformer_amd.__file__)
    for root, _, files in os.walk(src_dir):
        for f in files:
            if f.endswith('.py'):
                assert 'import oracle' not in open(os.path.join(root, f)).read(), f


def test_rollouter_asserts():
    from slotformer_amd.video_prediction.models import SlotRollouter, SingleStepSlotRollouter
    r = SlotRollouter(**gu.C1_ROLL['rollout_dict'])
    with pytest.raises(AssertionError, match='wrong burn-in steps'):
        r(torch.zeros(1, 5, 6, 128), 2)
    with pytest.raises(AssertionError):
        SingleStepSlotRollouter(**dict(gu.C5_ROLL['rollout_dict'], history_len=2))


def test_slotformer_alias_package():
    import importlib
    m = importlib.import_module('slotformer.base_slots')
    assert hasattr(m, 'build_model') and hasattr(m, 'build_dataset') and hasattr(m, 'build_method')
    from slotformer.video_prediction.models import SlotFormer  # noqa: F401


def test_shard_range():
    from slotformer_amd.parallel import shard_range
    for n in (0, 1, 7, 32, 33):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_slot_file_formats(tmp_path):
    """Row N3: pickle layout {split: {basename: [T,N,C] f32}} and per-sample PHYRE .npy files."""
    import numpy as np
    from slotformer_amd import slot_io
    files = ['/data/CLEVRER/videos/train/video_00000.mp4', '/data/CLEVRER/videos/train/video_00001.mp4']
    slots = np.random.RandomState(0).randn(2, 16, 7, 128)
    d = slot_io.slots_to_dict(files, slots)
    assert set(d) == {'video_00000.mp4', 'video_00001.mp4'} and d['video_00001.mp4'].dtype == np.float32
    path = str(tmp_path / 'out' / 'slots.pkl')
    slot_io.dump_slots(path, train=d, val={})
    back = slot_io.load_slots(path)
    assert set(back) == {'train', 'val'} and np.array_equal(back['train']['video_00000.mp4'], d['video_00000.mp4'])
    clip = slot_io.read_clip(back['train'], files[1], start_idx=1, n_sample_frames=6, frame_offset=2)
    assert clip.shape == (6, 7, 128) and np.array_equal(clip[2], d['video_00001.mp4'][5])
    with pytest.raises(ValueError):
        slot_io.read_clip(back['train'], '/x/missing.mp4', 0, 1, 1)
    root = str(tmp_path / 'phyre')
    assert slot_io.phyre_resume_index(root, 10, 20) == 10
    for i in (10, 11, 12):
        slot_io.save_phyre_slots(root, i, slots[0], vid_len=9)
    assert np.load(slot_io.phyre_path(root, 11)).shape == (9, 7, 128)
    assert slot_io.phyre_resume_index(root, 10, 20) == 12  # newest file is redone
    # a gap (crashed worker, other split bounds): the reference's loop (extract_phyre_slots.py:45-51) stops at the FIRST
    # missing file and restarts one before it -- files beyond the gap do not move the restart point
    slot_io.save_phyre_slots(root, 15, slots[0], vid_len=9)
    assert slot_io.phyre_resume_index(root, 10, 20) == 12
    for i in range(13, 20):
        slot_io.save_phyre_slots(root, i, slots[0], vid_len=9)
    assert slot_io.phyre_resume_index(root, 10, 20) == 19   # everything present: the last one is redone


def test_cu_mask_words_balance_every_shader_engine():
    """pipeline.encode_mask_words / the 'three' partition: bit b of word w = XCD b % 8, shader engine b // 8, CU row w.
    Every mask -- and what is left for the other streams -- must give each (XCD, shader engine) it touches the same
    number of CUs (the dispatcher deals workgroups in equal shares to them), and the three masks must tile the chip."""
    from slotformer_amd import pipeline as pl

    def per_se(words):
        cnt = {}
        for w, word in enumerate(words):
            for b in range(32):
                if word >> b & 1:
                    cnt[(b % 8, b // 8)] = cnt.get((b % 8, b // 8), 0) + 1
        return cnt

    for words, cus, ses in ((pl.ROLL_WORDS_3, 168, 3), (pl.LANE0_WORDS_3, 64, 1), (pl.LANE1_WORDS_3, 24, 3)):
        c = per_se(words)
        assert sum(c.values()) == cus and len(set(c.values())) == 1 and len(c) == 8 * ses
    for w in range(8):
        a, b, c = pl.ROLL_WORDS_3[w], pl.LANE0_WORDS_3[w], pl.LANE1_WORDS_3[w]
        assert a & b == 0 and a & c == 0 and b & c == 0 and (a | b | c) == 0xffffffff
    assert pl.encode_mask_words(0xff) == [0xff] * 8 and pl.encode_mask_words('ff') == [0xff] * 8
    assert pl.encode_mask_words('rows3') == [0xffffffff] * 3 + [0] * 5
    assert len(set(per_se(pl.encode_mask_words('rows3')).values())) == 1
    assert pl.encode_mask_words([1, 2, 3, 4, 5, 6, 7, 8]) == [1, 2, 3, 4, 5, 6, 7, 8]
    with pytest.raises(ValueError):
        pl.encode_mask_words('rows9')
    with pytest.raises(ValueError):
        pl.encode_mask_words([1, 2, 3])


def test_slot_file_link_next_to_the_weights(tmp_path):
    """extract_slots.py:86-93: the slot file is linked into the directory of the weights it was extracted with (`slots.pkl`, `<subset>_slots.pkl`)."""
    import os
    import numpy as np
    from slotformer_amd import slot_io
    wdir = tmp_path / 'ckpt'
    wdir.mkdir()
    weight = wdir / 'model_10.pth'
    weight.write_bytes(b'x')
    path = tmp_path / 'out' / 'clevrer_slots.pkl'
    slot_io.dump_slots(str(path), train=slot_io.slots_to_dict(['a/v0.mp4'], np.zeros((1, 2, 3, 4), np.float32)))
    ln = slot_io.link_slots(str(path), str(weight))
    assert ln == str(wdir / 'slots.pkl') and os.path.islink(ln) and list(slot_io.load_slots(ln)['train']) == ['v0.mp4']
    assert slot_io.link_slots(str(path), str(weight)) == ln     # replaced, not an error
    assert slot_io.link_slots(str(path), str(weight), subset='Collide').endswith('Collide_slots.pkl')


def test_switch_table_and_environment_check():
    """slotformer_amd/switches.py: at most 15 SF_* variables, each with a kind and a description; check_environment separates the ones that are
    set from the ones the table does not know (bench.py refuses to run with those)."""
    from slotformer_amd import switches
    assert 0 < len(switches.SWITCHES) <= 15
    assert all(k.startswith('SF_') and kind in ('product', 'tools') and len(text) > 10 for k, (kind, text) in switches.SWITCHES.items())
    known, unknown = switches.check_environment({'SF_PIPE_HYBRID': '3', 'SF_PIPE_HYBRD': '3', 'PATH': '/bin', 'SF_DBG': 'conv'})
    assert known == {'SF_PIPE_HYBRID': '3', 'SF_DBG': 'conv'} and unknown == ['SF_PIPE_HYBRD']
    assert switches.markdown_table().count('\n') == len(switches.SWITCHES) + 1
    # every variable the Python side reads is in the table
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = set()
    for rel in ('bench.py', 'slotformer_amd/pipeline.py', 'slotformer_amd/engine.py', 'slotformer_amd/_lib.py', 'slotformer_amd/harness.py', 'slotformer_amd/build.py'):
        seen |= set(re.findall(r"environ(?:\.get)?[\[(]\s*'(SF_[A-Z0-9_]+)'", open(os.path.join(root, rel)).read()))
    assert seen and seen <= set(switches.SWITCHES), sorted(seen - set(switches.SWITCHES))


def test_capture_gate_orders_captures_and_calls():
    """_lib.CAPTURE_GATE: calls hold it shared (many at once), a capture holds it exclusive (waits for the calls in flight, keeps new ones out, lets
    the capturing thread's own calls through)."""
    import threading
    import time
    from slotformer_amd._lib import _CaptureGate
    gate = _CaptureGate()
    log, lock = [], threading.Lock()

    def call(tag, hold):
        held = gate.enter_shared()
        with lock:
            log.append(('in', tag))
        time.sleep(hold)
        with lock:
            log.append(('out', tag))
        if held:
            gate.exit_shared()

    a = threading.Thread(target=call, args=('a', 0.2))
    b = threading.Thread(target=call, args=('b', 0.2))
    a.start(); b.start()
    time.sleep(0.05)
    assert [e for e in log if e[0] == 'in'] == [('in', 'a'), ('in', 'b')] or [e for e in log if e[0] == 'in'] == [('in', 'b'), ('in', 'a')]   # shared: both inside
    with gate:                                   # exclusive: only once both are out
        assert sorted(e for e in log if e[0] == 'out') == [('out', 'a'), ('out', 'b')]
        c = threading.Thread(target=call, args=('c', 0.0))
        c.start()
        time.sleep(0.1)
        assert ('in', 'c') not in log            # kept out while the capture lasts
        assert gate.enter_shared() is False      # the capturing thread's own calls pass (and hold nothing)
        with gate:                               # re-entrant
            pass
        assert ('in', 'c') not in log
    c.join(2)
    a.join(); b.join()
    assert ('out', 'c') in log
