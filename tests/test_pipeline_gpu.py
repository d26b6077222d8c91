"""The batch pipeline (slotformer_amd/pipeline.py: encode of batch i+1 on CU-masked lanes, each a share of the videos, beside the
rollout hipGraph of batch i, double-buffered slots, work stealing) must give bit-identical results to the serial
`savi({'img'}) -> rollout` sequence -- with DIFFERENT inputs and noise per batch, so that a slot-buffer, feature-buffer or
event mistake shows (VERDICT r01 'pipeline correctness is untested')."""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def _models(dev, savi_cfg, roll_cfg, seed=0):
    from slotformer_amd.base_slots import build_model
    from slotformer_amd.video_prediction.models import SlotRollouter
    torch.manual_seed(seed)
    savi = build_model(gu.ParamsView(savi_cfg)).eval().to(dev)
    savi.testing = True
    roll = SlotRollouter(**roll_cfg['rollout_dict']).eval().to(dev)
    return savi, roll


PAIR_OPTS = {'attn_heads': 8}   # the 'pair' partition's throughput kernel forms (all-heads attention workgroups, 128-row FFN
                                # workgroups, no seam launches) give the bits of the defaults -- asserted below


def _serial_reference(savi, roll, imgs, noises, T, H, opts=None):
    """The plain module API, one batch after the other on the default stream (opts: the kernel forms of the pipeline under
    test; they give the same bits as the library defaults, checked below)."""
    from slotformer_amd import engine
    outs = []
    for img, nz in zip(imgs, noises):
        post = savi({'img': img, 'noise': nz})['post_slots']
        buf = torch.zeros(post.shape[0], T + H, post.shape[2], post.shape[3], device=img.device)
        buf[:, :T] = post
        engine.rollout(roll, buf, T, H, opts=opts)
        outs.append(buf)
    return torch.stack(outs, 0)


def _opts_of(partition):
    return PAIR_OPTS if partition == 'pair' else None


def _close(a, b, tol=2e-5):
    """max |a - b| / max |b|: the token-stationary layer launches against the other forms (one accumulator per output block instead of per-chunk
    partial sums: ~1e-6 per layer, measured 5e-6 over a 50-step rollout)"""
    return ((a - b).abs().max() / b.abs().max()).item() <= tol


@pytest.mark.parametrize('B,steal,nbatch,partition,tok', [(32, None, 9, 'pair', False), (32, 1, 11, 'pair', False), (32, 1.25, 13, 'pair', False),
                                                          (32, None, 14, 'pair', None), (32, None, 20, 'pair', None), (32, 1, 7, 'pair', None),
                                                          (5, 2, 12, 'pair', None), (5, 0.5, 12, 'pair', None), (5, None, 7, 'pair', None), (32, None, 6, 'three', None),
                                                          (32, 1, 5, 'three', None), (5, 2, 7, 'three', None), (32, 1, 6, 'two', None), (32, 0, 5, 'two', None),
                                                          (5, 1, 5, 'two', None), (5, 2, 7, 'two', None)])
def test_pipeline_matches_serial(dev, B, steal, nbatch, partition, tok):
    """tok None = what the pipeline picks: at 32 videos per batch on the 'pair' partition the FULL units (6 batches) run the layers before the last as
    token-stationary launches -- compared with the serial module calls to 2e-5 and, bit for bit, with the serial SCHEDULE of the same unit graphs (what a
    buffer / event mistake of the pipelined schedule would break); everything else bit for bit with the serial module calls."""
    from slotformer_amd.pipeline import EncodeRolloutPipeline
    T, H = 6, 12
    savi, roll = _models(dev, gu.C2_SAVI, gu.C2_ROLL)
    rs = np.random.RandomState(7)
    imgs = [torch.from_numpy((rs.rand(B, T, 3, 128, 128) * 2 - 1).astype(np.float32)).to(dev) for _ in range(nbatch)]
    noises = [torch.from_numpy(rs.standard_normal((B, T, 7, 128)).astype(np.float32)).to(dev) for _ in range(nbatch)]
    with torch.no_grad():
        ref = _serial_reference(savi, roll, imgs, noises, T, H, _opts_of(partition))
        if partition == 'pair':   # the throughput kernel forms agree with the default ones to rounding
            ref_default = _serial_reference(savi, roll, imgs[:2], noises[:2], T, H)
            assert torch.equal(ref[:2], ref_default)
        pipe = EncodeRolloutPipeline(savi, roll, B, T, H, steal_steps=steal, partition=partition, tok=tok)
        use_tok = partition == 'pair' and B == 32 and tok is None
        assert pipe.tok == use_tok
        assert pipe.partition == partition and len(pipe.lanes) == (2 if partition == 'three' else 1)
        assert len(pipe.roll_streams) == (2 if partition == 'pair' else 1) and len(pipe.bufs) == (4 if partition == 'pair' else 2)
        assert pipe.G == ((6 if use_tok else 4) if partition == 'pair' else 1) and pipe.bufs[0].shape[0] == pipe.G * B
        assert [lo for _, lo, _ in pipe.lanes] + [pipe.lanes[-1][2]] == ([0, B - max(1, round(B * 24 / 88)), B] if partition == 'three' else [0, B])
        out = pipe.run(imgs, noises)
        torch.cuda.synchronize()
        assert out.shape == ref.shape
        # the serial schedule of the same object (graphs, one stream)
        out3 = pipe.run(imgs, noises, serial=True)
        torch.cuda.synchronize()
        if use_tok:
            assert _close(out, ref), ((out - ref).abs().max() / ref.abs().max()).item()
            assert torch.equal(out[:, :, :T], ref[:, :, :T])            # (the encoded frames: bit for bit)
            if nbatch >= 6:
                assert not torch.equal(out[:6], ref[:6])                # (a full unit took the other kernel form)
            assert torch.equal(out, out3), (out - out3).abs().max().item()
        else:
            assert torch.equal(out, ref), (out - ref).abs().max().item()
            assert torch.equal(out3, ref)
        # batches differ from each other (so a stale-buffer bug could not hide) and a second run reproduces the first
        assert not torch.equal(out[0], out[1])
        out2 = pipe.run(imgs, noises)
        torch.cuda.synchronize()
        assert torch.equal(out2, out)
        pipe.close()


@pytest.mark.parametrize('hybrid,nbatch,tok', [(2, 21, False), (5, 26, False), (0, 17, False), (2, 26, None), (None, 31, None)])
def test_pipeline_hybrid_lane_matches_serial(dev, hybrid, nbatch, tok):
    """Runs long enough to leave the whole-chip fill (12 batches): behind it every hybrid-th batch is encoded on an unmasked stream
    through the second fill graph, beside the CU-masked lane; unit buffers are reused (more than four units).  Bit for bit the
    serial calls."""
    from slotformer_amd.pipeline import EncodeRolloutPipeline
    B, T, H = 32, 6, 8
    savi, roll = _models(dev, gu.C2_SAVI, gu.C2_ROLL)
    rs = np.random.RandomState(17)
    base = [torch.from_numpy((rs.rand(B, T, 3, 128, 128) * 2 - 1).astype(np.float32)).to(dev) for _ in range(5)]
    nz = [torch.from_numpy(rs.standard_normal((B, T, 7, 128)).astype(np.float32)).to(dev) for _ in range(7)]
    imgs = [base[j % 5] for j in range(nbatch)]
    noises = [nz[j % 7] for j in range(nbatch)]     # (35 distinct (frames, noise) pairs)
    with torch.no_grad():
        ref = _serial_reference(savi, roll, imgs, noises, T, H, PAIR_OPTS)
        pipe = EncodeRolloutPipeline(savi, roll, B, T, H, hybrid=hybrid, tok=tok)
        # (token-stationary full units -- the default at this batch size: units of 6 batches, every 8th batch on the hybrid lane, 18 fill batches)
        assert pipe.tok == (tok is None) and pipe.hybrid == (8 if hybrid is None else hybrid) and pipe.fill_batches == (18 if pipe.tok else 12)
        serial = pipe.run(imgs, noises, serial=True) if pipe.tok else None
        for _ in range(2):
            out = pipe.run(imgs, noises)
            torch.cuda.synchronize()
            if pipe.tok:
                assert _close(out, ref) and torch.equal(out, serial), (out - serial).abs().max().item()
            else:
                assert torch.equal(out, ref), (out - ref).abs().max().item()
        pipe.close()
    assert EncodeRolloutPipeline.__init__.__defaults__ is not None


def test_pipeline_without_cu_partition_and_graph(dev):
    from slotformer_amd.pipeline import EncodeRolloutPipeline
    B, T, H, nbatch = 4, 6, 5, 4
    savi, roll = _models(dev, gu.C2_SAVI, gu.C2_ROLL, seed=3)
    rs = np.random.RandomState(11)
    imgs = [torch.from_numpy((rs.rand(B, T, 3, 128, 128) * 2 - 1).astype(np.float32)).to(dev) for _ in range(nbatch)]
    noises = [torch.from_numpy(rs.standard_normal((B, T, 7, 128)).astype(np.float32)).to(dev) for _ in range(nbatch)]
    with torch.no_grad():
        refs = {False: _serial_reference(savi, roll, imgs, noises, T, H), True: _serial_reference(savi, roll, imgs, noises, T, H, PAIR_OPTS)}
        for kw in (dict(encode_cu_word=0), dict(use_graph=False), dict(encode_cu_word=0, use_graph=False, steal_steps=0),
                   dict(partition='none', steal_steps=1), dict(partition='two', encode_cu_word='rows2'),
                   dict(partition='two', rollout_opts={'attn_heads': 8, 'ffn_rows': 64}),
                   # the row-tile forms of both layer blocks (the default of units of >= 4096 token rows: bench.py's C2 / C5 units)
                   dict(rollout_opts={'attn_rows': 128, 'ffn_tile': True, 'seam': False}),
                   dict(partition='two', rollout_opts={'attn_rows': 128, 'ffn_tile': True})):
            pipe = EncodeRolloutPipeline(savi, roll, B, T, H, **kw)
            out = pipe.run(imgs, noises)
            torch.cuda.synchronize()
            all_heads = pipe.rollout_opts is not None and pipe.rollout_opts.attn_heads_per_wg == 8
            assert torch.equal(out, refs[all_heads]), kw
            pipe.close()


def test_extract_and_rollout_entry(dev):
    """harness.extract_and_rollout: full batches through the pipeline + a ragged tail, equal to the serial module calls."""
    from slotformer_amd import harness
    T, H, bs, V = 6, 8, 4, 14     # 3 full batches + 2 videos
    savi, roll = _models(dev, gu.C2_SAVI, gu.C2_ROLL, seed=5)
    rs = np.random.RandomState(3)
    videos = torch.from_numpy((rs.rand(V, T, 3, 128, 128) * 2 - 1).astype(np.float32))
    noises = torch.from_numpy(rs.standard_normal((V, T, 7, 128)).astype(np.float32))
    with torch.no_grad():
        out = harness.extract_and_rollout(savi, roll, videos, H, batch_size=bs, noises=noises)
        chunks = [(0, 4), (4, 8), (8, 12), (12, 14)]
        ref = _serial_reference(savi, roll, [videos[a:b].to(dev) for a, b in chunks[:3]], [noises[a:b].to(dev) for a, b in chunks[:3]], T, H, PAIR_OPTS)
        tail = _serial_reference(savi, roll, [videos[12:].to(dev)], [noises[12:].to(dev)], T, H, PAIR_OPTS)
        torch.cuda.synchronize()
        assert out.shape == (V, T + H, 7, 128)
        assert torch.equal(out[:12], ref.reshape(12, T + H, 7, 128))
        assert torch.equal(out[12:], tail[0])
        # without fixed noise the call still runs (fresh kernel noise per frame) and differs from the fixed-noise run
        out_r = harness.extract_and_rollout(savi, roll, videos, H, batch_size=bs)
        assert out_r.shape == out.shape and torch.isfinite(out_r).all() and not torch.equal(out_r, out)
        # `videos` above was a HOST tensor: the frames were uploaded batch by batch by the pipeline's copy stage.  Device-resident
        # input and host-resident output (pinned, downloaded behind each rollout) give the same slots; the pipeline object is
        # reused between the calls (graphs captured once)
        n_pipes = len(harness._PIPES)
        out_d = harness.extract_and_rollout(savi, roll, videos.to(dev), H, batch_size=bs, noises=noises)
        out_h = harness.extract_and_rollout(savi, roll, videos, H, batch_size=bs, noises=noises, to_host=True)
        assert torch.equal(out_d, out) and not out_h.is_cuda and out_h.is_pinned() and torch.equal(out_h, out.cpu())
        # pre-pinned input (copied straight from where it lies) = pageable input (through the ring of page-locked staging buffers)
        assert torch.equal(harness.extract_and_rollout(savi, roll, videos.pin_memory(), H, batch_size=bs, noises=noises), out)
        assert len(harness._PIPES) == n_pipes == 1
        # another shape: only the most recently used pipeline stays alive by default (MAX_PIPELINES = 1: idle hardware queues
        # are not free on this platform), the earlier one is closed -- and rebuilt, with the same results, when its shape returns
        first = next(iter(harness._PIPES.values()))[2]
        out_2 = harness.extract_and_rollout(savi, roll, videos.to(dev), H, batch_size=2, noises=noises)
        assert len(harness._PIPES) == 1 and next(iter(harness._PIPES.values()))[2] is not first
        assert torch.equal(out_2, out)      # (per-video kernels: the slots do not depend on the batch size)
        assert torch.equal(harness.extract_and_rollout(savi, roll, videos.to(dev), H, batch_size=bs, noises=noises), out)
        harness.release_pipelines()
        assert not harness._PIPES


def test_unit_batches_for_long_runs(dev):
    """Long runs of small batches take larger rollout units (pipeline.unit_batches_for, used by harness.extract_and_rollout and bench.py):
    a unit's row-tile launches should fill one round of a rollout stream's 64 CUs.  The slots do not depend on the unit size."""
    from slotformer_amd import harness
    from slotformer_amd.pipeline import unit_batches_for
    T, H = 6, 4
    savi, roll = _models(dev, gu.C2_SAVI, gu.C2_ROLL, seed=9)
    assert unit_batches_for(roll, 32, 100, T) == 6 and unit_batches_for(roll, 32, 20, T) == 6     # C2: token-stationary units of 64 workgroups (6 x 32 videos / 3 per workgroup)
    assert unit_batches_for(roll, 14, 40, T) == 6 and unit_batches_for(roll, 2, 40, T) == 8             # 588 rows per batch (C4: 576 -> 7)
    # short runs of small batches: an even number of equal units of up to 8192 token rows (C4 at 20 batches: two units of 10)
    assert unit_batches_for(roll, 14, 20, T) == 10 and unit_batches_for(roll, 14, 12, T) == 6
    # no even split into equal units: the smallest even number of units of <= 8192 rows, the last one shorter (29 batches: 8, 8, 8, 5)
    assert unit_batches_for(roll, 14, 29, T) == 8 and unit_batches_for(roll, 14, 21, T) == 11 and unit_batches_for(roll, 2, 39, T) == 20
    assert unit_batches_for(roll, 14, 9, T) is None   # (short runs keep the default)
    # a rollouter of more than four layers (C4: eight) takes the token-stationary units in runs of three units or more only, and the pipeline takes the
    # form when its units were sized for such a run (or tok=True)
    from slotformer_amd.pipeline import tok_unit_batches, EncodeRolloutPipeline
    from slotformer_amd.video_prediction.models import SlotRollouter
    torch.manual_seed(3)
    deep = SlotRollouter(**gu.C4_ROLL['rollout_dict']).eval().to(dev)
    assert len(deep.transformer_encoder.layers) == 8
    assert tok_unit_batches(deep, 32, 6) is None and tok_unit_batches(deep, 32, 6, 10) is None and tok_unit_batches(deep, 32, 6, 18) == 6
    assert unit_batches_for(deep, 32, 20, 6) == 6 and unit_batches_for(deep, 32, 10, 6) != 6
    from slotformer_amd.pipeline import encode_group_for
    assert [encode_group_for(16, 20), encode_group_for(8, 48), encode_group_for(32, 20), encode_group_for(2, 41), encode_group_for(2, 16)] == [2, 4, 1, 1, 2]
    bs, V = 2, 83      # 41 full batches (units of 8) + 1 video
    rs = np.random.RandomState(5)
    base = torch.from_numpy((rs.rand(7, T, 3, 128, 128) * 2 - 1).astype(np.float32)).to(dev)
    videos = base[torch.arange(V) % 7]
    noises = torch.from_numpy(rs.standard_normal((V, T, 7, 128)).astype(np.float32)).to(dev)
    with torch.no_grad():
        out = harness.extract_and_rollout(savi, roll, videos, H, batch_size=bs, noises=noises)
        pipe = next(iter(harness._PIPES.values()))[2]
        assert pipe.G == 8
        ref = harness.extract_and_rollout(savi, roll, videos, H, batch_size=bs, noises=noises, group=4)
        assert next(iter(harness._PIPES.values()))[2].G == 4
        assert torch.equal(out, ref)
        one = _serial_reference(savi, roll, [videos[10:12]], [noises[10:12]], T, H, PAIR_OPTS)
        assert torch.equal(out[10:12], one[0])
        # 16 batches of 2: handed to the pipeline two at a time (encode_group_for): the same slots
        out_e = harness.extract_and_rollout(savi, roll, videos[:32], H, batch_size=bs, noises=noises[:32])
        assert next(iter(harness._PIPES.values()))[2].B == 2 * bs and torch.equal(out_e, out[:32])
        harness.release_pipelines()


@pytest.mark.parametrize('group,partition,nbatch', [(1, 'pair', 7), (3, 'pair', 8), (2, 'two', 5), (2, 'none', 3), (2, 'pair', 1), (2, 'pair', 9), (4, 'pair', 21),
                                                    (4, 'pair', 7)])
def test_pipeline_groups(dev, group, partition, nbatch):
    """batches per rollout graph (`group`): any grouping, ragged last unit included, gives the serial results bit for bit"""
    from slotformer_amd.pipeline import EncodeRolloutPipeline
    B, T, H = 6, 6, 9
    savi, roll = _models(dev, gu.C2_SAVI, gu.C2_ROLL, seed=2)
    rs = np.random.RandomState(17)
    imgs = [torch.from_numpy((rs.rand(B, T, 3, 128, 128) * 2 - 1).astype(np.float32)).to(dev) for _ in range(nbatch)]
    noises = [torch.from_numpy(rs.standard_normal((B, T, 7, 128)).astype(np.float32)).to(dev) for _ in range(nbatch)]
    with torch.no_grad():
        ref = _serial_reference(savi, roll, imgs, noises, T, H, _opts_of(partition))
        pipe = EncodeRolloutPipeline(savi, roll, B, T, H, partition=partition, group=group, steal_steps=1.5)
        assert pipe.G == group
        for _ in range(2):
            out = pipe.run(imgs, noises)
            torch.cuda.synchronize()
            assert torch.equal(out, ref), (out - ref).abs().max().item()
        assert sum(pipe.completion_batches) == nbatch and len(pipe.completion_events) == len(pipe.completion_batches)
        pipe.close()


def _build_pair(dev, savi_cfg, roll_cfg, single_step=False, seed=0):
    from slotformer_amd.base_slots import build_model
    from slotformer_amd.video_prediction.models import SlotRollouter, SingleStepSlotRollouter
    torch.manual_seed(seed)
    savi_cfg = dict(savi_cfg)
    if savi_cfg['model'] == 'STEVE' and 'dvae_dict' not in savi_cfg:   # the image side is not on this path: a small one
        savi_cfg.update(dvae_dict=dict(down_factor=4, vocab_size=64, dvae_ckp_path=''),
                        dec_dict=dict(dec_type='slate', dec_num_layers=1, dec_num_heads=4, dec_d_model=64),
                        loss_dict=dict(use_img_recon_loss=False))
    savi = build_model(gu.ParamsView(savi_cfg)).eval().to(dev)
    savi.testing = True
    cls = SingleStepSlotRollouter if single_step else SlotRollouter
    rd = dict(roll_cfg['rollout_dict'])
    roll = cls(**rd).eval().to(dev)
    return savi, roll


@pytest.mark.parametrize('name', ['C1', 'C4', 'C5'])
def test_pipeline_other_configs(dev, name):
    """The pipeline with the other BASELINE model pairs: C1 = OBJ3D SAVi with
    kld_method 'none' (`_sample_dist` returns the mean, savi.py:355-365: NO noise may be applied, also not the caller's) +
    SlotRollouter d_model 128; C4 = STEVE (no kernel distribution at all) + 8-layer SlotRollouter, slot size 192; C5 = PHYRE
    SAVi (kld 'none', Transformer + LSTM predictor) + SingleStepSlotRollouter with burn-in 1.  Reference = the plain module
    API (`savi({'img'})` applies the reference's own gating), batch by batch."""
    from slotformer_amd import engine
    from slotformer_amd.pipeline import EncodeRolloutPipeline
    savi_cfg, roll_cfg, single, res, T, H, B = {'C1': (gu.C1_SAVI, gu.C1_ROLL, False, 64, 6, 5, 4),
                                                 'C4': (gu.C4_STEVE, gu.C4_ROLL, False, 128, 6, 4, 3),
                                                 'C5': (gu.C5_SAVI, gu.C5_ROLL, True, 128, 1, 9, 5)}[name]
    savi, roll = _build_pair(dev, savi_cfg, roll_cfg, single_step=single, seed=4)
    N, D = roll.num_slots, roll.in_proj.in_features
    nbatch = 5
    rs = np.random.RandomState(23)
    imgs = [torch.from_numpy((rs.rand(B, T, 3, res, res) * 2 - 1).astype(np.float32)).to(dev) for _ in range(nbatch)]
    noises = [torch.from_numpy(rs.standard_normal((B, T, N, D)).astype(np.float32)).to(dev) for _ in range(nbatch)]
    key = 'post_slots' if hasattr(savi, 'kernel_dist_layer') else 'slots'
    with torch.no_grad():
        def reference(opts):
            refs = []
            for img in imgs:
                post = savi({'img': img})[key]
                buf = torch.zeros(B, T + H, N, D, device=dev)
                buf[:, :T] = post
                engine.rollout(roll, buf, T, H, opts=opts)
                refs.append(buf)
            return torch.stack(refs, 0)

        ref_default = reference(None)
        assert torch.isfinite(ref_default).all()
        for kw in (dict(), dict(partition='three'), dict(group=1, steal_steps=0), dict(steal_steps=1.5)):
            pipe = EncodeRolloutPipeline(savi, roll, B, T, H, **kw)
            # C1's d_model 128 is not on the fused-layer path: one batch per rollout unit (the generic GEMMs pick tiles by size)
            assert pipe.fused == (name != 'C1') and (pipe.G == 1 or pipe.fused)
            ref = ref_default if pipe.rollout_opts is None else reference(pipe.rollout_opts)   # the pipeline's kernel forms
            assert torch.equal(ref, ref_default)
            for nz in (None, noises):     # caller-supplied noise must be ignored too when the model samples nothing
                out = pipe.run(imgs, nz)
                torch.cuda.synchronize()
                assert torch.equal(out, ref), (name, kw, nz is None, (out - ref).abs().max().item())
            pipe.close()
        with pytest.raises(RuntimeError):
            EncodeRolloutPipeline(savi, roll, B, T + 1, H)   # burn-in must be what the rollouter consumes


def test_pipeline_recaptures_after_a_weight_update(dev):
    """The graphs point at packed weight copies of the rollouter's plan: after an in-place parameter update run() must
    notice (plan signature) and re-capture instead of replaying the old weights."""
    from slotformer_amd.pipeline import EncodeRolloutPipeline
    B, T, H, nbatch = 4, 6, 4, 3
    savi, roll = _models(dev, gu.C2_SAVI, gu.C2_ROLL, seed=6)
    rs = np.random.RandomState(29)
    imgs = [torch.from_numpy((rs.rand(B, T, 3, 128, 128) * 2 - 1).astype(np.float32)).to(dev) for _ in range(nbatch)]
    noises = [torch.from_numpy(rs.standard_normal((B, T, 7, 128)).astype(np.float32)).to(dev) for _ in range(nbatch)]
    with torch.no_grad():
        pipe = EncodeRolloutPipeline(savi, roll, B, T, H)
        pipe2 = EncodeRolloutPipeline(savi, roll, 2 * B, T, H)   # a second live pipeline: private workspaces, no interference
        out0 = pipe.run(imgs, noises).clone()
        assert torch.equal(out0, _serial_reference(savi, roll, imgs, noises, T, H, PAIR_OPTS))
        roll.out_proj.weight.mul_(1.5)     # bumps the version counter
        ref1 = _serial_reference(savi, roll, imgs, noises, T, H, PAIR_OPTS)
        out1 = pipe.run(imgs, noises)
        assert torch.equal(out1, ref1) and not torch.equal(out1, out0)
        pipe.close()
        pipe2.close()


def test_pipeline_recaptures_after_an_encoder_weight_update(dev):
    """The ENCODE graphs point into the SAVi encoder's plan (packed conv weights, fragment copies, folded Slot-Attention matrices):
    after an in-place update of an encoder parameter, a load_state_dict, or an `engine.invalidate` + eager encode that rebuilt (and
    freed) the plan, run() must re-capture them instead of replaying the old packed copies."""
    from slotformer_amd import engine
    from slotformer_amd.pipeline import EncodeRolloutPipeline
    B, T, H, nbatch = 4, 6, 4, 5
    savi, roll = _models(dev, gu.C2_SAVI, gu.C2_ROLL, seed=9)
    rs = np.random.RandomState(37)
    imgs = [torch.from_numpy((rs.rand(B, T, 3, 128, 128) * 2 - 1).astype(np.float32)).to(dev) for _ in range(nbatch)]
    noises = [torch.from_numpy(rs.standard_normal((B, T, 7, 128)).astype(np.float32)).to(dev) for _ in range(nbatch)]
    with torch.no_grad():
        pipe = EncodeRolloutPipeline(savi, roll, B, T, H, encode_graph=True)
        assert len(pipe._enc_graphs) >= 1
        out0 = pipe.run(imgs, noises).clone()
        assert torch.equal(out0, _serial_reference(savi, roll, imgs, noises, T, H, PAIR_OPTS))
        # (1) a packed weight (conv fragments), a folded one (project_k -> Wk^T Wq) and a plain vector, updated in place
        savi.encoder[2][0].weight.mul_(1.25)
        savi.slot_attention.project_k.weight.mul_(0.8)
        savi.encoder_out_layer[0].bias.add_(0.1)
        ref1 = _serial_reference(savi, roll, imgs, noises, T, H, PAIR_OPTS)
        out1 = pipe.run(imgs, noises).clone()
        assert torch.equal(out1, ref1) and not torch.equal(out1, out0)
        # (2) a second checkpoint loaded into the same module
        other, _ = _models(dev, gu.C2_SAVI, gu.C2_ROLL, seed=10)
        savi.load_state_dict(other.state_dict())
        ref2 = _serial_reference(savi, roll, imgs, noises, T, H, PAIR_OPTS)
        out2 = pipe.run(imgs, noises).clone()
        assert torch.equal(out2, ref2) and not torch.equal(out2, out1)
        # (3) the plan dropped and rebuilt by an eager call elsewhere (same parameter values: same results, fresh graphs)
        old = pipe._enc_plan
        engine.invalidate(savi)
        engine.savi_encode(savi, imgs[0], noise=noises[0])
        out3 = pipe.run(imgs, noises)
        assert pipe._enc_plan is not old and torch.equal(out3, ref2)
        pipe.close()


@pytest.mark.parametrize('name', ['C2', 'C5'])
def test_pipeline_with_the_encode_under_a_graph(dev, name):
    """encode_graph=True: the ~60 launches of every encode are replayed from hipGraphs over fixed buffers (frames, injected kernel
    noise and stolen features staged into them): bit-identical to the eager encode, also with the predictor's LSTM state
    allocated inside the capture (C5) and with work stealing (its own graph per number of stolen steps)."""
    from slotformer_amd.pipeline import EncodeRolloutPipeline
    if name == 'C2':
        savi, roll = _models(dev, gu.C2_SAVI, gu.C2_ROLL, seed=8)
        B, T, H, res = 6, 6, 5, 128
    else:
        savi, roll = _build_pair(dev, gu.C5_SAVI, gu.C5_ROLL, single_step=True, seed=8)
        B, T, H, res = 5, 1, 7, 128
    N, D = roll.num_slots, roll.in_proj.in_features
    rs = np.random.RandomState(31)
    imgs = [torch.from_numpy((rs.rand(B, T, 3, res, res) * 2 - 1).astype(np.float32)).to(dev) for _ in range(9)]
    noises = [torch.from_numpy(rs.standard_normal((B, T, N, D)).astype(np.float32)).to(dev) for _ in range(9)]
    with torch.no_grad():
        pipe = EncodeRolloutPipeline(savi, roll, B, T, H, steal_steps=1.25 if name == 'C2' else 0, encode_graph=False)
        ref = pipe.run(imgs, noises).clone()
        pipe.close()
        pipe = EncodeRolloutPipeline(savi, roll, B, T, H, steal_steps=1.25 if name == 'C2' else 0, encode_graph=True)
        for _ in range(2):
            out = pipe.run(imgs, noises)
            torch.cuda.synchronize()
            assert torch.equal(out, ref), (out - ref).abs().max().item()
        assert len(pipe._enc_graphs) >= 1
        pipe.close()


@pytest.mark.parametrize('res', [128, 64])
def test_pipeline_with_the_decode_stage(dev, res):
    """decoder=...: the predicted frames of every batch decoded behind their rollout (reconstruction + postproc_mask segmentation, what
    test_vp.py scores): bit-identical to the serial calls -- module decode + vp_utils.postproc_mask per batch -- through the pipelined
    schedule, the serial one and harness.extract_and_rollout with a ragged tail; int64 and uint8 segmentations agree."""
    from slotformer_amd import engine, harness
    from slotformer_amd.base_slots import build_model
    from slotformer_amd.pipeline import EncodeRolloutPipeline
    from slotformer_amd.video_prediction.models import SlotRollouter
    from slotformer_amd.video_prediction.vp_utils import postproc_mask
    B, T, H, nbatch = 3, 6, 4, 5
    torch.manual_seed(21)
    savi = build_model(gu.ParamsView(gu.savi_cfg(res, 7, iters=2, kernel_mlp=False, pred='mlp', rnn=False, kld='var-0.01'))).eval().to(dev)
    savi.testing = True
    roll = SlotRollouter(**gu.C2_ROLL['rollout_dict']).eval().to(dev)
    rs = np.random.RandomState(41)
    imgs = [torch.from_numpy((rs.rand(B, T, 3, res, res) * 2 - 1).astype(np.float32)).to(dev) for _ in range(nbatch)]
    noises = [torch.from_numpy(rs.standard_normal((B, T, 7, 128)).astype(np.float32)).to(dev) for _ in range(nbatch)]
    with torch.no_grad():
        ref = _serial_reference(savi, roll, imgs, noises, T, H, PAIR_OPTS)
        ref_rec, ref_seg = [], []
        for j in range(nbatch):
            recon, _, masks, _ = savi.decode(ref[j][:, T:].reshape(B * H, 7, 128))
            ref_rec.append(recon.view(B, H, 3, res, res))
            ref_seg.append(postproc_mask(masks.view(B, H, 7, 1, res, res)))
        ref_rec, ref_seg = torch.stack(ref_rec), torch.stack(ref_seg)
        # the device postproc_mask = the reference formulation in torch ops on the same masks
        mk = masks.view(B, H, 7, 1, res, res)
        mc = mk.clone().reshape(B * H, 7, res * res)
        bg = mc.max(-1)[0].argmin(-1)
        weak = mc.max(1)[0] < 0.5
        isbg = torch.zeros(B * H, 7, dtype=torch.bool, device=dev)
        isbg[torch.arange(B * H, device=dev), bg] = True
        mc[isbg.unsqueeze(-1) & weak.unsqueeze(1)] = 1.
        assert torch.equal(postproc_mask(mk), mc.argmax(1).reshape(B, H, res, res))
        assert torch.equal(postproc_mask(mk.cpu()), postproc_mask(mk).cpu())
        for dtype in (torch.uint8, torch.int64):
            pipe = EncodeRolloutPipeline(savi, roll, B, T, H, decoder=savi, seg_dtype=dtype)
            for serial in (False, True):
                d = {}
                out = pipe.run(imgs, noises, serial=serial, decoded=d)
                torch.cuda.synchronize()
                assert torch.equal(out, ref)
                assert d['seg'].dtype == dtype and torch.equal(d['recon'], ref_rec) and torch.equal(d['seg'].long(), ref_seg), (dtype, serial)
            pipe.close()
        with pytest.raises(RuntimeError):
            EncodeRolloutPipeline(savi, roll, B, T, H).run(imgs[:1], noises[:1], decoded={})
        # the harness entry with a ragged tail
        vids = torch.cat(imgs, 0)[:B * nbatch - 1]
        nz = torch.cat(noises, 0)[:B * nbatch - 1]
        o, dd = harness.extract_and_rollout(savi, roll, vids, H, batch_size=B, noises=nz, decoder=savi)
        V = vids.shape[0]
        assert torch.equal(o, ref.reshape(-1, T + H, 7, 128)[:V])
        assert torch.equal(dd['recon'], ref_rec.reshape(-1, H, 3, res, res)[:V]) and torch.equal(dd['seg'].long(), ref_seg.reshape(-1, H, res, res)[:V])
        harness.release_pipelines()


@torch.no_grad()
def test_harness_edge_counts(dev):
    """harness.extract_and_rollout on the edge counts of a data set: no video at all (an empty result, no launch), ONE video (the serial tail only),
    one full batch + one video, and -- the plan of section 15 of the round-4 probes -- a batch count whose last two units take the remainder;
    every result equals the plain module calls video by video (the kernels are per video)."""
    from slotformer_amd import harness
    T, H, bs = 6, 3, 2
    savi, roll = _models(dev, gu.C2_SAVI, gu.C2_ROLL, seed=21)
    rs = np.random.RandomState(8)
    base = torch.from_numpy((rs.rand(5, T, 3, 128, 128) * 2 - 1).astype(np.float32)).to(dev)
    nzb = torch.from_numpy(rs.standard_normal((5, T, 7, 128)).astype(np.float32)).to(dev)
    try:
        empty = harness.extract_and_rollout(savi, roll, base[:0], H, batch_size=bs, noises=nzb[:0])
        assert tuple(empty.shape) == (0, T + H, 7, 128)
        ref = _serial_reference(savi, roll, [base[i:i + 1] for i in range(5)], [nzb[i:i + 1] for i in range(5)], T, H)[:, 0]   # [5, T + H, N, D]
        for V in (1, 3):
            out = harness.extract_and_rollout(savi, roll, base[:V], H, batch_size=bs, noises=nzb[:V])
            assert torch.equal(out, ref[:V]), V
        V = 2 * 13 + 1          # 13 full batches: units 4, 4, 5 (the remainder goes into the last units) + a ragged last video
        idx = torch.arange(V) % 5
        out = harness.extract_and_rollout(savi, roll, base[idx], H, batch_size=bs, noises=nzb[idx])
        assert torch.equal(out, ref[idx])
    finally:
        harness.release_pipelines()


@pytest.mark.parametrize('chain_on,nbatch', [('enc', 14), ('roll', 9)])
def test_pipeline_with_the_encode_in_two_halves(dev, chain_on, nbatch):
    """EncodeRolloutPipeline(split=True) (opt-in): image features of a batch on the encode lane (sf_savi_features_planes_f32), the slot branch of a whole rollout
    unit as ONE video-stationary launch (sf_savi_slots_chain_f32, csrc/slot_chain.hip) behind the features of its last batch (`enc`) or at the head of its
    rollout graph (`roll`).  Bit for bit with the serial schedule of the same object; 2e-5 from the default pipeline (split-bf16 rounding of the attention
    products: 5e-6 measured on the encoded frames); with injected kernel noise; a run that ends in a short unit."""
    from slotformer_amd.pipeline import EncodeRolloutPipeline
    B, T, H = 32, 6, 12
    savi, roll = _models(dev, gu.C2_SAVI, gu.C2_ROLL)
    rs = np.random.RandomState(11)
    imgs = [torch.from_numpy((rs.rand(B, T, 3, 128, 128) * 2 - 1).astype(np.float32)).to(dev) for _ in range(nbatch)]
    noises = [torch.from_numpy(rs.standard_normal((B, T, 7, 128)).astype(np.float32)).to(dev) for _ in range(nbatch)]
    with torch.no_grad():
        base = EncodeRolloutPipeline(savi, roll, B, T, H)
        assert not base.split
        ref = base.run(imgs, noises)
        torch.cuda.synchronize()
        base.close()
        pipe = EncodeRolloutPipeline(savi, roll, B, T, H, split=True, chain_on=chain_on)
        assert pipe.split and pipe.chain_on == chain_on and pipe.units[0].planes is not None
        out = pipe.run(imgs, noises)
        torch.cuda.synchronize()
        out3 = pipe.run(imgs, noises, serial=True)
        torch.cuda.synchronize()
        assert torch.equal(out, out3), (out - out3).abs().max().item()
        assert not torch.equal(out[:, :, :T], ref[:, :, :T])        # (another kernel form encoded the frames)
        e_enc = ((out[:, :, :T] - ref[:, :, :T]).abs().max() / ref[:, :, :T].abs().max()).item()
        assert e_enc < 2e-5, e_enc
        assert _close(out, ref, tol=5e-5), ((out - ref).abs().max() / ref.abs().max()).item()
        assert not torch.equal(out[0], out[1])
        out2 = pipe.run(imgs, noises)
        torch.cuda.synchronize()
        assert torch.equal(out, out2)
        pipe.close()


def test_sharded_run_at_32_videos_per_batch_agrees_to_rounding(dev):
    """at 32 videos per batch the default pipeline rolls FULL units out as token-stationary launches, whose last bits depend on where a video
    lands (position in a three-video workgroup, unit, run length).  A set of videos processed in one call and the same set processed as two shards
    (parallel.shard_range: what two ranks do) therefore agree to rounding, not bit for bit: 2e-5 asserted (~5e-6 measured over the horizon);
    the encoded frames are bit-identical (the encode does not depend on the grouping)."""
    from slotformer_amd import harness, parallel
    T, H, B = 6, 12, 32
    savi, roll = _models(dev, gu.C2_SAVI, gu.C2_ROLL)
    V = 14 * B
    rs = np.random.RandomState(23)
    videos = torch.from_numpy((rs.rand(V, T, 3, 128, 128) * 2 - 1).astype(np.float32)).to(dev)
    noises = torch.from_numpy(rs.standard_normal((V, T, 7, 128)).astype(np.float32)).to(dev)
    with torch.no_grad():
        whole = harness.extract_and_rollout(savi, roll, videos, H, batch_size=B, noises=noises)
        parts = []
        for rank in range(2):
            lo, hi = parallel.shard_range(V, rank, 2)
            parts.append(harness.extract_and_rollout(savi, roll, videos[lo:hi], H, batch_size=B, noises=noises[lo:hi]))
        sharded = torch.cat(parts, 0)
        torch.cuda.synchronize()
    assert torch.equal(whole[:, :T], sharded[:, :T])
    err = ((whole - sharded).abs().max() / whole.abs().max()).item()
    print(f'sharded vs single-process at 32 videos per batch: {err:.2e}')
    assert err <= 2e-5, err
    harness.release_pipelines()
