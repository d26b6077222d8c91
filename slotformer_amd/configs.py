"""Model configurations of SURVEY.md 8 (C1..C5) in the dict form `build_model(params)` consumes -- the values of the
reference's `*_params.py` files (slotformer/base_slots/configs, slotformer/video_prediction/configs) with the documented
BASELINE.json overrides (resolution 128x128 for C2, the 6+50 / 6+40 / 1+80 horizons); tests/test_reference_configs.py
checks every entry against the reference files.  Used by bench.py, the tools and (re-exported through
tests/golden_util.py) the parity tests."""


# ---------------------------------------------------------------------------
# model configs (dict form of the reference's *_params.py)
# ---------------------------------------------------------------------------
def savi_cfg(res, num_slots, slot_size=128, mlp=256, iters=2, kernel_mlp=True, pred='transformer',
             rnn=True, kld='none', enc_out=128, pred_layers=2, pred_heads=4, pred_ffn=512,
             dec_res=(8, 8)):
    return dict(
        model='StoSAVi',
        resolution=(res, res),
        input_frames=6,
        slot_dict=dict(num_slots=num_slots, slot_size=slot_size, slot_mlp_size=mlp,
                       num_iterations=iters, kernel_mlp=kernel_mlp),
        enc_dict=dict(enc_channels=(3, 64, 64, 64, 64), enc_ks=5, enc_out_channels=enc_out,
                      enc_norm=''),
        dec_dict=dict(dec_channels=(slot_size, 64, 64, 64, 64), dec_resolution=dec_res, dec_ks=5,
                      dec_norm=''),
        pred_dict=dict(pred_type=pred, pred_rnn=rnn, pred_norm_first=True,
                       pred_num_layers=pred_layers, pred_num_heads=pred_heads,
                       pred_ffn_dim=pred_ffn, pred_sg_every=None),
        loss_dict=dict(use_post_recon_loss=True, kld_method=kld),
    )


def rollout_cfg(num_slots, slot_size, hist, d_model, layers, heads, ffn, cond_len=None,
                rollout_len=10, model='SlotFormer', res=64):
    rd = dict(num_slots=num_slots, slot_size=slot_size, history_len=hist, t_pe='sin', slots_pe='',
              d_model=d_model, num_layers=layers, num_heads=heads, ffn_dim=ffn, norm_first=True)
    if cond_len is not None:
        rd['cond_len'] = cond_len
    return dict(
        model=model,
        resolution=(res, res),
        input_frames=hist,
        slot_dict=dict(num_slots=num_slots, slot_size=slot_size),
        rollout_dict=rd,
        dec_dict=dict(dec_channels=(slot_size, 64, 64, 64, 64), dec_resolution=(8, 8), dec_ks=5,
                      dec_norm='', dec_ckp_path=''),
        loss_dict=dict(rollout_len=rollout_len, use_img_recon_loss=False),
    )


# C1: OBJ3D SAVi (savi_obj3d_params.py) -- deterministic, Transformer+LSTM predictor
C1_SAVI = savi_cfg(64, 6, iters=2, kernel_mlp=True, pred='transformer', rnn=True, kld='none')
C1_SAVI_IT3 = savi_cfg(64, 6, iters=3, kernel_mlp=True, pred='transformer', rnn=True, kld='none')
# C2: CLEVRER StoSAVi (stosavi_clevrer_params.py) at 128x128 -- stochastic, MLP predictor
C2_SAVI = savi_cfg(128, 7, iters=2, kernel_mlp=False, pred='mlp', rnn=False, kld='var-0.01')
# STEVE with its image side (dVAE tokens + Transformer decoder, row N2): a reduced steve_physion_params.py
def steve_tokens_cfg():
    cfg = savi_cfg(64, 4, slot_size=64, mlp=128, iters=2, pred='transformer', rnn=True, kld='none', enc_out=64,
                   pred_layers=1, pred_heads=4, pred_ffn=128)
    cfg['model'] = 'STEVE'
    cfg['dvae_dict'] = dict(down_factor=4, vocab_size=64, dvae_ckp_path='')
    cfg['dec_dict'] = dict(dec_type='slate', dec_num_layers=2, dec_num_heads=4, dec_d_model=64)
    cfg['loss_dict'] = dict(use_img_recon_loss=False)
    return cfg


def steve_slotformer_cfg():
    """A reduced slotformer_physion_params.py: STEVESlotFormer on the steve_tokens_cfg() STEVE."""
    sc = steve_tokens_cfg()
    return dict(
        model='STEVESlotFormer', resolution=sc['resolution'], input_frames=3,
        slot_dict=dict(num_slots=4, slot_size=64), dvae_dict=dict(sc['dvae_dict']),
        dec_dict=dict(dec_num_layers=2, dec_num_heads=4, dec_d_model=64, dec_ckp_path=''),
        rollout_dict=dict(num_slots=4, slot_size=64, history_len=3, t_pe='sin', slots_pe='', d_model=64, num_layers=2,
                          num_heads=4, ffn_dim=256, norm_first=True),
        loss_dict=dict(rollout_len=2, use_img_recon_loss=True))


# C4: Physion STEVE encoder side (steve_physion_params.py)
C4_STEVE = savi_cfg(128, 6, slot_size=192, mlp=384, iters=2, pred='transformer', rnn=True,
                    enc_out=192, pred_ffn=768)
C4_STEVE['model'] = 'STEVE'
# C5: PHYRE SAVi (savi_phyre_params-fold0.py)
C5_SAVI = savi_cfg(128, 8, iters=2, kernel_mlp=True, pred='transformer', rnn=True, kld='none',
                   dec_res=(16, 16))

C1_ROLL = rollout_cfg(6, 128, 6, 128, 4, 8, 512, rollout_len=10)
C2_ROLL = rollout_cfg(7, 128, 6, 256, 4, 8, 1024, rollout_len=50)
C4_ROLL = rollout_cfg(6, 192, 6, 256, 8, 8, 1024, rollout_len=40)
C4_ROLL_REF = rollout_cfg(6, 192, 15, 256, 8, 8, 1024, rollout_len=10)
C5_ROLL = rollout_cfg(8, 128, 1, 256, 8, 8, 1024, cond_len=6, rollout_len=80,
                      model='SingleStepSlotFormer', res=128)
# row N1 (training of StoSAVi itself): stosavi_clevrer_params.py at its own 64x64 training resolution
TRAIN_SAVI = savi_cfg(64, 7, iters=2, kernel_mlp=False, pred='mlp', rnn=False, kld='var-0.01')
# row N1 (training): a reduced slotformer_clevrer_params.py whose gradients fit a small fixture
TRAIN_ROLL = rollout_cfg(3, 64, 3, 64, 2, 2, 128, rollout_len=3)
TRAIN_ROLL_IMG = rollout_cfg(3, 64, 3, 64, 2, 2, 128, rollout_len=2)   # with the image term (use_img_recon_loss=True)
TRAIN_ROLL_IMG['loss_dict'] = dict(rollout_len=2, use_img_recon_loss=True)
# trained position tables (build_pos_enc 'learnable', slotformer.py:19-29): no shipped config uses them, the constructor accepts them
TRAIN_ROLL_PE = rollout_cfg(3, 64, 3, 64, 2, 2, 128, rollout_len=3)
TRAIN_ROLL_PE['rollout_dict'].update(t_pe='learnable', slots_pe='learnable')


class ParamsView:
    """Attribute view of a cfg dict -- what ``build_model(params)`` consumes."""

    def __init__(self, cfg):
        self.__dict__.update(cfg)

    def get(self, k, default=None):
        return self.__dict__.get(k, default)
