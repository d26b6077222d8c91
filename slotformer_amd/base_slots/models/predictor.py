"""Slot transition modules t -> t+1 (reference: slotformer/base_slots/models/predictor.py).

These classes are *parameter containers* with the reference's constructor signatures and
state-dict keys; the arithmetic runs inside libslotformer_hip (engine.hip: predictor step).
"""
import torch.nn as nn


class Predictor(nn.Module):

    def burnin(self, x):
        pass

    def reset(self):
        pass


class TransformerPredictor(Predictor):
    """predictor.py:20-44: nn.TransformerEncoder over the N slots (batch_first)."""

    def __init__(self, d_model=128, num_layers=1, num_heads=4, ffn_dim=256, norm_first=True):
        super().__init__()
        layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=num_heads, dim_feedforward=ffn_dim,
                                           norm_first=norm_first, batch_first=True)
        self.transformer_encoder = nn.TransformerEncoder(encoder_layer=layer, num_layers=num_layers,
                                                         enable_nested_tensor=False)
        self.d_model, self.num_layers, self.num_heads = d_model, num_layers, num_heads
        self.ffn_dim, self.norm_first = ffn_dim, norm_first


class ResidualMLPPredictor(Predictor):
    """predictor.py:47-73: x = LN(x); out = MLP(x) + (x if norm_first else input)."""

    def __init__(self, channels, norm_first=True):
        super().__init__()
        assert len(channels) >= 2
        if len(channels) != 3:
            raise NotImplementedError('ResidualMLPPredictor: only [D, 2D, D] (the reference configuration)')
        self.ln = nn.LayerNorm(channels[0])
        mods = []
        for i in range(len(channels) - 2):
            mods += [nn.Linear(channels[i], channels[i + 1]), nn.ReLU()]
        mods.append(nn.Linear(channels[-2], channels[-1]))
        self.mlp = nn.Sequential(*mods)
        self.norm_first = norm_first


class RNNPredictorWrapper(Predictor):
    """predictor.py:76-135: base predictor -> single-step nn.LSTM -> Linear.

    `hidden_state` is (h, c), each [1, B*N, hidden] exactly as nn.LSTM keeps it; it persists
    across frames and across temporal chunks until reset() (savi.py:474-475).
    """

    def __init__(self, base_predictor, input_size=128, hidden_size=256, num_layers=1, rnn_cell='LSTM',
                 sg_every=None):
        super().__init__()
        if rnn_cell != 'LSTM' or num_layers != 1:
            raise NotImplementedError('only a 1-layer LSTM is supported (all reference configs)')
        self.base_predictor = base_predictor
        self.rnn = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)
        self.step = 0
        self.hidden_state = None
        self.out_projector = nn.Linear(hidden_size, input_size)
        self.sg_every = sg_every  # stop-gradient schedule: training-only, no effect on the forward values
        self.hidden_size = hidden_size

    def reset(self):
        self.step = 0
        self.hidden_state = None
