"""dVAE image tokenizer on the MI355X engine (reference: slotformer/base_slots/models/dVAE.py, steve_utils.py:73-126).

The modules below only hold parameters under the reference's state-dict names (`encoder.{i}.m.weight`,
`encoder.{i}.weight/bias`, `decoder.…`); `tokenize` / `detokenize` run on the HIP library: 1x1 convolutions and the
4x4/stride-4 patch embedding as GEMMs over channels-last pixels (`sf_linear_f32`), the 3x3 convolutions on the
implicit-GEMM conv (`sf_conv2d_nhwc_f32`), GroupNorm(1)+ReLU(+PixelShuffle) in `sf_groupnorm1_nhwc_f32`, the token
pick in `sf_argmax_rows_f32`.  In training (`forward` outside `testing`, row N1) the same kernels run as autograd nodes of
`slotformer_amd.train` (`sf_linear_bwd_f32`, `sf_groupnorm1_nhwc_bwd_f32`, `sf_softmax_rows_bwd_f32`)."""
import torch
from torch import nn

from ...nerv_compat import BaseModel
from ... import ops


def conv2d(in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, weight_init='xavier'):
    """Parameter holder with the reference's initialisers (steve_utils.py:73-97)."""
    m = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=bias)
    if weight_init == 'kaiming':
        nn.init.kaiming_uniform_(m.weight, nonlinearity='relu')
    else:
        nn.init.xavier_uniform_(m.weight)
    if bias:
        nn.init.zeros_(m.bias)
    return m


class Conv2dBlock(nn.Module):
    """bias-free conv -> GroupNorm(1 group, affine weight/bias) -> ReLU (steve_utils.py:100-126)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.m = conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=False, weight_init='kaiming')
        self.weight = nn.Parameter(torch.ones(out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))


class dVAE(BaseModel):

    def __init__(self, vocab_size, img_channels=3):
        super().__init__()
        self.vocab_size = vocab_size
        self.img_channels = img_channels
        self.tau = 1.
        # dVAE.py:24-50
        self.encoder = nn.Sequential(
            Conv2dBlock(img_channels, 64, 4, 4), *[Conv2dBlock(64, 64, 1, 1) for _ in range(6)],
            conv2d(64, vocab_size, 1))
        self.decoder = nn.Sequential(
            Conv2dBlock(vocab_size, 64, 1), Conv2dBlock(64, 64, 3, 1, 1), Conv2dBlock(64, 64, 1, 1),
            Conv2dBlock(64, 64, 1, 1), Conv2dBlock(64, 256, 1), nn.PixelShuffle(2),
            Conv2dBlock(64, 64, 3, 1, 1), Conv2dBlock(64, 64, 1, 1), Conv2dBlock(64, 64, 1, 1),
            Conv2dBlock(64, 256, 1), nn.PixelShuffle(2), conv2d(64, img_channels, 1))
        self.testing = False
        self._packed = {}

    # ---- HIP compute ------------------------------------------------------------------------------------
    def _w3x3(self, w):
        key = (w.data_ptr(), w._version)
        if key not in self._packed:
            self._packed = {key: None} if len(self._packed) > 8 else self._packed
            self._packed[key] = ops.pack_conv_weight(w.detach().contiguous())
        return self._packed[key]

    def _block(self, x, blk, pixel_shuffle=1):
        """x NHWC [F,H,W,Cin] -> Conv2dBlock(x) (optionally followed by PixelShuffle(2)), NHWC."""
        w = blk.m.weight.detach()
        if w.shape[2] == 1:
            y = ops.linear(x, w.reshape(w.shape[0], w.shape[1]).contiguous())
        else:
            y = ops.conv2d_nhwc(x, self._w3x3(w), None, relu=False)
        return ops.groupnorm1_nhwc(y, blk.weight.detach(), blk.bias.detach(), relu=True, pixel_shuffle=pixel_shuffle)

    def _logits_nhwc(self, imgs):
        """imgs [F,3,H,W] -> logits [F,h,w,V] (channels-last)."""
        if torch.is_grad_enabled():
            raise RuntimeError('slotformer_amd dVAE.tokenize is a hard argmax: call it under torch.no_grad()')
        F_, C_, H, W = imgs.shape
        h, w = H // 4, W // 4
        # 4x4 / stride-4 convolution = GEMM over (c, ky, kx) patch vectors (dVAE.py:26)
        x = imgs.reshape(F_, C_, h, 4, w, 4).permute(0, 2, 4, 1, 3, 5).reshape(F_, h, w, C_ * 16).contiguous()
        b0 = self.encoder[0]
        y = ops.linear(x, b0.m.weight.detach().reshape(64, C_ * 16).contiguous())
        x = ops.groupnorm1_nhwc(y, b0.weight.detach(), b0.bias.detach(), relu=True)
        for i in range(1, 7):
            x = self._block(x, self.encoder[i])
        last = self.encoder[7]
        return ops.linear(x, last.weight.detach().reshape(self.vocab_size, 64).contiguous(), last.bias.detach())

    def tokenize(self, imgs, one_hot=True):
        """dVAE.py:52-77: one-hot map [B,(T,)V,h,w] or token ids [B,(T,)h,w]."""
        B = imgs.shape[0]
        unflatten = imgs.dim() == 5
        if unflatten:
            imgs = imgs.flatten(0, 1)
        logits = self._logits_nhwc(imgs.contiguous())
        idx = ops.argmax_rows(logits)                      # [F,h,w]
        if one_hot:
            F_, h, w, V = logits.shape
            z = torch.zeros(F_, V, h, w, device=logits.device, dtype=logits.dtype).scatter_(1, idx.unsqueeze(1), 1.)
        else:
            z = idx
        return z.unflatten(0, (B, -1)) if unflatten else z

    def detokenize(self, z):
        """dVAE.py:79-100: z [B,(T,)V,h,w] (probabilities over the vocabulary) -> image [B,(T,)3,4h,4w]."""
        assert z.shape[-3] == self.vocab_size
        B = z.shape[0]
        unflatten = z.dim() == 5
        if unflatten:
            z = z.flatten(0, 1)
        if torch.is_grad_enabled():   # under autograd: the same arithmetic as a chain of differentiable nodes
            recon = self.decode_nhwc(z.permute(0, 2, 3, 1).float()).permute(0, 3, 1, 2)
            return recon.unflatten(0, (B, -1)) if unflatten else recon
        d = self.decoder
        x = z.permute(0, 2, 3, 1).contiguous().float()
        x = self._block(x, d[0])
        x = self._block(x, d[1])
        x = self._block(x, d[2])
        x = self._block(x, d[3])
        x = self._block(x, d[4], pixel_shuffle=2)
        x = self._block(x, d[6])
        x = self._block(x, d[7])
        x = self._block(x, d[8])
        x = self._block(x, d[9], pixel_shuffle=2)
        last = d[11]
        x = ops.linear(x, last.weight.detach().reshape(self.img_channels, 64).contiguous(), last.bias.detach())
        recon = x.permute(0, 3, 1, 2).contiguous()
        return recon.unflatten(0, (B, -1)) if unflatten else recon

    # ---- training forward (dVAE.py:113-139) ---------------------------------------------------------------
    def _block_t(self, x, blk, pixel_shuffle=1):
        from ... import train
        w = blk.m.weight
        y = train.linear_weight(x, w.reshape(w.shape[0], w.shape[1])) if w.shape[2] == 1 else train.conv3x3_nhwc(x, w)
        return train.groupnorm1(y, blk.weight, blk.bias, relu=True, pixel_shuffle=pixel_shuffle)

    def decode_nhwc(self, z):
        """The decoder stack (dVAE.py:36-50) on a channels-last token map z [F,h,w,V] as autograd nodes -> image [F,4h,4w,3]
        channels-last.  Also what STEVE's Gumbel image loss differentiates through with the dVAE frozen (steve.py:327-335)."""
        from ... import train
        d = self.decoder
        y = self._block_t(z, d[0])
        y = self._block_t(y, d[1])
        y = self._block_t(y, d[2])
        y = self._block_t(y, d[3])
        y = self._block_t(y, d[4], pixel_shuffle=2)
        y = self._block_t(y, d[6])
        y = self._block_t(y, d[7])
        y = self._block_t(y, d[8])
        y = self._block_t(y, d[9], pixel_shuffle=2)
        return train.linear_weight(y, d[11].weight.reshape(self.img_channels, 64), d[11].bias)

    def forward(self, data_dict):
        """`img` [B,(T,)3,H,W]; optional `gumbel_tau`, `hard` as in the reference, and `gumbel` [B,(T,)V,h,w]: the Gumbel noise
        to use instead of a fresh draw (the reference draws it inside steve_utils.gumbel_softmax)."""
        if self.testing:
            return self.tokenize(data_dict['img'], one_hot=False)
        from ... import train
        x = data_dict['img']
        if not x.is_cuda:
            raise RuntimeError('slotformer_amd: inputs must live on a HIP device; there is no CPU fallback')
        tau = data_dict.get('gumbel_tau', self.tau)
        hard = data_dict.get('hard', False)
        B = x.shape[0]
        unflatten = x.dim() == 5
        if unflatten:
            x = x.flatten(0, 1)
        F_, C_, H, W = x.shape
        h, w = H // 4, W // 4
        e = self.encoder
        # encoder: 4x4 / stride-4 patch embedding as a GEMM over (c, ky, kx) patch vectors, six 1x1 blocks, logits
        p = x.float().reshape(F_, C_, h, 4, w, 4).permute(0, 2, 4, 1, 3, 5).reshape(F_, h, w, C_ * 16)
        y = train.linear_weight(p, e[0].m.weight.reshape(64, C_ * 16))
        y = train.groupnorm1(y, e[0].weight, e[0].bias, relu=True)
        for i in range(1, 7):
            y = self._block_t(y, e[i])
        logits = train.linear_weight(y, e[7].weight.reshape(self.vocab_size, 64), e[7].bias)     # [F,h,w,V]
        # relaxed sample of the token map
        g = data_dict.get('gumbel')
        if g is not None:
            g = (g.flatten(0, 1) if unflatten else g).permute(0, 2, 3, 1).to(logits.device).float().contiguous()
        z = train.gumbel_softmax(logits, g, tau, hard)   # without `gumbel`: noise generated inside the softmax kernel
        recon = self.decode_nhwc(z).permute(0, 3, 1, 2)
        # log-probabilities of the token map: reported, not part of the loss
        z_logits = ops.log_softmax_rows(logits.detach()).permute(0, 3, 1, 2)
        if unflatten:
            recon, z_logits = recon.unflatten(0, (B, -1)), z_logits.unflatten(0, (B, -1))
        return {'recon': recon, 'z_logits': z_logits}

    def calc_train_loss(self, data_dict, out_dict):
        """dVAE.py:141-146."""
        from ...host import losses
        return {'recon_loss': losses.image_recon_loss(out_dict['recon'], data_dict['img'])}

    @property
    def dtype(self):
        return self.encoder[-1].weight.dtype

    @property
    def device(self):
        return self.encoder[-1].weight.device
