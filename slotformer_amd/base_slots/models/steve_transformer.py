"""STEVE's slot-conditioned Transformer decoder on the MI355X engine
(reference: slotformer/base_slots/models/steve_transformer.py, after singhgautam/slate).

The nn.Modules hold parameters under the reference's state-dict names; `STEVETransformerDecoder.forward/generate` run
on the HIP library: bias-free q|k|v projections as one GEMM per attention (LayerNorm fused as the GEMM prologue),
`sf_slate_attention_f32` (online-softmax attention, causal for the token stream, plain for the slots), output
projection with the residual fused in the GEMM epilogue, FFN as two GEMMs, `sf_embed_tokens_f32`,
`sf_argmax_rows_f32`.  Inference only."""
import torch
from torch import nn

from ... import ops


def linear(in_features, out_features, bias=True, weight_init='xavier', gain=1.):
    """Parameter holder with the reference's initialisers (steve_utils.py:151-166)."""
    m = nn.Linear(in_features, out_features, bias)
    if weight_init == 'kaiming':
        nn.init.kaiming_uniform_(m.weight, nonlinearity='relu')
    else:
        nn.init.xavier_uniform_(m.weight, gain)
    if bias:
        nn.init.zeros_(m.bias)
    return m


class MultiHeadAttention(nn.Module):
    """steve_transformer.py:12-55 (bias-free projections)."""

    def __init__(self, d_model, num_heads, dropout=0., gain=1.):
        super().__init__()
        assert d_model % num_heads == 0, 'd_model must be divisible by num_heads'
        self.d_model = d_model
        self.num_heads = num_heads
        self.attn_dropout = nn.Dropout(dropout)
        self.output_dropout = nn.Dropout(dropout)
        self.proj_q = linear(d_model, d_model, bias=False)
        self.proj_k = linear(d_model, d_model, bias=False)
        self.proj_v = linear(d_model, d_model, bias=False)
        self.proj_o = linear(d_model, d_model, bias=False, gain=gain)


class PositionalEncoding(nn.Module):
    """Learned position table (steve_transformer.py:58-74)."""

    def __init__(self, max_len, d_model, dropout=0.1):
        super().__init__()
        self.dropout = nn.Dropout(dropout)
        self.pe = nn.Parameter(torch.zeros(1, max_len, d_model), requires_grad=True)
        nn.init.trunc_normal_(self.pe)


class TransformerDecoderBlock(nn.Module):
    """steve_transformer.py:146-199."""

    def __init__(self, max_len, d_model, num_heads, dropout=0., gain=1., is_first=False):
        super().__init__()
        self.is_first = is_first
        self.self_attn_layer_norm = nn.LayerNorm(d_model)
        self.self_attn = MultiHeadAttention(d_model, num_heads, dropout, gain)
        mask = torch.triu(torch.ones((max_len, max_len), dtype=torch.bool), diagonal=1)
        self.self_attn_mask = nn.Parameter(mask, requires_grad=False)
        self.encoder_decoder_attn_layer_norm = nn.LayerNorm(d_model)
        self.encoder_decoder_attn = MultiHeadAttention(d_model, num_heads, dropout, gain)
        self.ffn_layer_norm = nn.LayerNorm(d_model)
        self.ffn = nn.Sequential(
            linear(d_model, 4 * d_model, weight_init='kaiming'), nn.ReLU(),
            linear(4 * d_model, d_model, gain=gain), nn.Dropout(dropout))


class TransformerDecoder(nn.Module):
    """steve_transformer.py:202-241."""

    def __init__(self, num_blocks, max_len, d_model, num_heads, dropout=0.):
        super().__init__()
        if num_blocks > 0:
            gain = (3 * num_blocks)**(-0.5)
            self.blocks = nn.ModuleList(
                [TransformerDecoderBlock(max_len, d_model, num_heads, dropout, gain, is_first=True)] +
                [TransformerDecoderBlock(max_len, d_model, num_heads, dropout, gain, is_first=False)
                 for _ in range(num_blocks - 1)])
        else:
            self.blocks = nn.ModuleList()
        self.layer_norm = nn.LayerNorm(d_model)


class STEVETransformerDecoder(nn.Module):
    """steve_transformer.py:244-333."""

    def __init__(self, vocab_size, d_model, n_head, max_len, num_slots, num_layers, dropout=0.1):
        super().__init__()
        self.max_len = max_len
        self.vocab_size = vocab_size
        self.num_slots = num_slots
        self.d_model = d_model
        self.n_head = n_head
        self.in_proj = nn.Linear(d_model, d_model)
        self.tok_emb = nn.Embedding(vocab_size + 1, d_model)
        self.pos_emb = PositionalEncoding(max_len + 1, d_model, dropout)
        self.tf_dec = TransformerDecoder(num_blocks=num_layers, max_len=max_len + 1, d_model=d_model, num_heads=n_head,
                                         dropout=dropout)
        self.head = nn.Linear(d_model, vocab_size, bias=False)
        self._cat = {}

    # ---- HIP compute ------------------------------------------------------------------------------------
    def _catw(self, name, *ws):
        """cat of projection weights (q|k|v or k|v), rebuilt when a weight changes."""
        key = tuple((w.data_ptr(), w._version) for w in ws)
        hit = self._cat.get(name)
        if hit is None or hit[0] != key:
            hit = (key, torch.cat([w.detach() for w in ws], 0).contiguous())
            self._cat[name] = hit
        return hit[1]

    def forward(self, slots, idx):
        """slots [B,N,d], idx int64 [B,t] (targets without the last token) -> logits [B,1+t,V]."""
        assert slots.shape[1] == self.num_slots
        B, T = idx.shape
        assert T <= self.max_len
        if torch.is_grad_enabled() and (slots.requires_grad or any(p.requires_grad for p in self.parameters())):
            from ... import train
            return train.slate_decoder_forward(self, slots, idx)   # chain of HIP-backed autograd nodes (row N1)
        if self.training:
            raise RuntimeError('slotformer_amd STEVETransformerDecoder: train() mode without autograd; use .eval() + torch.no_grad()')
        d, H = self.d_model, self.n_head
        mem = ops.linear(slots.contiguous(), self.in_proj.weight.detach(), self.in_proj.bias.detach())   # [B,N,d]
        bos = torch.full((B, 1), self.vocab_size, dtype=torch.int64, device=idx.device)
        tokens = torch.cat([bos, idx.to(torch.int64)], 1).contiguous()                                # [B,1+t]
        x = ops.embed_tokens(tokens, self.tok_emb.weight.detach(), self.pos_emb.pe.detach()[0])        # [B,L,d]
        for i, blk in enumerate(self.tf_dec.blocks):
            sa, ca = blk.self_attn, blk.encoder_decoder_attn
            ln1 = (blk.self_attn_layer_norm.weight.detach(), blk.self_attn_layer_norm.bias.detach())
            wqkv = self._catw(f'sa{i}', sa.proj_q.weight, sa.proj_k.weight, sa.proj_v.weight)
            if blk.is_first:   # the first block normalises its input IN PLACE (steve_transformer.py:186-190)
                x = ops.layernorm(x, *ln1)
                qkv = ops.linear(x, wqkv)
            else:
                qkv = ops.linear(x, wqkv, ln=ln1)
            att = ops.slate_attention(qkv, qkv, qkv, H, True, 0, d, 2 * d, d_model=d)
            x = ops.linear(att, sa.proj_o.weight.detach(), residual=x)
            ln2 = (blk.encoder_decoder_attn_layer_norm.weight.detach(), blk.encoder_decoder_attn_layer_norm.bias.detach())
            q = ops.linear(x, ca.proj_q.weight.detach(), ln=ln2)
            kv = ops.linear(mem, self._catw(f'ca{i}', ca.proj_k.weight, ca.proj_v.weight))             # [B,N,2d]
            att = ops.slate_attention(q, kv, kv, H, False, 0, 0, d, d_model=d)
            x = ops.linear(att, ca.proj_o.weight.detach(), residual=x)
            ln3 = (blk.ffn_layer_norm.weight.detach(), blk.ffn_layer_norm.bias.detach())
            hdn = ops.linear(x, blk.ffn[0].weight.detach(), blk.ffn[0].bias.detach(), ln=ln3, relu=True)
            x = ops.linear(hdn, blk.ffn[2].weight.detach(), blk.ffn[2].bias.detach(), residual=x)
        fin = self.tf_dec.layer_norm
        return ops.linear(x, self.head.weight.detach(), ln=(fin.weight.detach(), fin.bias.detach()))

    def _slate_plan(self):
        """sf_slate_decoder descriptor (device pointers of the weights + concatenated projections), rebuilt when a
        parameter changes."""
        import ctypes as C
        from ..._lib import sf_slate_block, sf_slate_decoder
        sig = tuple((p.data_ptr(), p._version) for p in self.parameters())
        plan = getattr(self, '_plan', None)
        if plan is not None and plan[0] == sig:
            return plan[1]
        keep = []

        def dp(t):
            t = t.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            keep.append(t)
            return t.data_ptr()

        blocks = (sf_slate_block * len(self.tf_dec.blocks))()
        for i, blk in enumerate(self.tf_dec.blocks):
            sa, ca, k = blk.self_attn, blk.encoder_decoder_attn, blocks[i]
            k.ln1_g, k.ln1_b = dp(blk.self_attn_layer_norm.weight), dp(blk.self_attn_layer_norm.bias)
            k.wqkv = dp(torch.cat([sa.proj_q.weight, sa.proj_k.weight, sa.proj_v.weight], 0))
            k.wo = dp(sa.proj_o.weight)
            k.ln2_g, k.ln2_b = dp(blk.encoder_decoder_attn_layer_norm.weight), dp(blk.encoder_decoder_attn_layer_norm.bias)
            k.wq_c, k.wo_c = dp(ca.proj_q.weight), dp(ca.proj_o.weight)
            k.wkv_c = dp(torch.cat([ca.proj_k.weight, ca.proj_v.weight], 0))
            k.ln3_g, k.ln3_b = dp(blk.ffn_layer_norm.weight), dp(blk.ffn_layer_norm.bias)
            k.w1, k.b1, k.w2, k.b2 = dp(blk.ffn[0].weight), dp(blk.ffn[0].bias), dp(blk.ffn[2].weight), dp(blk.ffn[2].bias)
            k.is_first = int(blk.is_first)
        m = sf_slate_decoder()
        m.d_model, m.num_heads, m.num_layers = self.d_model, self.n_head, len(self.tf_dec.blocks)
        m.vocab_size, m.num_slots, m.max_len = self.vocab_size, self.num_slots, self.max_len
        m.in_proj_w, m.in_proj_b = dp(self.in_proj.weight), dp(self.in_proj.bias)
        m.tok_emb, m.pos_emb = dp(self.tok_emb.weight), dp(self.pos_emb.pe[0])
        m.lnf_g, m.lnf_b = dp(self.tf_dec.layer_norm.weight), dp(self.tf_dec.layer_norm.bias)
        m.head_w = dp(self.head.weight)
        m.blocks = C.cast(blocks, C.POINTER(sf_slate_block))
        keep.append(blocks)
        self._plan = (sig, (m, keep))
        return self._plan[1]

    def generate_cached(self, slots, steps):
        """Greedy generation with a K/V cache (`sf_slate_generate_f32`): the same arithmetic as `generate(sample=False)`,
        one new token per step instead of re-running the prefix, the whole loop inside one C call; returns (tokens
        [B,steps] on device, logits [B,steps,V] on the CPU like the reference)."""
        import ctypes as C
        from ..._lib import check, lib
        if self.training or torch.is_grad_enabled():
            raise RuntimeError('slotformer_amd STEVETransformerDecoder is inference-only: .eval() + torch.no_grad()')
        if not (slots.is_cuda and slots.dtype == torch.float32):
            raise RuntimeError('generate_cached needs float32 slots on a HIP device; there is no CPU fallback')
        assert slots.shape[1] == self.num_slots and steps - 1 <= self.max_len
        slots = slots.contiguous()
        B = slots.shape[0]
        m, _ = self._slate_plan()
        tokens = torch.empty(B, steps, dtype=torch.int64, device=slots.device)
        logits = torch.empty(B, steps, self.vocab_size, dtype=torch.float32, device=slots.device)
        nb = lib().sf_slate_generate_workspace_bytes(C.byref(m), B, steps)
        ws = torch.empty(nb, dtype=torch.uint8, device=slots.device)
        check(lib().sf_slate_generate_f32(C.byref(m), slots.data_ptr(), B, steps, tokens.data_ptr(), logits.data_ptr(),
                                          ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream))
        return tokens, logits.cpu()

    def generate(self, slots, steps, sample=False, temperature=1.0):
        """Autoregressive generation (steve_transformer.py:305-333): the whole prefix is re-run every step, as in the reference;
        greedy (sample=False: argmax) or sampled (sample=True: one draw per step from softmax(logits / temperature) -- the softmax a HIP
        launch, the draw torch.multinomial on the device generator, as the reference draws it).  Returns (tokens [B,steps] on device,
        logits [B,steps,V] on the CPU like the reference)."""
        assert not self.training
        B = slots.shape[0]
        assert steps - 1 <= self.max_len
        idx_cond = torch.zeros((B, 0), dtype=torch.int64, device=slots.device)
        all_logits = []
        for _ in range(steps):
            logits = self.forward(slots, idx_cond)[:, -1].contiguous()   # [B,V]
            all_logits.append(logits.cpu())
            if sample:
                ix = torch.multinomial(ops.softmax_rows(logits, scale=1.0 / float(temperature)), num_samples=1)
            else:
                ix = ops.argmax_rows(logits).unsqueeze(1)                 # argmax of softmax(logits / T) = argmax of logits
            idx_cond = torch.cat((idx_cond, ix), dim=1)
        return idx_cond, torch.stack(all_logits, dim=1)
