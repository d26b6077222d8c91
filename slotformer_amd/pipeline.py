"""Software pipeline over independent batches of videos: slot extraction of batch i+1 overlaps the rollout of batch i.

The two halves of the hot path have opposite shapes -- the SAVi encode is throughput work (convolutions, Slot Attention
over 4096 pixels), the SlotFormer rollout a latency chain of short dependent kernels that cannot fill the chip at B = 32 --
so they run side by side on disjoint sets of CUs:

* the rollout of every slot buffer is captured ONCE into a hipGraph (one graph, one slot buffer and one workspace per batch
  in flight) and replayed on the *rollout stream*;
* the encode runs on the *encode stream*; both streams are created with CU masks (`sf_stream_create_cu_mask` ->
  hipExtStreamCreateWithCUMask): the encode gets one shader engine of every XCD (mask byte 0xff in every word = 64 CUs),
  the rollout the other three (192 CUs).  Measured in round 1 (profiles/r01_probes.txt): whole bytes, identical in all
  words, are the only masks that do not unbalance the shader engines;
* work stealing: the CNN features of the first `steal_steps` time steps of batch j+2 do not depend on any slots, so the
  rollout stream computes them on its larger partition after the rollout graph of batch j while it would otherwise idle,
  and the encode of batch j+2 skips those convolutions (`engine.savi_cnn` / `savi_encode(feat_pre=...)`);
* the first encode of a run takes the whole chip (the calling stream): nothing else is active yet.

Every batch still runs its complete encode + rollout; results are bit-identical to the serial
`savi({'img'}) -> rollout` sequence (tests/test_pipeline_gpu.py).  Reference caller shapes this replaces:
phyre_planning/test_phyre_planning.py:159-174 (encode -> pad -> rollout per batch), base_slots/extract_slots.py:19-38
followed by video_prediction/rollout_clevrer_slots.py:20-65.
"""
import ctypes as C

import torch

from . import _lib, engine


class EncodeRolloutPipeline:
    """savi: StoSAVi container (eval, testing=True); rollouter: SlotRollouter / SingleStepSlotRollouter container.

    batch: videos per batch (fixed: the rollout graphs are captured for it); burn_in: encoded frames per video
    (= rollouter.history_len, or 1 for the single-step rollouter); pred_len: rollout steps.
    encode_cu_word: 32-bit CU mask word of the encode stream, repeated for all 8 words (0 = no CU partition).
    """

    def __init__(self, savi, rollouter, batch, burn_in, pred_len, encode_cu_word=0xff, steal_steps=1, use_graph=True):
        self.savi, self.roll = savi, rollouter
        self.B, self.T, self.H = int(batch), int(burn_in), int(pred_len)
        p = next(rollouter.parameters())
        if not p.is_cuda:
            raise RuntimeError('slotformer_amd: the pipeline needs the models on a HIP device; there is no CPU fallback')
        self.dev = p.device
        self.N, self.D = rollouter.num_slots, rollouter.in_proj.in_features
        self.NB = 2                      # slot buffers / graphs / workspaces: one rolling out + one being encoded
        self.steal = max(0, min(int(steal_steps), self.T))
        self._masked = []
        self._lib = _lib.lib()
        with torch.no_grad():
            self.bufs = [torch.zeros(self.B, self.T + self.H, self.N, self.D, device=self.dev) for _ in range(self.NB)]
            self.graphs = []
            for gi in range(self.NB):
                engine.rollout(rollouter, self.bufs[gi], self.T, self.H, ws_slot=('pipe', gi))   # allocates its workspace
                torch.cuda.synchronize(self.dev)
                if use_graph:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        engine.rollout(rollouter, self.bufs[gi], self.T, self.H, ws_slot=('pipe', gi))
                    self.graphs.append(g)
        self.cu_split = False
        self.s_enc = self.s_roll = None
        if encode_cu_word:
            try:
                words = [encode_cu_word & 0xffffffff] * 8
                self.s_enc = self._masked_stream(words)
                self.s_roll = self._masked_stream([~w & 0xffffffff for w in words])
                self.cu_split = True
            except RuntimeError:      # CU masking unavailable on this runtime: keep the pipeline, on shared CUs
                self.s_enc = self.s_roll = None
        if self.s_enc is None:
            self.s_enc = torch.cuda.Stream(device=self.dev)
            self.s_roll = torch.cuda.Stream(device=self.dev, priority=-1)
        self.encode_cus = 8 * bin(encode_cu_word & 0xffffffff).count('1') if self.cu_split else 256
        self.feat_bufs = None
        self.completion_events = []

    def _masked_stream(self, words):
        arr = (C.c_uint * 8)(*words)
        h = C.c_void_p()
        _lib.check(self._lib.sf_stream_create_cu_mask(C.byref(h), arr, 8))
        self._masked.append(h)
        return torch.cuda.ExternalStream(h.value, device=self.dev)

    def close(self):
        for h in self._masked:
            self._lib.sf_stream_destroy(h)
        self._masked = []

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    # ------------------------------------------------------------------------------------------------------------
    def _rollout(self, gi):
        if self.graphs:
            self.graphs[gi].replay()
        else:
            engine.rollout(self.roll, self.bufs[gi], self.T, self.H, ws_slot=('pipe', gi))

    def _encode(self, img, noise, dst, feat_pre):
        if noise is None and getattr(self.savi, 'kernel_dist_layer', None) is not None:
            # fresh eps ~ N(0,1) per frame, as the reference draws it (savi.py:363-365)
            noise = torch.randn(self.B, self.T, self.N, self.D, device=self.dev)
        post, _, _ = engine.savi_encode(self.savi, img, noise=noise, feat_pre=feat_pre, ws_slot='pipe')
        dst[:, :self.T].copy_(post)

    @torch.no_grad()
    def run(self, imgs, noises=None, out=None, serial=False):
        """imgs: sequence of n device tensors [B, burn_in, 3, H, W]; noises: None or n tensors [B, burn_in, N, D]
        (the kernel noise of every frame, for reproducible runs).  Returns out [n, B, burn_in + pred_len, N, D].
        serial=True runs the same calls back to back on the calling stream (reference schedule for the tests)."""
        n = len(imgs)
        for im in imgs:
            if tuple(im.shape[:2]) != (self.B, self.T) or not im.is_cuda:
                raise RuntimeError(f'every batch must be a device tensor [{self.B},{self.T},3,H,W], got {tuple(im.shape)}')
        if out is None:
            out = torch.empty(n, self.B, self.T + self.H, self.N, self.D, device=self.dev)
        nz = (lambda j: None) if noises is None else (lambda j: noises[j])
        cur = torch.cuda.current_stream(self.dev)
        if serial or n == 0:
            for j in range(n):
                self._encode(imgs[j], nz(j), self.bufs[0], None)
                self._rollout(0)
                out[j].copy_(self.bufs[0])
            return out
        NB, steal = self.NB, self.steal
        s_enc, s_roll = self.s_enc, self.s_roll
        if steal and self.feat_bufs is None:
            self.feat_bufs = [engine.savi_cnn(self.savi, imgs[0], 0, steal, ws_slot='pipe_steal') for _ in range(2)]
        s_enc.wait_stream(cur)
        s_roll.wait_stream(cur)
        ev_enc = [torch.cuda.Event() for _ in range(n)]
        ev_roll = [torch.cuda.Event(enable_timing=True) for _ in range(n)]   # also: completion time of every batch
        ev_pre = [torch.cuda.Event() for _ in range(n + 2)]
        for j in range(n):
            # (the first two batches compute their own convolutions: stealing starts with batch 2, whose features are
            #  produced after the rollout of batch 0)
            pre = self.feat_bufs[j % 2] if (steal and j >= 2) else None
            if j == 0 and self.cu_split:
                # pipeline fill: the first encode takes the whole chip (the calling stream); the masked encode stream
                # starts after it
                self._encode(imgs[0], nz(0), self.bufs[0], None)
                ev_enc[0].record(cur)
                s_enc.wait_event(ev_enc[0])
            else:
                with torch.cuda.stream(s_enc):
                    if j >= NB:
                        s_enc.wait_event(ev_roll[j - NB])   # slot buffer j % NB is free once batch j-NB has left it
                    if pre is not None:
                        s_enc.wait_event(ev_pre[j])
                    self._encode(imgs[j], nz(j), self.bufs[j % NB], pre)
                    ev_enc[j].record(s_enc)
            with torch.cuda.stream(s_roll):
                s_roll.wait_event(ev_enc[j])
                self._rollout(j % NB)
                out[j].copy_(self.bufs[j % NB])
                ev_roll[j].record(s_roll)
                if steal and j + 2 < n:
                    # feature buffer (j+2) % 2 == j % 2 was consumed by encode j, which this stream has waited for
                    engine.savi_cnn(self.savi, imgs[j + 2], 0, steal, out=self.feat_bufs[j % 2], ws_slot='pipe_steal')
                    ev_pre[j + 2].record(s_roll)
        cur.wait_stream(s_enc)
        cur.wait_stream(s_roll)
        self.completion_events = ev_roll
        return out
