"""Software pipeline over independent batches of videos: slot extraction of later batches overlaps the rollout of earlier ones.

The two halves of the hot path have opposite shapes -- the SAVi encode is throughput work (convolutions, Slot Attention
over 4096 pixels), the SlotFormer rollout a chain of ~350 short dependent launches whose 128-168 workgroups mostly wait for
their weights -- so they run side by side on disjoint sets of CUs (streams created with CU masks, `sf_stream_create_cu_mask`
-> hipExtStreamCreateWithCUMask):

* the rollout of every slot buffer is captured ONCE into a hipGraph (one graph, one slot buffer and one workspace per batch
  in flight) and replayed on a rollout stream;
* partition 'pair' (default): a single rollout chain leaves most of its CUs idle most of the time, so TWO batches roll out
  side by side on two rollout streams that share CU rows 0-4 of all four shader engines of every XCD (160 CUs: 8.4 ms for two
  rollouts against 6.7 ms for one there), and the encode runs on rows 5-7 (96 CUs).  Four slot buffers / graphs.  The
  rollout kernels run in their throughput settings for these graphs (no seam launches, 64-row FFN workgroups).  Three busy
  CU-masked queues are the limit -- a fourth slows all of them (two encode lanes + two rollout streams: 8.9 ms per batch),
  five collapse (25 ms) -- hence two rollout streams and ONE encode stream;
* partition 'three': one rollout stream on CU rows 0-6 of shader engines 1-3 (168 CUs: what its widest launch needs, 21 per
  XCD) and two encode *lanes*, each with its share of a batch's videos: shader engine 0 (64 CUs, 3/4 of the videos) and row 7
  of shader engines 1-3 (24 CUs).  partition 'two' is the round-1 split (encode: shader engine 0, rollout: the other three);
* every mask gives each shader engine it touches the same number of CUs -- the rule for masks that do not unbalance the
  dispatch (encode_mask_words);
* work stealing: the CNN features of the first time steps of a batch do not depend on any slots, so the rollout stream that
  batch will roll out on computes them ahead of time, right after the rollout graph of an earlier batch (`engine.savi_cnn` ->
  `savi_encode(feat_pre=...)`); `steal_steps` may be fractional (1.25 = one step per batch, two for every fourth);
* fill and drain: the first encode of a run takes the whole chip (the calling stream) and is waited for on the host; while
  the second rollout stream is still idle it computes the stolen features of the next batches; the last rollout of a run
  goes to an unmasked stream; run() returns when the last batch is done (a wait left pending on the calling stream slows
  the masked queues).

Every batch still runs its complete encode + rollout; results are bit-identical to the serial
`savi({'img'}) -> rollout` sequence (tests/test_pipeline_gpu.py).  Reference caller shapes this replaces:
phyre_planning/test_phyre_planning.py:159-174 (encode -> pad -> rollout per batch), base_slots/extract_slots.py:19-38
followed by video_prediction/rollout_clevrer_slots.py:20-65.
"""
import ctypes as C
import os

import torch

from . import _lib, engine


def encode_mask_words(spec):
    """The 8 x 32-bit CU mask of the encode stream.  `spec`: 'rows<R>' = CU rows 0..R-1 of every shader engine of every XCD
    (32 R CUs; the rollout stream gets rows R..7), a sequence of 8 words, or one 32-bit word repeated 8 times (0xff = one
    whole shader engine per XCD, the round-1 mask).

    How the 256 mask bits reach the hardware (measured with tools/mask_probe.py, profiles/r02_probes.txt): bit b of word
    w is XCD b % 8, shader engine b // 8, CU row w.  Workgroups are dealt out in EQUAL shares to the XCDs and, inside an
    XCD, to every shader engine that has at least one CU enabled -- so the (XCD, shader engine) with the fewest enabled
    CUs sets the pace: [0xffff, 0xff x 7] (72 CUs: one extra CU in a second shader engine) runs the encode 3.7x SLOWER
    than 0xff x 8 (64 CUs).  A mask must give every shader engine it touches the same number of CUs, and so must its
    complement: whole shader engines (0xff x 8) or whole CU rows ('rows<R>').  An XCD whose mask is empty runs on all CUs."""
    if isinstance(spec, str) and spec.startswith('rows'):
        r = int(spec[4:])
        if not 1 <= r <= 7:
            raise ValueError('slotformer_amd: rows<R> needs 1 <= R <= 7')
        return [0xffffffff if w < r else 0 for w in range(8)]
    if isinstance(spec, str):
        spec = int(spec, 16)
    if isinstance(spec, (list, tuple)):
        if len(spec) != 8:
            raise ValueError('slotformer_amd: a CU mask has 8 words')
        return [int(w) & 0xffffffff for w in spec]
    return [int(spec) & 0xffffffff] * 8


# partition 'three' (word w = CU row w, byte s = shader engine s, all 8 XCDs alike)
ROLL_WORDS_3 = [0xffffff00] * 7 + [0]          # rows 0-6 of shader engines 1-3: 21 CUs per XCD
LANE0_WORDS_3 = [0x000000ff] * 8               # shader engine 0: 8 CUs per XCD
LANE1_WORDS_3 = [0] * 7 + [0xffffff00]         # row 7 of shader engines 1-3: 3 CUs per XCD
# partition 'pair': two rollout streams share CU rows 0-4 of all four shader engines, the encode gets rows 5-7
ROLL_WORDS_P = [0xffffffff] * 5 + [0] * 3      # 20 CUs per XCD
ENC_WORDS_P = [0] * 5 + [0xffffffff] * 3       # 12 CUs per XCD


class EncodeRolloutPipeline:
    """savi: StoSAVi container (eval, testing=True); rollouter: SlotRollouter / SingleStepSlotRollouter container.

    batch: videos per batch (fixed: the rollout graphs are captured for it); burn_in: encoded frames per video
    (= rollouter.history_len, or 1 for the single-step rollouter); pred_len: rollout steps.
    partition: 'pair' (default; see the module docstring), 'three', 'two' (one encode stream on `encode_cu_word`, the
    rollout on the complement) or 'none' (plain streams, shared CUs).  encode_cu_word: see encode_mask_words.
    steal_steps: time steps of convolutions per batch computed on the rollout stream (may be fractional: 1.25 = one step,
    two for every fourth batch); None = 0.75 for 'pair', 1 for 'two', 0 for 'three' (the rollout is the longer side there).
    """

    def __init__(self, savi, rollouter, batch, burn_in, pred_len, encode_cu_word=0xff, steal_steps=None, use_graph=True,
                 partition='pair'):
        self.savi, self.roll = savi, rollouter
        self.B, self.T, self.H = int(batch), int(burn_in), int(pred_len)
        p = next(rollouter.parameters())
        if not p.is_cuda:
            raise RuntimeError('slotformer_amd: the pipeline needs the models on a HIP device; there is no CPU fallback')
        self.dev = p.device
        self.N, self.D = rollouter.num_slots, rollouter.in_proj.in_features
        if partition not in ('pair', 'three', 'two', 'none'):
            raise ValueError("slotformer_amd: partition must be 'pair', 'three', 'two' or 'none'")
        if not encode_cu_word:
            partition = 'none'
        if partition == 'three' and self.B < 4:
            partition = 'two'
        # slot buffers / graphs / workspaces: one per batch in flight -- 'pair': two rolling out + one being encoded + one spare
        self.NB = 4 if partition == 'pair' else 2
        if steal_steps is None:
            steal_steps = {'pair': 0.75, 'two': 1, 'three': 0}.get(partition, 1)
        self.steal = max(0.0, min(float(steal_steps), float(self.T)))   # may be fractional: see _steal_of
        self._masked = []
        self._lib = _lib.lib()
        # 'pair': two chains share the rollout CUs -- seam launches (consumers spinning on a CU each) cost more than they save
        self.seam = None if partition != 'pair' else int(os.environ.get('SF_PIPE_SEAM', '0'))
        with torch.no_grad():
            self.bufs = [torch.zeros(self.B, self.T + self.H, self.N, self.D, device=self.dev) for _ in range(self.NB)]
            self.graphs = []
            for gi in range(self.NB):
                self._rollout_eager(gi)   # allocates its workspace
                torch.cuda.synchronize(self.dev)
                if use_graph:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._rollout_eager(gi)
                    self.graphs.append(g)
        self.cu_split = False
        self.partition = 'none'
        self.s_roll = None
        self.roll_streams = []           # rollout streams: batch j rolls out on roll_streams[j % len]
        self.lanes = []                  # encode lanes: (stream, first video, end video)
        if partition != 'none':
            try:
                if partition == 'pair':
                    self.roll_streams = [self._masked_stream(ROLL_WORDS_P), self._masked_stream(ROLL_WORDS_P)]
                    self.s_roll = self.roll_streams[0]
                    self.lanes = [(self._masked_stream(ENC_WORDS_P), 0, self.B)]
                    self.encode_cus, self.rollout_cus = 96, 160
                elif partition == 'three':
                    nb = max(1, round(self.B * 24 / 88))
                    self.s_roll = self._masked_stream(ROLL_WORDS_3)
                    self.lanes = [(self._masked_stream(LANE0_WORDS_3), 0, self.B - nb),
                                  (self._masked_stream(LANE1_WORDS_3), self.B - nb, self.B)]
                    self.encode_cus, self.rollout_cus = 88, 168
                else:
                    words = encode_mask_words(encode_cu_word)
                    self.s_roll = self._masked_stream([~w & 0xffffffff for w in words])
                    self.lanes = [(self._masked_stream(words), 0, self.B)]
                    self.encode_cus = sum(bin(w).count('1') for w in words)
                    self.rollout_cus = 256 - self.encode_cus
                self.cu_split = True
                self.partition = partition
            except RuntimeError:      # CU masking unavailable on this runtime: keep the pipeline, on shared CUs
                self.close()
                self.s_roll, self.lanes, self.roll_streams = None, [], []
        if self.s_roll is None:
            self.s_roll = torch.cuda.Stream(device=self.dev, priority=-1)
            self.lanes = [(torch.cuda.Stream(device=self.dev), 0, self.B)]
            self.encode_cus = self.rollout_cus = 256
        if not self.roll_streams:
            self.roll_streams = [self.s_roll]
        self.s_free = torch.cuda.Stream(device=self.dev) if len(self.roll_streams) > 1 else None   # unmasked: the drain
        self.s_enc = self.lanes[0][0]
        self.fill_whole_chip = True      # the first encode(s) of a run on the calling stream (all CUs)
        self.fill_batches = 1            # (2 = batch 1 on the whole chip as well: measured worse, 311 vs 323 k frames/s at 20 steps)
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
    def _steal_of(self, j):
        """time steps of convolutions stolen for batch j: integers that average to self.steal (1.25 -> 1, 1, 1, 2, ...)"""
        import math
        return int(math.floor(self.steal * (j + 1) + 1e-9) - math.floor(self.steal * j + 1e-9))

    def _rollout_eager(self, gi):
        if self.seam is None:
            engine.rollout(self.roll, self.bufs[gi], self.T, self.H, ws_slot=('pipe', gi))
            return
        # throughput mode of the rollout kernels: no seam launches, 64-row FFN workgroups
        old, old64 = self._lib.sf_get_seam_fused(), self._lib.sf_get_ffn_rows64()
        self._lib.sf_set_seam_fused(self.seam)
        self._lib.sf_set_ffn_rows64(int(os.environ.get('SF_PIPE_FFN64', '1')))
        try:
            engine.rollout(self.roll, self.bufs[gi], self.T, self.H, ws_slot=('pipe', gi))
        finally:
            self._lib.sf_set_seam_fused(old)
            self._lib.sf_set_ffn_rows64(old64)

    def _rollout(self, gi):
        if self.graphs:
            self.graphs[gi].replay()
        else:
            self._rollout_eager(gi)

    def _encode(self, img, noise, dst, feat_pre, lo=0, hi=None, lane=0):
        """videos [lo, hi) of one batch -> dst[lo:hi, :burn_in] on the current stream"""
        hi = self.B if hi is None else hi
        if lo != 0 or hi != self.B:
            img = img[lo:hi]
            noise = None if noise is None else noise[lo:hi]
        if noise is None and getattr(self.savi, 'kernel_dist_layer', None) is not None:
            # fresh eps ~ N(0,1) per frame, as the reference draws it (savi.py:363-365)
            noise = torch.randn(hi - lo, self.T, self.N, self.D, device=self.dev)
        post, _, _ = engine.savi_encode(self.savi, img, noise=noise, feat_pre=feat_pre, ws_slot=('pipe', lane))
        dst[lo:hi, :self.T].copy_(post)

    @torch.no_grad()
    def run(self, imgs, noises=None, out=None, serial=False):
        """imgs: sequence of n device tensors [B, burn_in, 3, H, W]; noises: None or n tensors [B, burn_in, N, D]
        (the kernel noise of every frame, for reproducible runs).  Returns out [n, B, burn_in + pred_len, N, D]; the
        pipelined schedule returns when the last batch is finished (the host waits for it, see the end of this function).
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
        lanes, rolls = self.lanes, self.roll_streams
        nl = len(lanes)
        # work stealing: the features of batch j are computed `lead` batches earlier, on the rollout stream of batch j - lead
        # (the same stream batch j will roll out on) right after that batch's rollout
        lead = 2 * len(rolls)
        kmax = int(-(-steal // 1))
        if steal and self.feat_bufs is None:
            self.feat_bufs = [[engine.savi_cnn(self.savi, imgs[0][lo:hi], 0, kmax, ws_slot=('pipe_steal', li)) for _ in range(lead)]
                              for li, (_, lo, hi) in enumerate(lanes)]
        for st, _, _ in lanes:
            st.wait_stream(cur)
        for st in rolls:
            st.wait_stream(cur)
        ev_enc = [[torch.cuda.Event() for _ in range(nl)] for _ in range(n)]
        ev_roll = [torch.cuda.Event(enable_timing=True) for _ in range(n)]   # also: completion time of every batch
        ev_pre = [torch.cuda.Event() for _ in range(n + lead)]
        for j in range(n):
            if j < self.fill_batches and self.cu_split and self.fill_whole_chip:
                # pipeline fill: the first encode takes the whole chip (the calling stream); the masked lanes start after it
                self._encode(imgs[j], nz(j), self.bufs[j % NB], None)
                ev_enc[j][0].record(cur)
                # the host waits for it: with the three other queues parked in a wait on this event the encode was measured
                # at 4.8 instead of 3.05 ms (a queue stalled in a cross-queue wait slows the queue that is running)
                ev_enc[j][0].synchronize()
                for st, _, _ in lanes:
                    st.wait_event(ev_enc[j][0])
                ev_wait = ev_enc[j][:1]
            else:
                for li, (st, lo, hi) in enumerate(lanes):
                    # (the first `lead` batches compute their own convolutions: stealing starts with batch `lead`, whose
                    #  features are produced after the rollout of batch 0)
                    kj = self._steal_of(j) if (steal and (j >= lead or (j >= 2 and len(rolls) > 1))) else 0
                    pre = self.feat_bufs[li][j % lead][:kj] if kj else None
                    with torch.cuda.stream(st):
                        if j >= NB:
                            st.wait_event(ev_roll[j - NB])   # slot buffer j % NB is free once batch j-NB has left it
                        if pre is not None:
                            st.wait_event(ev_pre[j])
                        self._encode(imgs[j], nz(j), self.bufs[j % NB], pre, lo, hi, li)
                        ev_enc[j][li].record(st)
                ev_wait = ev_enc[j]
            s_roll = rolls[j % len(rolls)]
            if len(rolls) > 1 and j == n - 1 and self.s_free is not None:
                # drain: the encode lane is idle from here on -- the last rollout takes an unmasked stream (all CUs) instead of
                # sharing the rollout partition with the one before it
                s_roll = self.s_free
                s_roll.wait_stream(rolls[j % len(rolls)])   # (order behind batch j - 2 on the stream it would have used)
            with torch.cuda.stream(s_roll):
                for e in ev_wait:
                    s_roll.wait_event(e)
                self._rollout(j % NB)
                out[j].copy_(self.bufs[j % NB])
                ev_roll[j].record(s_roll)
                if steal and j + lead < n:
                    # feature buffers (j + lead) % lead == j % lead were consumed by encode j, which this stream has waited for
                    kj = self._steal_of(j + lead)
                    for li, (_, lo, hi) in enumerate(lanes):
                        if kj:
                            engine.savi_cnn(self.savi, imgs[j + lead][lo:hi], 0, kj, out=self.feat_bufs[li][j % lead][:kj],
                                            ws_slot=('pipe_steal', li, j % len(rolls)))
                    ev_pre[j + lead].record(s_roll)
                if steal and j == 0 and len(rolls) > 1:
                    # fill: the second rollout stream idles until batch 1 is encoded -- it computes the features of batches
                    # 2 .. lead-1 now, so only batch 1 pays for its own convolutions
                    with torch.cuda.stream(rolls[1]):
                        for jj in range(2, min(lead, n)):
                            kj = self._steal_of(jj)
                            for li, (_, lo, hi) in enumerate(lanes):
                                if kj:
                                    engine.savi_cnn(self.savi, imgs[jj][lo:hi], 0, kj, out=self.feat_bufs[li][jj % lead][:kj],
                                                    ws_slot=('pipe_steal', li, 1))
                            ev_pre[jj].record(rolls[1])
        # The host waits for the last batch HERE, before the calling stream is made to wait for the pipeline's streams: a
        # wait that sits pending on the calling stream (PyTorch's default stream is the legacy null stream) for the whole
        # run was measured to slow the kernels of the masked encode lane that shares shader engines with the rollout by
        # 30 % (7.7 instead of 6.4 ms per batch, tools/lane_probe.py COPY=1) -- so run() returns when the results are done.
        for e in ev_roll[-len(rolls):]:
            e.synchronize()
        for st, _, _ in lanes:
            cur.wait_stream(st)
        for st in rolls + ([self.s_free] if self.s_free is not None else []):
            cur.wait_stream(st)
        self.completion_events = ev_roll
        return out
