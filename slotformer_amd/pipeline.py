"""Software pipeline over independent batches of videos: slot extraction of later batches overlaps the rollout of earlier ones.

The two halves of the hot path have opposite shapes -- the SAVi encode is throughput work (convolutions, Slot Attention
over 4096 pixels), the SlotFormer rollout a chain of ~350 short dependent launches whose workgroups mostly wait for their
weights -- so they run side by side on disjoint sets of CUs (streams created with CU masks, `sf_stream_create_cu_mask`
-> hipExtStreamCreateWithCUMask):

* a rollout UNIT is `group` consecutive batches rolled out by ONE hipGraph over one slot buffer [group * B, T, N, D]:
  every launch of the chain then covers group * B * L rows, so the weights a workgroup drags through its CU, the launch
  gaps and the first-load latencies are paid once per `group` batches (round 3: group = 4 with all-heads attention and
  128-row FFN workgroups; the rollout kernels are per-video / per-row, so the results do not depend on how videos are
  grouped -- tested).  Every unit buffer has its own graph and workspace, captured ONCE; a ragged last unit (fewer batches)
  gets its own graph; the encode of a batch replays from a hipGraph over fixed input buffers as well;
* partition 'pair' (default): a single rollout chain leaves most of its CUs idle most of the time, so TWO units roll out
  side by side on two rollout streams that share whole CU rows of all four shader engines of every XCD, and the encode
  runs on the other rows: 128 / 128 CUs for balanced model pairs, fewer encode rows for rollout-heavy ones (_encode_rows).
  Three busy CU-masked queues are the limit on this platform -- a fourth slows all of them, five collapse
  (profiles/r02_probes.txt) -- hence two rollout streams and ONE encode stream;
* partition 'three': one rollout stream on CU rows 0-6 of shader engines 1-3 (168 CUs) and two encode *lanes*, each with
  its share of a batch's videos: shader engine 0 (64 CUs, 3/4 of the videos) and row 7 of shader engines 1-3 (24 CUs).
  partition 'two' is the round-1 split (encode: `encode_cu_word`, rollout: the complement);
* every mask gives each shader engine it touches the same number of CUs -- the rule for masks that do not unbalance the
  dispatch (encode_mask_words);
* work stealing: the CNN features of the first time steps of a batch do not depend on any slots, so a rollout stream
  computes them ahead of time, right after the rollout graph of an earlier unit (`engine.savi_cnn` ->
  `savi_encode(feat_pre=...)`); `steal_steps` may be fractional (1.25 = one step per batch, two for every fourth);
* fill and drain: the first encodes of a run take the whole chip (the calling stream) and are waited for on the host;
  while the rollout streams are still idle they compute the stolen features of the next batches; the last unit of a run
  goes to an unmasked stream; run() returns when the last batch is done (a wait left pending on the calling stream slows
  the masked queues).

Every batch still runs its complete encode + rollout; results are bit-identical to the serial
`savi({'img'}) -> rollout` sequence (tests/test_pipeline_gpu.py).  Reference caller shapes this replaces:
phyre_planning/test_phyre_planning.py:159-174 (encode -> pad -> rollout per batch), base_slots/extract_slots.py:19-38
followed by video_prediction/rollout_clevrer_slots.py:20-65.

Ownership: the captured graphs hold raw pointers into the slot buffers, the workspaces and the packed weight copies of
the rollouter's plan.  The pipeline therefore owns private workspaces (slot keys carrying id(self); freed by close()),
keeps the plan alive, and re-captures when the rollouter's parameters changed since the capture (run() compares the
plan signature).
"""
import ctypes as C
import math
import os
import threading
import time

import torch

from . import _lib, engine

_STREAMS = {}                      # process-wide stream pool (EncodeRolloutPipeline._masked_stream / _pool_stream)
_STREAMS_LOCK = threading.Lock()
_POOL = True                        # one process-wide pool of streams (private streams per pipeline object share hardware queues: 405 vs 455 k frames/s)


def encode_mask_words(spec):
    """The 8 x 32-bit CU mask of the encode stream.  `spec`: 'rows<R>' = CU rows 0..R-1 of every shader engine of every XCD
    (32 R CUs; the rollout streams get rows R..7), a sequence of 8 words, or one 32-bit word repeated 8 times (0xff = one
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
# partition 'pair': two rollout streams share CU rows 0-3 of all four shader engines, the encode gets rows 4-7
ROLL_WORDS_P = [0xffffffff] * 4 + [0] * 4      # 16 CUs per XCD
ENC_WORDS_P = [0] * 4 + [0xffffffff] * 4       # 16 CUs per XCD


class _Unit:
    """One rollout unit: slot buffer [nb * B, T + H, N, D], its graph (None: eager) and the workspace slot key."""

    def __init__(self, buf, key):
        self.buf, self.key, self.graph = buf, key, None
        self.busy = None   # event of the last rollout + copy-out that used the buffer in the current run()
        self.latency_form = False   # captured with the kernels' latency forms (the drain unit of a run: alone on the whole chip)
        self.row_form = False       # captured with the row-tile forms instead of the token-stationary launches (a drain unit: latency counts)
        self.planes = None          # split encode: the Slot-Attention inputs of the unit's batches ([nb][T][B][4096] rows of 512 B, uint8)
        self.noise = None           # split encode: the kernel noise of the unit's videos [nb * B, T, N, D] (None: the model samples nothing)


def encode_group_for(batch, n_batches):
    """How many consecutive batches a caller should hand to the pipeline as ONE pipeline batch (harness.extract_and_rollout, bench.py): the slot
    branch of an encode (predictor, Slot-Attention iterations, slot updates) is a chain of latency-bound launches that costs the same for 16 videos
    as for 32 -- C4 (16 videos per batch): 3.3 ms per batch on the encode lane one at a time, 2.5 two at a time; with units of 160 videos 199 ->
    253 k frames/s at 20 batches (profiles/r04_probes.txt section 8).  The largest E <= 32 // batch that divides the run and leaves >= 8 pipeline batches."""
    for e in range(max(1, 32 // max(int(batch), 1)), 1, -1):
        if n_batches % e == 0 and n_batches // e >= 8:
            return e
    return 1


def tok_unit_batches(rollouter, batch, burn_in=None, n_batches=None):
    """Batches per rollout unit when the layers before the last run as token-stationary launches (csrc/layer_tok.hip): a 128-token workgroup owns
    128 // L whole videos and needs a CU to itself, so a unit should bring 64 workgroups -- two units side by side then fill the 128 CUs of the rollout
    partition without queueing behind each other (C2: 6 batches of 32 videos = 64 workgroups; 592 k frames/s at 60 batches against 528 k with units of
    4).  None when the rollouter cannot take that form, or the unit would stay below 96 videos (small batches: the latency forms).  n_batches: pipeline
    batches of the run, where known -- models of more than four layers take the form in runs of three units or more only."""
    from . import _lib as _l
    if os.environ.get('SF_PIPE_TOK', '1') == '0' or not next(rollouter.parameters()).is_cuda:
        return None
    if not _l.lib().sf_rollout_tok_ok(C.byref(engine.rollouter_plan(rollouter).struct)):
        return None
    # Measured where it pays (profiles/r05_probes.txt): C2 (4 layers, 32 videos per batch) 497 -> 550 k frames/s at 20 batches, 528 -> 580 k at 60.
    # Not C5 (growing window of the single-step rollouter: 8-token windows leave a 128-token workgroup 16 videos, 80 rollout-bound steps: 450 -> 280 k).
    # C4 (8 layers; two 16-video batches per encode): 250 -> 240 k at 20 batches (two units), but 257 -> 299 k at 40 and 262 -> 318 k at 80: the longer
    # unit (8 layers x 118 us per step) needs a run of three units or more to pay.
    if hasattr(rollouter, 'cond_len') or int(batch) < 24:
        return None
    hist = getattr(rollouter, 'history_len', burn_in or 1)
    vpw = 128 // max(int(rollouter.num_slots) * int(hist), 1)
    g = max(1, min(8, (64 * vpw) // max(int(batch), 1)))
    if g * int(batch) < 96:
        return None
    if len(rollouter.transformer_encoder.layers) > 4 and (n_batches is None or int(n_batches) < 3 * g):
        return None
    return g


def unit_batches_for(rollouter, batch, n_batches, burn_in=None):
    """Batches per rollout unit for a run of n_batches (None = the constructor's default of 4).  A unit's row-tile launches should fill ONE
    round of the 64 CUs a rollout stream gets with 64-row tiles: 4096 token rows.  C2 (1344 rows per batch) and C5 (3072) stay at 4;
    C4 (16 videos x 36 tokens = 576 rows per batch) takes 7 -- 63 tiles per launch instead of 36: 231 vs 209 k frames/s at 84 batches,
    227 at 42 -- but only in runs of five units or more: at 20 batches the ragged last unit and the longer drain cost more (170 vs 199 k;
    `profiles/r03_probes.txt` section 18)."""
    gt = tok_unit_batches(rollouter, batch, burn_in, n_batches)
    if gt is not None:
        return gt
    hist = getattr(rollouter, 'cond_len', None) or getattr(rollouter, 'history_len', burn_in or 1)
    rows = int(batch) * int(rollouter.num_slots) * int(hist)
    g = max(4, min(8, 4096 // max(rows, 1)))
    from . import _lib as _l   # (units of several batches need the fused-layer path: its results do not depend on the batch size)
    fused = lambda: bool(_l.lib().sf_rollout_is_fused(C.byref(engine.rollouter_plan(rollouter).struct)))  # noqa: E731
    if g > 4 and n_batches >= 5 * g:
        return g if fused() else None
    # SHORT runs of small batches (round 4): a rollout launch of such a unit is latency-bound -- a C4 unit of 64 videos (36 row tiles) and one of 128
    # take the same 23.9 ms -- so the run is cut into an EVEN number (two rollout streams) of equal units of up to 8192 token rows: C4 at 20 batches of
    # 16: units of 4 / 5 / 10 batches 199 / 196 / 228 k frames/s (253 k with two batches per encode, encode_group_for above); C2 (5376 rows in its default
    # unit) and C5 keep 4: 493 / 490 k with 5 / 10 against 500+, 405 / 447 k against 445 (profiles/r04_probes.txt section 8)
    if 4 * rows < 5376:
        cands = [G for G in range(5, n_batches // 2 + 1) if n_batches % G == 0 and (n_batches // G) % 2 == 0 and G * rows <= 8192]
        if cands:
            return max(cands) if fused() else None
        # no even split into equal units (42 C4 batches = 21 encodes of 32 videos): the smallest even number of units of <= 8192 rows with a shorter
        # last one (21 encodes: 6, 6, 6, 3) instead of falling back to units of 4 (C4 at 42 batches: 238 k frames/s with units of 4)
        gmax = 8192 // max(rows, 1)
        if gmax >= 5 and n_batches >= 10:
            units = 2 * -(-n_batches // (2 * gmax))
            G = -(-n_batches // units)
            if G >= 5:
                return G if fused() else None
    return None


def pair_unit_options(rollouter, batch, group, roll_cus=128, tok=False, burn_in=None):
    """The per-call kernel options (engine.rollout_opts keys) of the FULL rollout units of a 'pair' pipeline: `group` batches of `batch` videos per unit,
    `roll_cus` CUs for the two rollout streams.  What EncodeRolloutPipeline uses when no options are given -- and what the tests of the kernel forms ask
    for, so that they follow the pipeline's choice (tests/test_rollout_opts_gpu.py, tests/test_layer_tok_gpu.py)."""
    # several chains share the rollout CUs: seam launches (consumers spinning on a CU each) cost more than they
    # save.  When the attention workgroups of the two units in flight cover the rollout partition at one per video,
    # the CUs are the bound: wide FFN workgroups (one load of the weight chunk per 128 rows) and one attention
    # workgroup per video running all 8 heads (the layer input ingested and normalised once, finished rows out
    # instead of four partials).  Smaller units (C5 at 8 or 16 videos per batch) would leave most of those CUs idle:
    # the latency forms, four times the workgroups (224 vs 120 k frames/s at B = 8, 251 vs 219 k at B = 16).  Same bits.
    wide = 2 * int(group) * int(batch) >= int(roll_cus)
    opts = {'cus': int(roll_cus),   # (seam launches -- off below anyway -- only when their grid fits the rollout CUs)
            'seam': False,
            'ffn_rows': 128 if wide else 64,
            'attn_heads': 8 if wide else 2,
            'attn_rows': 0, 'ffn_tile': 0}
    # units of >= 2048 token rows (C2: 128 videos x 42 rows; C4: 64 x 36; C5: 256 x 48) run both blocks of a layer in their
    # ROW-TILE forms: LN1 + q|k|v on 64-row tiles of the whole unit + one attention-core workgroup per video (attn_rows.hip),
    # and the FFN as one workgroup per 64-row tile over all hidden chunks (ffn_tile.hip) -- rows packed across videos, each
    # row ingested and normalised once per block, weights streamed as fragments: about half the CU time of the all-heads /
    # chunk-partial workgroups (C5 305 -> 383 k frames/s, C2 405 -> 415-419 k, C4 172 -> 180 k).  Same bits.
    hist = getattr(rollouter, 'cond_len', None) or getattr(rollouter, 'history_len', burn_in or 1)
    tiles = wide and int(group) * int(batch) * int(rollouter.num_slots) * int(hist) >= 2048
    opts['attn_rows'] = 128 if tiles else 0
    # (2: the FFN tile launch also runs LN1 + q|k|v of the next layer on its rows -- one launch less per layer: C2 441 -> 447 k,
    #  C5 432 -> 446 k, C4 175 -> 181 k)
    opts['ffn_tile'] = 2 if tiles else 0
    # the layers before the last as ONE token-stationary launch (layer_tok.hip) -- see tok_unit_batches; the last (row-pruned) layer keeps the forms above
    opts['layer_tok'] = bool(tok)
    return opts


def unit_sizes_for(n, group, rows_per_batch, ramp=False, spread=True):
    """Batches per rollout unit of a run over n batches (a pure function of the counts: tests/test_pipeline_plan.py).  Units of `group` batches;
    with `ramp` the LAST batches go into ever smaller units (group 4: .., 4, 2, 1, 1); otherwise, where a unit may grow beyond `group` within one
    round of row tiles (8192 token rows), the last TWO units split what the full units in front of them leave -- no short unit of its own behind
    the last encode (it would roll out in the latency forms with nothing beside it: C2 at 21 / 22 / 23 batches 452 / 464 / 476 k frames/s against
    504 k at 20; now 20: 4, 4, 6, 6; 21: 4, 4, 4, 4, 5; 23: 4, 4, 4, 5, 6 -- 507-515 k; profiles/r04_probes.txt section 15).  Two units at most: a
    size other than `group` has two unit objects, a third unit of it in a row waits for the first (4, 4, 5, 5, 5: 413 k)."""
    G, n = int(group), int(n)
    tail = []
    if ramp and G >= 2 and n >= 2 * G:
        g = G // 2
        while g >= 1:
            tail.append(g)
            g //= 2
        tail.append(1)                      # G = 4: [2, 1, 1];  G = 2: [1, 1]
        while sum(tail) > G:                # (G = 3: [1, 1, 1] -> [1, 1]; the tail replaces at most one full unit)
            tail.pop(0)
    rest = n - sum(tail)
    sizes = [G] * (rest // G) + ([rest % G] if rest % G else []) + tail
    gmax = 8192 // max(int(rows_per_batch), 1)
    if not tail and spread and gmax > G and rest >= 3 * G:
        x = 1
        while rest - x * G > 2 * gmax:
            x += 1
        last = rest - x * G
        if last >= 2 * G:
            sizes = [G] * x + [last // 2, last - last // 2]
    return sizes, len(tail)


class EncodeRolloutPipeline:
    """savi: StoSAVi / STEVE container (eval, testing=True); rollouter: SlotRollouter / SingleStepSlotRollouter container.

    batch: videos per batch (fixed: the rollout graphs are captured for it); burn_in: encoded frames per video
    (= rollouter.history_len, or 1 for the single-step rollouter); pred_len: rollout steps.
    partition: 'pair' (default; see the module docstring), 'three', 'two' (one encode stream on `encode_cu_word`, the
    rollout on the complement) or 'none' (plain streams, shared CUs).  encode_cu_word: see encode_mask_words; for 'pair'
    a 'rows<R>' string moves the split (encode on 32 R CUs; default: sized from the two sides' CU time, _encode_rows).
    group: batches per rollout unit / graph (None = 4 for 'pair' with a rollouter on the fused-layer path, 1 otherwise).
    steal_steps: time steps of convolutions per batch computed on the rollout streams (may be fractional: 1.25 = one step,
    two for every fourth batch); None = the partition's tuned default.
    encode_graph: replay the encode of a batch from a hipGraph over fixed input buffers as well (frames / noise / stolen
    features staged into them; bit-identical; default on, False: eager launches).
    rollout_opts: per-call kernel options of the captured rollouts (engine.rollout_opts); None = the partition's default
    ('pair': no seam launches; 128-row FFN workgroups and all-heads attention workgroups -- the throughput settings -- when
    the two units in flight cover the rollout CUs with one attention workgroup per video, else 64-row / head-pair workgroups;
    the same bits as the library defaults either way; other partitions: the library defaults).
    hybrid: behind the whole-chip fill, every hybrid-th batch is encoded on an unmasked stream beside the CU-masked lane (None = 5
    for a balanced 'pair' on row tiles, 0 = never otherwise; bit-identical).
    tok: None (default) = the FULL rollout units of a 'pair' pipeline run the layers before the last as token-stationary launches (csrc/layer_tok.hip)
    where the rollouter allows it and a unit reaches 96 videos (C2: units of 6 batches = 64 workgroups each, two side by side on the rollout CUs);
    False = never (every unit in the row-tile / latency forms: bit-identical to the serial module calls); True = required.  With it the results agree
    with the serial calls to ~5e-6 over 50 steps instead of bit for bit (one accumulator per output block instead of per-chunk partial sums), and are
    bit-identical to run(serial=True) of the same object.
    roll_streams: rollout streams of the 'pair' partition = units in flight on the rollout CUs (default 2).
    split: True = the encode in two halves where the model allows it (image features on the encode lane, the slot branch of a whole unit as one
    video-stationary launch behind the features of its last batch: csrc/slot_chain.hip; agrees with the default to split-bf16 rounding, ~5e-6);
    None / False (default) = the whole encode on the lane -- measured faster (profiles/r06_probes.txt: the pipeline is bound by its rollout units).
    chain_on: with split, where a unit's slot branch runs: 'enc' (behind the features of its last batch, on that batch's stream) or 'roll' (at the head
    of the unit's rollout graph).
    decoder: module holding the SAVi decoder weights (StoSAVi / SlotFormer); enables run(..., decoded={...}): the predicted frames of
    every batch decoded to reconstruction + segmentation behind its rollout, on an unmasked stream of its own (the decode is 12x the
    FLOPs of encode + rollout at C2: it bounds such a run, the other two stages hide beside it).  seg_dtype: uint8 (default) or int64.
    """

    def __init__(self, savi, rollouter, batch, burn_in, pred_len, encode_cu_word=0xff, steal_steps=None, use_graph=True,
                 partition='pair', group=None, rollout_opts=None, encode_graph=None, hybrid=None, decoder=None, seg_dtype=torch.uint8,
                 encode_fork=None, tok=None, split=None, chain_on='enc', roll_streams=2):
        self.savi, self.roll = savi, rollouter
        # optional third stage (row N2; video_prediction/test_vp.py:55-63,145-146 -> slotformer.py:244-259 -> savi.py:504-525 ->
        # vp_utils.py:20-41): the predicted frames of every batch are decoded behind its rollout -- spatial-broadcast decoder, softmax
        # over slots, postproc_mask -- into the reconstruction and the segmentation test_vp.py scores.  decoder: the module that holds
        # the (frozen) decoder weights: the StoSAVi itself or the SlotFormer that copied them (slotformer.py:203-210)
        self.decoder, self.seg_dtype = decoder, seg_dtype
        self._s_dec = None
        self.B, self.T, self.H = int(batch), int(burn_in), int(pred_len)
        p = next(rollouter.parameters())
        if not p.is_cuda:
            raise RuntimeError('slotformer_amd: the pipeline needs the models on a HIP device; there is no CPU fallback')
        if self.T != engine.burn_in_of(rollouter):
            raise RuntimeError(f'slotformer_amd: burn_in = {self.T}, but this rollouter consumes {engine.burn_in_of(rollouter)} frames '
                               '(history_len; 1 for SingleStepSlotRollouter)')
        self.dev = p.device
        self.N, self.D = rollouter.num_slots, rollouter.in_proj.in_features
        if partition not in ('pair', 'three', 'two', 'none'):
            raise ValueError("slotformer_amd: partition must be 'pair', 'three', 'two' or 'none'")
        if not encode_cu_word:
            partition = 'none'
        if partition == 'three' and self.B < 4:
            partition = 'two'
        self._lib = _lib.lib()
        # rollouters on the fused-layer path (per-video / per-row kernels) give a video the same bits in any batch: their
        # batches can share a rollout unit.  The generic GEMM path picks tiles by problem size: one batch per unit there.
        self.fused = bool(self._lib.sf_rollout_is_fused(C.byref(engine.rollouter_plan(rollouter).struct)))
        if group and int(group) > 1 and not self.fused:
            raise RuntimeError('slotformer_amd: group > 1 needs a rollouter on the fused-layer path (d_model 256, 8 heads, ffn 1024, '
                               'window <= 64 tokens): the generic path\'s results depend on the batch size')
        # token-stationary layer launches (csrc/layer_tok.hip) for the FULL units of a 'pair' pipeline: 64 workgroups per unit, two units side by side on
        # the 128 rollout CUs; drain units and units of fewer batches keep the row-tile / latency forms (a unit alone is faster in them: 14.5 against
        # 20.8 ms for 192 videos) -- whose results differ from the token-stationary ones in the last bits (1e-6 per layer, 5e-6 over 50 steps)
        g_tok = tok_unit_batches(rollouter, self.B, self.T) if (partition == 'pair' and self.fused and tok is not False) else None
        if g_tok is None and partition == 'pair' and self.fused and tok is not False and (tok or group):
            # models of more than four layers take the form in long runs only, and the run length is the caller's knowledge: units sized by
            # unit_batches_for(.., n_batches) for such a run (bench.py, harness.extract_and_rollout), or tok=True, say so
            g_long = tok_unit_batches(rollouter, self.B, self.T, n_batches=1 << 30)
            if g_long is not None and (tok or int(group) == g_long):
                g_tok = g_long
        if tok and g_tok is None:
            raise RuntimeError('slotformer_amd: tok=True needs the pair partition, a rollouter whose layers take the token-stationary form '
                               '(d_model 256, 8 heads, ffn 1024, windows of <= 64 tokens) and units of >= 96 videos')
        self.G = int(group) if group else (int(os.environ.get('SF_PIPE_GROUP', str(g_tok or 4))) if (partition == 'pair' and self.fused) else 1)
        self.tok = g_tok is not None and self.G * self.B >= 96
        if self.G < 1:
            raise ValueError('slotformer_amd: group >= 1')
        nroll = max(1, int(roll_streams)) if partition == 'pair' else 1
        self._nroll = nroll
        # unit buffers / graphs / workspaces: per rollout stream one rolling out + one being filled or spare
        self.NU = 2 * nroll
        self.lead = 2 * nroll                    # stolen features of unit u are computed behind the rollout of unit u - lead
        if steal_steps is None:
            steal_steps = float({'pair': 0, 'two': 1, 'three': 0}.get(partition, 1))
        self.steal = max(0.0, min(float(steal_steps), float(self.T)))   # may be fractional: see _steal_of
        self._steal_arg = self.steal
        self._masked_taken = {}
        self._lib = _lib.lib()
        # 'pair': the encode partition's CU mask (the rollout streams get the complement)
        self._enc_words_pair = ENC_WORDS_P
        if partition == 'pair':
            if isinstance(encode_cu_word, str) and encode_cu_word.startswith('rows'):
                self._enc_words_pair = encode_mask_words(encode_cu_word)
            elif os.environ.get('SF_PIPE_CU_SPLIT'):
                self._enc_words_pair = encode_mask_words(os.environ['SF_PIPE_CU_SPLIT'])
            elif self._encode_rows() != 4:
                self._enc_words_pair = encode_mask_words(f'rows{self._encode_rows()}')
        if rollout_opts is None and partition == 'pair':
            roll_cus = 256 - sum(bin(w).count('1') for w in self._enc_words_pair)
            rollout_opts = pair_unit_options(self.roll, self.B, self.G, roll_cus, self.tok, self.T)
            self._row_tiles = bool(rollout_opts['attn_rows'])
        if partition in ('three', 'two') and (rollout_opts is None or (isinstance(rollout_opts, dict) and 'cus' not in rollout_opts)):
            # the library's seam launches need their whole grid resident on the CUs the rollout stream may use: tell it how many those are
            cus = 168 if partition == 'three' else 256 - sum(bin(w).count('1') for w in encode_mask_words(encode_cu_word))
            rollout_opts = dict(rollout_opts or {}, cus=cus)
        if isinstance(rollout_opts, dict) and 'layer_tok' not in rollout_opts:
            rollout_opts = dict(rollout_opts, layer_tok=False)   # (caller-given options: the token-stationary form only when asked for)
        self.rollout_opts = engine.rollout_opts(rollout_opts)
        self.tok = self.tok and self.rollout_opts is not None and self.rollout_opts.layer_tok > 0
        # the same options without the token-stationary launches: a full-size unit that rolls out alone (the drain of a run)
        self.row_opts = self.rollout_opts
        if self.tok:
            o = self.rollout_opts
            self.row_opts = _lib.sf_rollout_opts(o.precision, o.seam_fused, o.ffn_rows, o.attn_heads_per_wg, o.attn_qkv_rows, o.ffn_tile, o.cus_available, -1)
        # units of fewer batches (the ramp at both ends of a run) are on the critical path of fill and drain: the latency forms
        # of the kernels (head-pair attention workgroups, narrower FFN workgroups: more, shorter workgroups per launch) -- the
        # same bits
        self.tail_opts = self.rollout_opts
        if self.rollout_opts is not None and (self.rollout_opts.ffn_rows > 64 or self.rollout_opts.attn_heads_per_wg == 8 or
                                              self.rollout_opts.attn_qkv_rows or self.rollout_opts.ffn_tile):
            self.tail_opts = _lib.sf_rollout_opts(self.rollout_opts.precision, self.rollout_opts.seam_fused, min(self.rollout_opts.ffn_rows or 64, 64), 2, 0, 0,
                                                  self.rollout_opts.cus_available, -1)
        self.use_graph = bool(use_graph)
        # the encode under a hipGraph too: the gaps between its ~60 short launches shrink (374-377 vs 373 k frames/s at 20
        # batches, 399.5 vs 398.2 at 40) at the price of a 38 MB device copy of the frames into the fixed input buffer per batch
        self.encode_graph = bool(int(os.environ.get('SF_PIPE_ENCODE_GRAPH', '1'))) if encode_graph is None else bool(encode_graph)
        self._enc_graphs = {}
        # (encode_fork=True: every encode graph with two branches; measured slower inside the pipeline -- the kwarg stays for measurements)
        self.encode_fork = bool(encode_fork)
        # the last unit of a run rolls out alone on an unmasked stream: in the kernels' latency forms (head-pair attention, 64-row FFN
        # workgroups) while a unit is small -- C4, 64 videos: 172 vs 165 k frames/s -- but a large unit fills the chip with its row tiles and
        # four times the workgroups only queue: C5, 256 videos: 392 -> 435 k; C2, 128 videos: 436 / 440 k
        self.drain_latency_form = self.G * self.B < 128
        self._key = ('pipe', id(self))
        # The encode in two halves (round 6, csrc/slot_chain.hip): the image features of a batch on the encode lane (five dense launches), the slot
        # branch of a WHOLE rollout unit as one video-stationary launch in front of the unit's rollout, on the rollout stream -- one workgroup per video
        # for ~0.6 ms, which on the lane left 96 of its 128 CUs idle while nothing else could start.  Where the model's slot branch has that form
        # (engine.savi_chain_ok) and the partition is 'pair'.  OPT-IN (split=True; bench.py --split): the encode lane gets 22 % faster (2.83 -> 2.2 ms per
        # C2 batch) but the rollout units, already slowed 25-40 % by whatever else runs on the chip, become the bound: 577 k against 626 k frames/s at 100
        # batches (profiles/r06_probes.txt).
        want_split = bool(split)
        self.split = bool(want_split and partition == 'pair' and self.fused and engine.savi_chain_ok(savi, self.B, self.T))
        self._with_noise = engine.kernel_noise(self.savi, torch.empty(0), 1, self.T, self.dev) is not None   # (no draw: the gate only)
        # where the slot branch of a unit runs: 'enc' = behind the features of the unit's last batch, on that batch's stream (the encode side);
        # 'roll' = at the head of the unit's rollout graph (the rollout stream)
        self.chain_on = 'roll' if chain_on == 'roll' else 'enc'
        if self.split:
            self.steal = 0.0   # (no stolen convolutions: the feature half IS the lane's work)
        self._plan = None
        self._sig = None
        self.units = []
        self.spread_remainder = True
        hist_ = getattr(rollouter, 'cond_len', None) or getattr(rollouter, 'history_len', None) or self.T
        self._rows_per_batch = self.B * int(rollouter.num_slots) * int(hist_)
        self._tails = {}
        self._capture_all()
        self.cu_split = False
        self.partition = 'none'
        self.s_roll = None
        self.roll_streams = []           # rollout streams: unit u rolls out on roll_streams[u % len]
        self.lanes = []                  # encode lanes: (stream, first video, end video)
        if partition != 'none':
            try:
                if partition == 'pair':
                    enc_words = self._enc_words_pair
                    roll_words = [~w & 0xffffffff for w in enc_words]
                    self.roll_streams = [self._masked_stream(roll_words) for _ in range(self._nroll)]
                    self.s_roll = self.roll_streams[0]
                    self.lanes = [(self._masked_stream(enc_words), 0, self.B)]
                    self.encode_cus = sum(bin(w).count('1') for w in enc_words)
                    self.rollout_cus = 256 - self.encode_cus
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
                self._close_streams()
                self.s_roll, self.lanes, self.roll_streams = None, [], []
        if self.s_roll is None:
            self.s_roll = self._pool_stream('roll', priority=-1)
            self.lanes = [(self._pool_stream('lane'), 0, self.B)]
            self.encode_cus = self.rollout_cus = 256
        if not self.roll_streams:
            self.roll_streams = [self.s_roll]
        # unmasked streams for the drain units
        self.stream_placement = None
        self.s_free = self._pick_free_streams(2) if len(self.roll_streams) > 1 else []

        self.s_enc = self.lanes[0][0]
        self.fill_whole_chip = True      # the first encode(s) of a run on the calling stream (all CUs)
        # (group 1: a second whole-chip encode was measured worse, 311 vs 323 k frames/s at 20 steps; with group 2 the first
        #  rollout cannot start before both batches of its unit are encoded)
        # smaller units at the end of a run (_unit_plan): measured WORSE with group 4 (unmasked drain units take CUs from the
        # encode and the full units: 322 vs 376 k frames/s at 20 batches) -- off
        self.ramp = False
        self._hybrid_arg = hybrid
        self.fill_par = 2   # whole-chip encodes side by side during the fill (3: 495 vs 552 k frames/s)
        # batches encoded on the WHOLE chip (unmasked streams, fill_par at a time) at the start of a run.  The first unit's had to be
        # (nothing else runs yet); since the row-tile kernels the rollout streams have slack, and the unmasked encodes of the NEXT
        # units -- 2.3 ms per batch beside the first rollouts instead of 3.9 on the encode partition -- build a backlog the masked lane
        # then works from: two units by default (C4 172 -> 175 k, C5 381 -> 390 k frames/s at 20 batches), three where the encode lane is
        # the bound of a balanced pair running row tiles (C2: 427 -> 445-447 k at 20 batches, 441 -> 460 k at 40; four or more units
        # starve the rollouts: 432 / 410 k).  SF_PIPE_FILL overrides (batches)
        balanced = getattr(self, '_row_tiles', False) and partition == 'pair' and self._encode_rows() == 4
        fill_units = 3 if balanced else 2
        # hybrid lane: behind the fill every `hybrid`-th batch is encoded on an UNMASKED stream (the second fill graph) beside the CU-masked
        # lane.  With the encode lane the bound (3.9 ms per batch against ~3.0-3.3 of rollout capacity) the rollout partition has slack to
        # lend: C2 at 40 / 100 / 200 batches 484 / 489 / 487 -> 501 / 511 / 526 k frames/s with every 5th (4: 496 / 510, 6: 497 / 507), nothing
        # at 20 (one such batch); C4 209 -> 212 k at 60; the rollout-bound C5 (one CU row for the encode) keeps the lane alone.  None
        # among the last three batches of a run: the lane is about to fall idle there (C2 487.5 vs 480.5 at 20, 517.8 vs 513.2 at 100 against none among the last five).  Same bits (tests/test_pipeline_gpu.py).
        if self._hybrid_arg is not None:
            self.hybrid = int(self._hybrid_arg)
        else:
            # (units of more than 4 batches -- unit_batches_for: long runs of small batches -- make the rollouts cheaper per batch and the encode the
            #  bound by more: C4 with units of 7 at 84 batches, every 5th / 4th / 3rd / 2nd batch: 236.0 / 242.9 / 250.6 / 250.9 k)
            # (token-stationary units: every 8th batch -- round 6, encode lane 2.55 ms per batch: C2 at 40 / 60 / 100 batches 593.7 / 612.0 / 632.8 k frames/s with 6,
            #  601.7 / 621.6 / 643.2 k with 8, 639.6 k with 10 at 100; 20 batches 592.2 / 590.1 / 594.2 k with 6 / 8 / 4: neutral; profiles/r06_probes.txt section 21)
            self.hybrid = int(os.environ.get('SF_PIPE_HYBRID', ('8' if self.tok else '3' if self.G > 4 else '5') if balanced else '0'))
        self.hybrid_tail = min(self.hybrid, 3)
        self.fill_batches = int(os.environ.get('SF_PIPE_FILL', '0')) or (fill_units * self.G if partition == 'pair' else 0)
        self.fill_steal = 0
        # pre_steal[h]: time steps of convolutions of batch h of the NEXT unit computed on a rollout stream right before a unit
        # rolls out (group >= 2 only; batch 0 of the next unit starts encoding at once and cannot wait)
        ps = ''
        self.pre_steal = [int(x) for x in ps.split(',')] if ps else []   # (measured neutral at 20 batches: 381-383 k frames/s with [0,0,1,1] .. [0,1,1,1] and off)
        self.pre_steal = [min(k, self.T) for k in self.pre_steal]
        if not any(self.pre_steal):
            self.pre_steal = []
        self.feat_bufs = None
        self._stage, self._s_copy, self._s_out = None, None, None   # staging ring + copy streams for host-resident inputs / outputs
        self.completion_events = []      # one event per unit of the last run() ...
        self.completion_batches = []     # ... and the number of batches it completed
        # the encode graphs hold raw pointers into the SAVi encoder's plan (packed conv weights, folded Slot-Attention matrices,
        # predictor fragments): the pipeline keeps that plan alive and re-captures when the encoder's parameters change (_check_plan)
        self._enc_plan = engine.encoder_plan(self.savi)
        self._enc_sig = self._enc_plan.sig
        self._capture_encode_graphs()

    def _capture_encode_graphs(self):
        """Capture the encode graphs of the lanes a run uses NOW, not inside the first run that reaches them (a short warm-up only
        touches the fill lanes): full batches, no stolen features."""
        if self.split:
            # the feature halves run eagerly (five launches per batch): only the persistent convolution needs to know how many CUs an UNMASKED
            # stream of the fill / hybrid lanes may take (csrc/conv_ws.hip sizes its grid for the stream's CUs)
            if self.cu_split and self.tok:
                for st in list(self.s_free):
                    self._lib.sf_stream_set_cus(C.c_void_p(st.cuda_stream), int(self._lib.sf_stream_cus(None)))
            return
        if self.encode_graph and self.steal == 0 and not self.fill_steal and not self.pre_steal:
            res = getattr(self.savi, 'resolution', (128, 128))[0]
            with_noise = engine.kernel_noise(self.savi, torch.empty(0), 1, self.T, self.dev) is not None   # (no draw: the gate only)
            with torch.no_grad():
                for li, (_, lo, hi) in enumerate(self.lanes if self.lanes else [(None, 0, self.B)]):
                    self._encode_graph_for(li, hi - lo, 0, res, with_noise)
                if self.cu_split and self.fill_whole_chip and self.s_free and self.fill_par > 1:
                    for fi in range(self.fill_par):
                        self._encode_graph_for(('fill', fi), self.B, 0, res, with_noise)
            torch.cuda.synchronize(self.dev)

    # ------------------------------------------------------------------------------------------------------------
    @property
    def bufs(self):
        """slot buffers of the rollout units ([group * B, T + H, N, D] each)"""
        return [u.buf for u in self.units]

    @property
    def graphs(self):
        return [u.graph for u in self.units if u.graph is not None]

    def _encode_rows(self):
        """CU rows (of 8; 32 CUs each) of the encode partition when the caller names no split.  Both partitions are bound by CU
        time (workgroups x time in the kernel), so the split follows the two sides' CU time per batch, estimated from the
        per-launch figures of profiles/r03_kernel_stats.csv: encode 2.45 CU-ms per 128 x 128 frame at slot size 128 (more
        with wider slots); rollout per video, step and layer 51 us (all-heads attention workgroup) + 28 us per 128 FFN rows.
        Balanced pairs (C2: share 0.55; C4: 0.59) keep the even split, the one every other split lost to; a rollout-heavy pair
        (C5, 1 + 80 frames: share 0.06) gives the encode one row: 305 k vs 221 k frames/s (profiles/r03_probes.txt)."""
        if not self.fused:
            return 4
        res = getattr(self.savi, 'resolution', (128, 128))
        slot = self.D
        enc = self.B * self.T * (res[0] * res[1] / 16384.0) * 2.45 * (slot / 128.0) ** 1.5
        hist = getattr(self.roll, 'cond_len', None) or getattr(self.roll, 'history_len', self.T)
        L = self.N * hist
        roll = self.B * self.H * len(self.roll.transformer_encoder.layers) * (0.051 + 0.028 * L / 128.0)
        share = enc / (enc + roll)
        return 4 if share >= 0.4 else max(1, round(8 * share))

    def _masked_stream(self, words):
        """A CU-masked stream from the process-wide pool (_STREAMS): the k-th stream with these mask words a pipeline asks for is the same
        object for every pipeline of the process.  The streams a process creates FIRST get hardware queues of their own; a second
        pipeline object with streams of its own ran at 405 k frames/s beside 455 k for the first one -- and at 373 k once the first was
        closed (tools/two_pipes_probe.py): its queues share hardware queues.  Pipelines are used one call at a time; two of them alive
        (the harness keeps one per shape) now run on the same queues, in call order."""
        key = (self.dev.index, tuple(int(w) for w in words))
        k = self._masked_taken.get(key, 0)
        self._masked_taken[key] = k + 1
        with _STREAMS_LOCK:
            st = _STREAMS.get(('masked', ) + key + (k, )) if _POOL else None
            if st is None:
                arr = (C.c_uint * 8)(*words)
                h = C.c_void_p()
                _lib.check(self._lib.sf_stream_create_cu_mask(C.byref(h), arr, 8))
                st = torch.cuda.ExternalStream(h.value, device=self.dev)
                _STREAMS[('masked', ) + key + (k, )] = st
        return st

    def _pool_stream(self, kind, k=0, priority=0):
        """an unmasked torch stream from the same pool (kind: 'out', 'copy', 'roll', 'lane')"""
        key = ('torch', self.dev.index, kind, k, priority)
        with _STREAMS_LOCK:
            st = _STREAMS.get(key) if _POOL else None
            if st is None:
                st = _STREAMS[key] = torch.cuda.Stream(device=self.dev, priority=priority)
        return st

    def _pick_free_streams(self, n):
        """n unmasked streams from the process-wide pool for the whole-chip fill encodes and the drain units.
        WHICH hardware queue a stream lands on decides 10 % of the pipeline's throughput: the runtime gives a stream its queue on first
        use, round-robin over four, and the same run takes 84 ms with the free streams on the first two queues behind the null stream's
        (the streams PyTorch hands out first), 79.5 ms on the second and third, 94-97 ms when one of them shares the null stream's queue
        (`profiles/r03_probes.txt` section 16: period four in the number of streams used before; C2 447 / 474 / 400 / 390 k frames/s).
        So ONE pool stream is used (and parked) first and the next n are taken; dedicated queues (full CU masks) were measured no better
        than the worst-but-one arrangement, and choosing by a collision test (two 300 us one-wave kernels per candidate pair) changed the
        order of first use and landed on a slow arrangement -- the plain rule is what was validated in the bench, the harness and a
        process with several pipeline objects (tools/two_pipes_probe.py, tools/pcie_probe.py)."""
        key = ('free-set', self.dev.index, n)
        with _STREAMS_LOCK:
            got = _STREAMS.get(key)
            placed = _STREAMS.get(('placement', self.dev.index))
        if got is not None:
            self.stream_placement = placed
            return list(got)
        # (a process that has initialised RCCL already has that one stream in use -- RCCL's: with torch.distributed on the nccl backend
        #  initialised before the pipeline, the bench's multi-GPU path, no parked stream 474 k, one 418 k, two / three 385 / 399 k)
        rccl = pg_nccl = False
        try:
            import torch.distributed as dist
            rccl = pg_nccl = dist.is_available() and dist.is_initialized() and 'nccl' in str(dist.get_backend())
            if rccl:
                # what counts is whether RCCL's communicator -- and with it its stream -- EXISTS yet: a process group initialised without device_id creates it
                # at its first collective; the pipeline's streams come first then, as in a process without RCCL (bench.py --force-dist, same box:
                # communicator first 522-537 k frames/s, pipeline first 546-563 k, no process group 558-562 k; profiles/r05_probes.txt section 9)
                try:
                    rccl = bool(dist.distributed_c10d._get_default_group()._get_backend(torch.device('cuda', self.dev.index))._is_initialized())
                except Exception:  # noqa: BLE001
                    rccl = True
        except Exception:  # noqa: BLE001
            rccl = False
        n_skip = 0 if rccl else 1
        # what was chosen, for the bench line / the logs of every rank (the rule is tuned to this runtime's round-robin over four
        # hardware queues)
        self.stream_placement = {'parked_streams_before_the_free_ones': n_skip, 'free_streams': n, 'rccl_initialised': bool(pg_nccl), 'rccl_communicator_exists': bool(rccl),
                                 'rule': 'rccl communicator exists: none parked' if rccl else 'one parked',
                                 'rank': int(os.environ.get('RANK', '0')), 'device': self.dev.index}
        if int(os.environ.get('WORLD_SIZE', '1')) > 1:
            import sys
            print(f'[slotformer_amd.pipeline] stream placement: {self.stream_placement}', file=sys.stderr, flush=True)
        with _STREAMS_LOCK:
            _STREAMS[('placement', self.dev.index)] = self.stream_placement
        for _ in range(n_skip):
            sk = torch.cuda.Stream(device=self.dev)
            with torch.cuda.device(self.dev):
                _lib.check(self._lib.sf_debug_spin(1, sk.cuda_stream))   # (used: a stream gets its hardware queue on first use)
            sk.synchronize()
            with _STREAMS_LOCK:
                _STREAMS[('parked', self.dev.index, len(_STREAMS))] = sk
        picked = [torch.cuda.Stream(device=self.dev) for _ in range(n)]
        with _STREAMS_LOCK:
            _STREAMS[key] = tuple(picked)
        return picked

    def _close_streams(self):
        pass   # (the streams belong to the process-wide pool and live as long as the process)

    def close(self):
        self._close_streams()
        self.units, self._tails, self._enc_graphs = [], {}, {}
        engine.release_workspaces(self._key)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    # ------------------------------------------------------------------------------------------------------------
    def _new_unit(self, nb, tag, latency_form=False, row_form=False):
        with torch.no_grad():
            u = _Unit(torch.zeros(nb * self.B, self.T + self.H, self.N, self.D, device=self.dev), self._key + (tag, ))
            u.latency_form, u.row_form = latency_form, row_form
            if self.split:
                u.planes = torch.zeros(nb, engine.savi_planes_bytes(self.savi, self.B, self.T), dtype=torch.uint8, device=self.dev)
                u.noise = torch.zeros(nb * self.B, self.T, self.N, self.D, device=self.dev) if self._with_noise else None
            self._rollout_eager(u)   # allocates its workspace, builds the plan
            torch.cuda.synchronize(self.dev)
            if self.use_graph:
                g = torch.cuda.CUDAGraph()
                # (thread_local: other host threads may keep allocating / synchronising while this one captures; their calls into the library
                #  wait at _lib.CAPTURE_GATE: launches beside a capture are not safe on this runtime)
                with _lib.CAPTURE_GATE, torch.cuda.graph(g, capture_error_mode='thread_local'):
                    self._rollout_eager(u)
                u.graph = g
        return u

    def _capture_all(self):
        """(Re-)capture every unit graph for the rollouter's CURRENT parameters."""
        self.units, self._tails = [], {}
        self.units = [self._new_unit(self.G, k) for k in range(self.NU)]
        if self.G > 1 and self.NU > 2 and self.tail_opts is not self.rollout_opts and self.drain_latency_form:
            self._tails['drain'] = self._new_unit(self.G, ('drain', ), latency_form=True)   # (not inside a timed run)
        elif self.G > 1 and self.NU > 2 and self.tok:
            self._tails['drain'] = self._new_unit(self.G, ('drain', ), row_form=True)
        self._plan = engine.rollouter_plan(self.roll)   # keeps the packed weight copies the graphs point to alive
        self._sig = self._plan.sig

    def _tail_unit(self, nb, k=0, row_form=False):
        """a unit of fewer than `group` batches (the ramp at both ends of a run, a ragged remainder); k: which of the two; row_form: the last unit of a
        run of token-stationary units"""
        key = (nb, k, bool(row_form))
        if key not in self._tails:
            self._tails[key] = self._new_unit(nb, ('tail', ) + key, row_form=bool(row_form))
        return self._tails[key]

    def _check_plan(self):
        if engine._signature(self.roll) != self._sig or getattr(self.roll, '_sf_plan', None) is not self._plan:
            # parameters changed since the capture (optimizer step, load_state_dict, invalidate): the graphs point at the
            # old packed weight copies
            torch.cuda.synchronize(self.dev)
            self._capture_all()
        if engine._signature(self.savi) != self._enc_sig or getattr(self.savi, '_sf_plan', None) is not self._enc_plan:
            # the same for the encoder: its graphs replay kernels with pointers into the OLD plan's packed / folded copies (and an
            # eager savi_encode / savi_cnn call elsewhere may have rebuilt -- and freed -- that plan)
            torch.cuda.synchronize(self.dev)
            self._enc_graphs = {}
            self._enc_plan = engine.encoder_plan(self.savi)
            self._enc_sig = self._enc_plan.sig
            if self.split:
                self._capture_all()   # (the unit graphs hold the slot branch: pointers into the encoder's packed copies)
            self._capture_encode_graphs()

    def _steal_of(self, j):
        """time steps of convolutions stolen for batch j: integers that average to self.steal (1.25 -> 1, 1, 1, 2, ...)"""
        return int(math.floor(self.steal * (j + 1) + 1e-9) - math.floor(self.steal * j + 1e-9))

    def _chain(self, u):
        """split encode: the slot branch of the unit's batches -- one launch, one workgroup per video; the slots of the burn-in frames land in the unit buffer"""
        engine.savi_slots_chain(self.savi, u.planes, u.buf.shape[0] // self.B, self.B, self.T, u.buf, noise=u.noise, ws_slot=self._key + ('chain', ) + u.key[2:])

    def _rollout_eager(self, u):
        if self.split and self.chain_on == 'roll':
            # the slot branch of the unit's batches: one launch, one workgroup per video; the slots of the burn-in frames land in the unit buffer
            engine.savi_slots_chain(self.savi, u.planes, u.buf.shape[0] // self.B, self.B, self.T, u.buf, noise=u.noise, ws_slot=self._key + ('chain', ) + u.key[2:])
        full = u.buf.shape[0] >= self.G * self.B
        if self.tok:
            # token-stationary launches for every unit of >= 96 videos but the LAST of a run (row_form: alone on the chip at the end, latency counts:
            # row tiles, 14.5 against 20.8 ms for 192 videos); units below 96 videos in the latency forms
            if u.buf.shape[0] < 96 or u.latency_form:
                opts = self.tail_opts
            elif u.row_form:
                opts = self.row_opts
            else:
                opts = self.rollout_opts
        elif full and not u.latency_form:
            opts = self.rollout_opts
        else:
            opts = self.tail_opts
        engine.rollout(self.roll, u.buf, self.T, self.H, ws_slot=u.key, opts=opts)

    def _rollout(self, u):
        if u.graph is not None:
            u.graph.replay()
        else:
            self._rollout_eager(u)

    def _encode(self, img, noise, dst, feat_pre, lo=0, hi=None, lane=0, unit=None):
        """videos [lo, hi) of one batch -> dst[lo:hi, :burn_in] on the current stream (split encode: the batch's feature rows -> slot h of the unit's
        planes, its kernel noise -> the unit's noise rows; unit = (_Unit, h))"""
        hi = self.B if hi is None else hi
        if self.split:
            u, h = unit
            if u.noise is not None:
                u.noise[h * self.B:(h + 1) * self.B].copy_(engine.kernel_noise(self.savi, noise, self.B, self.T, self.dev))
            engine.savi_features(self.savi, img, u.planes[h], ws_slot=self._key + ('feat', lane))
            return
        if lo != 0 or hi != self.B:
            img = img[lo:hi]
            noise = None if noise is None else noise[lo:hi]
        # None for models that sample nothing (kld_method 'none': OBJ3D / PHYRE SAVi; STEVE), the caller's tensor, or fresh
        # eps ~ N(0,1) per frame as the reference draws it (savi.py:355-365)
        noise = engine.kernel_noise(self.savi, noise, hi - lo, self.T, self.dev)
        if self.encode_graph:
            # the ~60 launches of an encode replayed from a hipGraph over fixed buffers (one graph per (lane, stolen steps)):
            # frames, noise and stolen features are staged into them, the slots copied out
            eg = self._encode_graph_for(lane, hi - lo, 0 if feat_pre is None else feat_pre.shape[0], img.shape[-1], noise is not None)
            eg['img'].copy_(img)
            if noise is not None:
                eg['noise'].copy_(noise)
            if feat_pre is not None:
                eg['feat'].copy_(feat_pre)
            eg['graph'].replay()
            dst[lo:hi, :self.T].copy_(eg['post'])
            return
        side = None   # (one stream: see engine.savi_encode)
        if self.encode_fork:
            # (probe) eager two-branch encode: the slot branch on a second stream with the lane's own CU mask (an unmasked one for the fill lanes)
            cache = self.__dict__.setdefault('_fork_side', {})
            side = cache.get(lane)
            if side is None:
                if isinstance(lane, int) and self.cu_split:
                    side = self._masked_stream(self._lane_words(lane))
                else:
                    side = self._pool_stream('fork', k=len(cache))
                cache[lane] = side
        post, _, _ = engine.savi_encode(self.savi, img, noise=noise, feat_pre=feat_pre, ws_slot=self._key + ('enc', lane), side_stream=side)
        dst[lo:hi, :self.T].copy_(post)

    def _lane_words(self, lane):
        if self.partition == 'pair':
            return self._enc_words_pair
        if self.partition == 'three':
            return LANE0_WORDS_3 if lane == 0 else LANE1_WORDS_3
        return [0xffffffff] * 8

    def _encode_graph_for(self, lane, nv, k, res, with_noise):
        key = (lane, nv, k, res, with_noise)
        eg = self._enc_graphs.get(key)
        if eg is None:
            cur = torch.cuda.current_stream(self.dev)
            cl = list(self.savi.enc_channels)[-1]
            eg = {'img': torch.zeros(nv, self.T, 3, res, res, device=self.dev),
                  'noise': torch.zeros(nv, self.T, self.N, self.D, device=self.dev) if with_noise else None,
                  'feat': torch.zeros(k, nv, 64 * 64, cl, device=self.dev) if k else None}
            ws = self._key + ('encg', ) + key
            # capture on a side stream (the caller may be inside a masked stream) -- a FRESH one per graph: with every encode graph captured
            # on one pooled stream the same run took 93 instead of 78 ms (tools/two_pipes_probe.py)
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(cur)
            # encode_fork: the graph gets two parallel branches -- the image features of all time steps, and one step behind them the slot
            # branches (engine.savi_encode side_stream=; the seven-workgroup launches of the slot branch no longer hold up the convolutions)
            fork_here = self.encode_fork
            side2 = torch.cuda.Stream(device=self.dev) if fork_here else None
            # the graph of a CU-masked lane replays on that lane's CUs: the persistent kernels inside it (csrc/conv_ws.hip: one workgroup per CU)
            # size their grids for those, not for the capture stream's whole chip
            lane_cus = int(self.encode_cus // max(len(self.lanes), 1)) if (isinstance(lane, int) and self.cu_split) else 0
            if not isinstance(lane, int) and self.cu_split and self.tok:
                # the whole-chip fill / hybrid graphs of a pipeline whose rollout units hold their CUs for a whole launch anyway (token-stationary
                # units): one persistent convolution workgroup per CU of the device -- C2 at 20 / 60 batches 545 -> 567 k, 551 -> 612 k frames/s with
                # every 4th batch on the hybrid lane (192 / 384 / 512 workgroups: 574 / 559 / 565 k at 20; profiles/r05_probes.txt)
                lane_cus = int(self._lib.sf_stream_cus(None))
            if lane_cus:
                self._lib.sf_stream_set_cus(C.c_void_p(side.cuda_stream), lane_cus)
            try:
                with torch.cuda.stream(side):
                    engine.savi_encode(self.savi, eg['img'], noise=eg['noise'], feat_pre=eg['feat'], ws_slot=ws, side_stream=side2)   # workspace, plans
                    side.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with _lib.CAPTURE_GATE, torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
                        post, _, _ = engine.savi_encode(self.savi, eg['img'], noise=eg['noise'], feat_pre=eg['feat'], ws_slot=ws, side_stream=side2)
                cur.wait_stream(side)
            finally:
                # (a failed capture must not leave a CU count keyed by a stream handle torch hands out again)
                if lane_cus:
                    self._lib.sf_stream_set_cus(C.c_void_p(side.cuda_stream), 0)
            eg['graph'], eg['post'] = g, post
            self._enc_graphs[key] = eg
        return eg

    def _unit_plan(self, n):
        """[(first batch, number of batches, _Unit, drain?)] of a run over n batches.  With `ramp` the LAST batches of a run
        go into ever smaller units (group 4: .., 4, 2, 1, 1): the drain -- the rollouts still running once the last encode
        is done, which nothing overlaps -- then ends with short units in the kernels' latency forms, and those units run on
        unmasked streams (the encode partition is about to fall idle)."""
        G = self.G
        sizes, n_tail = unit_sizes_for(n, G, self._rows_per_batch, ramp=self.ramp, spread=self.spread_remainder)
        probe = getattr(self, 'unit_sizes_override', None)   # (tools: an explicit unit plan, e.g. "4,4,6,6"; at most two units of a size other than `group` in a row)
        if probe and sum(int(x) for x in probe.split(',')) == n:
            sizes = [int(x) for x in probe.split(',')]
        plan, u0, nfull, ntail = [], 0, 0, {}
        n_drain = n_tail
        if self.tok and len(sizes) >= 3 and sizes[-1] <= getattr(self, 'hybrid_tail', 2):
            # token-stationary units take 22.7 ms each whatever runs beside them: when the run ends in a SHORT unit, the last TWO units go to unmasked
            # streams -- the third unit of the driver's 20 batches (6, 6, 6, 2) would otherwise wait 15 ms for a rollout stream (488 -> 550 k frames/s).
            # Only when no hybrid-lane encode follows (the last unit's batches all take the masked lane): the first drain unit sits on that lane's
            # stream (24 batches as 6, 6, 6, 6 with two drain units: 506 k; 60 batches: 538 against 580 k)
            n_drain = max(n_drain, 2)
        for i, nb in enumerate(sizes):
            if nb == G:
                u = self.units[nfull % self.NU]
                nfull += 1
            else:
                u = self._tail_unit(nb, ntail.get(nb, 0) % 2, row_form=self.tok and i == len(sizes) - 1)
                ntail[nb] = ntail.get(nb, 0) + 1
            drain = i >= len(sizes) - max(n_drain, 1)
            if drain and nb == G and len(sizes) > 1 and self.tail_opts is not self.rollout_opts and self.drain_latency_form:
                # the last unit of a run rolls out alone on the whole chip (unmasked stream): a graph of the same buffer size
                # captured with the kernels' LATENCY forms (many short workgroups: 12.1 instead of 15.5 ms for 4 batches)
                if 'drain' not in self._tails:
                    self._tails['drain'] = self._new_unit(G, ('drain', ), latency_form=True)
                u = self._tails['drain']
            elif drain and nb == G and len(sizes) > 1 and self.tok and i == len(sizes) - 1:
                # ... with token-stationary full units: the LAST unit in the row-tile forms (alone on the chip 14.5 against 20.8 ms for 192 videos)
                if 'drain' not in self._tails:
                    self._tails['drain'] = self._new_unit(G, ('drain', ), row_form=True)
                u = self._tails['drain']
            plan.append((u0, nb, u, drain))
            u0 += nb
        return plan

    def prepare(self, n):
        """Create (and capture) every rollout unit a run over n batches uses -- the units of a remainder are built on first use otherwise, inside
        that run -- and return the unit sizes of the run."""
        self._check_plan()
        return [nb for _, nb, _, _ in self._unit_plan(int(n))]

    @torch.no_grad()
    def _decoded_buffers(self, decoded, n):
        """The decode stage's outputs: decoded['recon'] [n, B, pred_len, 3, R, R] float32 and decoded['seg'] [n, B, pred_len, R, R]
        (allocated when the caller's dict does not hold them)."""
        if self.decoder is None:
            raise RuntimeError('slotformer_amd: run(decoded=...) needs a pipeline built with decoder=<module holding the SAVi decoder>')
        R = engine.decoder_plan(self.decoder).struct.resolution
        if decoded.get('recon') is None:
            decoded['recon'] = torch.empty(n, self.B, self.H, 3, R, R, device=self.dev)
        if decoded.get('seg') is None:
            decoded['seg'] = torch.empty(n, self.B, self.H, R, R, device=self.dev, dtype=self.seg_dtype)
        rc, sg = decoded['recon'], decoded['seg']
        if tuple(rc.shape) != (n, self.B, self.H, 3, R, R) or tuple(sg.shape) != (n, self.B, self.H, R, R) or not rc.is_contiguous() or not sg.is_contiguous():
            raise RuntimeError(f'decoded buffers must be contiguous [n,{self.B},{self.H},3,{R},{R}] / [n,{self.B},{self.H},{R},{R}]')
        return rc, sg

    def _decode(self, slots_all, recon, seg):
        """predicted frames of one batch: slots_all [B, T + H, N, D] -> recon [B, H, 3, R, R], seg [B, H, R, R] on the current stream"""
        sl = slots_all[:, self.T:].reshape(self.B * self.H, self.N, self.D)
        engine.savi_decode(self.decoder, sl, ws_slot=self._key + ('dec', ), want=('seg', ), seg_dtype=self.seg_dtype,
                           out_recon=recon.view(self.B * self.H, *recon.shape[2:]), out_seg=seg.view(self.B * self.H, *seg.shape[2:]))

    @torch.no_grad()
    def run(self, imgs, noises=None, out=None, serial=False, decoded=None):
        """imgs: sequence of n device tensors [B, burn_in, 3, H, W]; noises: None or n tensors [B, burn_in, N, D]
        (the kernel noise of every frame, for reproducible runs).  Returns out [n, B, burn_in + pred_len, N, D]; the
        pipelined schedule returns when the last batch is finished (the host waits for it, see the end of this function).
        serial=True runs the same calls back to back on the calling stream (reference schedule for the tests).
        decoded: None, or a dict the decode stage fills (pipeline built with decoder=...): 'recon' [n, B, pred_len, 3, R, R] and
        'seg' [n, B, pred_len, R, R] of the PREDICTED frames (test_vp.py's `pred` / `pred_mask`)."""
        n = len(imgs)
        B = self.B
        rc_all, sg_all = self._decoded_buffers(decoded, n) if decoded is not None else (None, None)
        if decoded is not None and out is not None and not out.is_cuda:
            raise RuntimeError('slotformer_amd: the decode stage reads the slots from a device-resident `out`')
        host_in = n > 0 and not imgs[0].is_cuda
        for im in imgs:
            if tuple(im.shape[:2]) != (B, self.T) or im.is_cuda == host_in or im.dtype != torch.float32:
                raise RuntimeError(f'every batch must be a float32 tensor [{B},{self.T},3,H,W], all on the device or all in (pinned) host '
                                   f'memory, got {tuple(im.shape)} {im.dtype} on {im.device}')
        if out is None:
            out = torch.empty(n, B, self.T + self.H, self.N, self.D, device=self.dev)
        self._check_plan()
        nz = (lambda j: None) if noises is None else (lambda j: noises[j])
        cur = torch.cuda.current_stream(self.dev)
        units = self._unit_plan(n)
        if serial or n == 0:
            for u0, nb, u, _ in units:
                for h in range(nb):
                    im = imgs[u0 + h].to(self.dev, non_blocking=True) if host_in else imgs[u0 + h]
                    # (on the calling stream the encode has the whole chip: the fill graph -- one persistent convolution workgroup per CU of the
                    #  device -- not the lane's, which is sized for the encode partition)
                    whole = ('fill', 0) if (self.encode_graph and self.cu_split and self.fill_whole_chip and self.s_free and self.fill_par > 1) else 0
                    self._encode(im, nz(u0 + h), u.buf[h * B:(h + 1) * B], None, lane=whole, unit=(u, h))
                if self.split and self.chain_on != 'roll':
                    self._chain(u)
                self._rollout(u)
                for h in range(nb):
                    out[u0 + h].copy_(u.buf[h * B:(h + 1) * B], non_blocking=True)
                    if decoded is not None:
                        self._decode(out[u0 + h], rc_all[u0 + h], sg_all[u0 + h])
            self._check_seam()
            return out
        G, NU, steal = self.G, self.NU, self.steal
        lanes, rolls = self.lanes, self.roll_streams
        nl, nu = len(lanes), len(units)
        # work stealing: the features of the batches of unit u are computed behind the rollout of unit u - lead, on its stream
        lead = self.lead
        NF = lead * G + G                # feature buffers per lane (slot of batch j: j % NF)
        # during the fill of a run the rollout partition is idle (the last rollout stream until the SECOND unit is encoded):
        # it takes `fill_steal` whole time steps of convolutions off the encodes of the batches behind the fill
        fill_k = min(self.fill_steal, self.T) if (len(rolls) > 1 and self.cu_split) else 0
        kmax = max([int(math.ceil(steal)), fill_k] + list(self.pre_steal))
        if kmax and self.feat_bufs is None:
            cl = list(self.savi.enc_channels)[-1]
            self.feat_bufs = [[torch.empty(kmax, hi - lo, 64 * 64, cl, device=self.dev) for _ in range(NF)] for _, lo, hi in lanes]
        for st, _, _ in lanes:
            st.wait_stream(cur)
        for st in rolls + list(self.s_free):
            st.wait_stream(cur)
        if not out.is_cuda:
            if self._s_out is None:
                self._s_out = self._pool_stream('out')
            self._s_out.wait_stream(cur)
        ev_dec = None
        if decoded is not None:
            if self._s_dec is None:
                self._s_dec = self._pool_stream('dec')
            self._s_dec.wait_stream(cur)
            ev_dec = torch.cuda.Event()
        trace = bool(int(os.environ.get('SF_PIPE_TRACE', '0')))   # timeline of a run (tools/pipe_timeline.py): timing events everywhere
        ev_enc = [[torch.cuda.Event(enable_timing=trace) for _ in range(nl)] for _ in range(n)]
        ev_roll = [torch.cuda.Event(enable_timing=True) for _ in range(nu)]   # also: completion time of every unit
        ev_pre = [torch.cuda.Event() for _ in range(n)]
        stolen = [0] * n                 # time steps of precomputed features batch j will find in its feature buffer
        # host-resident inputs (pinned): an upload stage on its own stream runs ahead of the consumers -- far enough for the
        # work stealing, which reads the frames of a batch lead * G + G batches before its encode (extract_slots.py:19-38 reads
        # its videos from a DataLoader; this is the device side of that hand-over)
        NS = NF + 2
        ev_up, up_next = [], [0]
        pinned_in = host_in and all(im.is_pinned() for im in imgs)
        if host_in:
            if self._stage is None or self._stage[0].shape != imgs[0].shape:
                self._stage = [torch.empty(imgs[0].shape, device=self.dev) for _ in range(NS)]
                self._s_copy = self._pool_stream('copy')
                self._pin = None
            if not pinned_in and getattr(self, '_pin', None) is None:
                # pageable input: batches pass through a ring of NS page-locked staging buffers (never the whole set page-locked at once;
                # the host pays one memcpy per batch -- pre-pinned input skips it)
                self._pin = [torch.empty(imgs[0].shape, pin_memory=True) for _ in range(NS)]
            self._s_copy.wait_stream(cur)
            ev_up = [torch.cuda.Event() for _ in range(n)]

        def img_of(j, stream):
            """the frames of batch j as a device tensor `stream` may read"""
            if not host_in:
                return imgs[j]
            while up_next[0] <= min(j, n - 1):
                k = up_next[0]
                with torch.cuda.stream(self._s_copy):
                    if k >= NS:
                        # the staging slot's previous batch has been encoded (issued NS batches ago).  The HOST waits, not the copy
                        # stream: a queue parked in a cross-queue wait slows the queues that are running on this platform (the
                        # masked encode lane ran at 5.8 instead of 4.05 ms per batch with the wait on the stream)
                        for e in ev_enc[k - NS]:
                            e.synchronize()
                    src = imgs[k]
                    if not pinned_in:
                        if k >= NS:
                            ev_up[k - NS].synchronize()   # the upload out of this page-locked slot is done
                        self._pin[k % NS].copy_(src)
                        src = self._pin[k % NS]
                    self._stage[k % NS].copy_(src, non_blocking=True)
                    ev_up[k].record(self._s_copy)
                up_next[0] += 1
            stream.wait_event(ev_up[j])
            return self._stage[j % NS]

        def steal_for(jj, stream, tag, kj=None):
            """features of the first steps of batch jj on `stream` (the current stream)"""
            kj = self._steal_of(jj) if kj is None else kj
            if kj:
                im = img_of(jj, stream)
                for li, (_, lo, hi) in enumerate(lanes):
                    engine.savi_cnn(self.savi, im[lo:hi], 0, kj, out=self.feat_bufs[li][jj % NF][:kj],
                                    ws_slot=self._key + ('steal', li, tag))
            stolen[jj] = kj
            ev_pre[jj].record(stream)

        n_fill = min(self.fill_batches or units[0][1], n) if (self.cu_split and self.fill_whole_chip) else 0
        for _, _, u, _ in units:
            u.busy = None
        ev_t0 = torch.cuda.Event(enable_timing=True)
        ev_t0.record(cur)
        ev_rstart = [torch.cuda.Event(enable_timing=True) for _ in range(nu)] if trace else None
        n_free = 0
        downloads = []   # host-resident output: units rolled out whose slots are not yet on their way to the host

        def download_ready(for_unit=None, everything=False):
            """enqueue the downloads of the units whose rollout is done (for_unit: WAIT for that unit buffer's rollout -- the buffer is
            about to be encoded into again; everything: wait for all of them)"""
            for d in list(downloads):
                dui, du0, dnb, du, done = d
                if everything or du is for_unit:
                    done.synchronize()
                if done.query():
                    with torch.cuda.stream(self._s_out):
                        for h in range(dnb):
                            out[du0 + h].copy_(du.buf[h * B:(h + 1) * B], non_blocking=True)
                        ev_roll[dui].record(self._s_out)
                    downloads.remove(d)

        fill_last = {}   # fill graph index -> the last batch encoded through it
        for ui, (u0, nb, u, drain) in enumerate(units):
            download_ready(for_unit=u)
            for h in range(nb):
                j = u0 + h
                dst = u.buf[h * B:(h + 1) * B]
                hyb = (self.hybrid > 0 and j >= n_fill and n_fill > 0 and (j - n_fill) % self.hybrid == self.hybrid - 1
                       and j + self.hybrid_tail < n and len(self.s_free) > 0 and self.fill_par > 1)
                if hyb:
                    # hybrid lane: every `hybrid`-th batch behind the fill is encoded on an UNMASKED stream (fill graph 1) beside the masked
                    # lane -- the rollout partition has slack, the encode lane is the bound
                    fs = self.s_free[0]
                    if fill_last.get(1) is not None:
                        fs.wait_event(ev_enc[fill_last[1]][0])
                    if u.busy is not None:
                        fs.wait_event(u.busy)
                    with torch.cuda.stream(fs):
                        self._encode(img_of(j, fs), nz(j), dst, None, lane=('fill', 1), unit=(u, h))
                        ev_enc[j][0].record(fs)
                    fill_last[1] = j
                    last_stream = fs
                    ev_wait_j = ev_enc[j][:1]
                elif j < n_fill:
                    # pipeline fill: the first encodes take the whole chip (unmasked streams); the masked lanes start after them.  One
                    # encode alone cannot fill 256 CUs (2.7 ms for 470 CU-ms of work), so `fill_par` of them run side by side,
                    # each from its own graph / buffers (380-382 vs 375-378 k frames/s at 20 batches).  The host waits: with the
                    # other queues parked in a wait on this event the encode was measured at 4.8 instead of 3.05 ms (a queue stalled
                    # in a cross-queue wait slows the queue that is running)
                    fill_par = self.fill_par if self.s_free else 1
                    fi = j % fill_par
                    fs = cur if fi == 0 else self.s_free[(fi - 1) % len(self.s_free)]
                    if (j >= units[0][1] or self.split) and self.s_free and fi == 0:
                        # fill batches behind the first unit: a rollout is already enqueued, and the calling stream -- the legacy null
                        # stream -- would wait for it.  Fill graph 0 (its fixed buffers) moves to an unmasked stream, behind its last use
                        fs = self.s_free[-1]
                    if j >= fill_par:
                        fs.wait_event(ev_enc[j - fill_par][0])   # the previous batch through this fill graph / its buffers
                    with torch.cuda.stream(fs):
                        self._encode(img_of(j, fs), nz(j), dst, None, lane=('fill', fi) if fill_par > 1 else 0, unit=(u, h))
                        ev_enc[j][0].record(fs)
                    fill_last[fi] = j
                    last_stream = fs
                    if j == 0 and (steal or fill_k) and len(rolls) > 1:
                        # the rollout streams idle until the first unit is encoded: they compute the stolen features of the
                        # batches behind the fill now (round-robin), so that only the fill batches pay for their own convolutions
                        # (all on the LAST rollout stream, whose first unit starts latest: on the first one they delayed the first
                        #  rollout of a run by 3.3 ms, tools/pipe_timeline.py)
                        first = max(n_fill, 1)
                        fill_end = units[lead][0] if nu > lead else n      # (unit `lead` onwards: stolen behind the rollouts)
                        if fill_k:   # only what the last rollout stream can finish before its first unit is ready: the second unit's batches
                            fill_end = min(fill_end, units[2][0] if nu > 2 else n)
                        for jj in range(first, fill_end):
                            with torch.cuda.stream(rolls[-1]):
                                steal_for(jj, rolls[-1], len(rolls) - 1, fill_k or None)
                    if j == n_fill - 1:
                        for jj in range(max(0, n_fill - fill_par), n_fill):
                            ev_enc[jj][0].synchronize()
                        for st, _, _ in lanes:
                            for jj in range(max(0, n_fill - fill_par), n_fill):
                                st.wait_event(ev_enc[jj][0])
                    ev_wait_j = ev_enc[j][:1]
                else:
                    for li, (st, lo, hi) in enumerate(lanes):
                        with torch.cuda.stream(st):
                            if u.busy is not None:
                                st.wait_event(u.busy)   # the unit buffer is free once its previous rollout + copy-out are done
                            pre = None
                            if stolen[j]:
                                st.wait_event(ev_pre[j])
                                pre = self.feat_bufs[li][j % NF][:stolen[j]]
                            self._encode(img_of(j, st), nz(j), dst, pre, lo, hi, li, unit=(u, h))
                            ev_enc[j][li].record(st)
                    last_stream = lanes[0][0]
                    ev_wait_j = ev_enc[j]
                if h == 0:
                    ev_wait = []
                ev_wait = ev_wait + list(ev_wait_j)
            if self.split and self.chain_on != 'roll':
                # the slot branch of the unit on the stream that encoded its last batch, behind the features of all its batches
                cst = last_stream
                with torch.cuda.stream(cst):
                    for e in ev_wait:
                        cst.wait_event(e)
                    self._chain(u)
                    ev_c = torch.cuda.Event()
                    ev_c.record(cst)
                ev_wait = [ev_c]
            s_roll = rolls[ui % len(rolls)]
            if len(rolls) > 1 and drain and self.s_free:
                # drain: the encode lane is (about to be) idle -- the last units take unmasked streams (all CUs) instead of queueing
                # behind the full units on the rollout partition (their slot / feature buffers are ordered by events)
                s_roll = self.s_free[n_free % len(self.s_free)]
                n_free += 1
            with torch.cuda.stream(s_roll):
                for e in ev_wait:
                    s_roll.wait_event(e)
                if self.pre_steal and ui + 1 < nu and self.cu_split and not drain:
                    # the rollout streams have a little slack per unit (the encode is the longer side): before this unit rolls
                    # out, its stream takes pre_steal[h] time steps of convolutions off the LATER batches of the NEXT unit --
                    # stealing behind a rollout is too late for units of several batches (the next unit is encoded meanwhile)
                    nu0, nnb = units[ui + 1][:2]
                    for h2 in range(nnb):
                        k2 = self.pre_steal[min(h2, len(self.pre_steal) - 1)]
                        if k2 and not stolen[nu0 + h2]:
                            steal_for(nu0 + h2, s_roll, ui % len(rolls), k2)
                if trace:
                    ev_rstart[ui].record(s_roll)
                self._rollout(u)
                if out.is_cuda:
                    for h in range(nb):
                        out[u0 + h].copy_(u.buf[h * B:(h + 1) * B])
                    ev_roll[ui].record(s_roll)
                    if decoded is not None:
                        # decode stage: reads the unit's slots from `out` (the unit buffer is free for the next encode at once), on an
                        # unmasked stream of its own, one batch after the other
                        with torch.cuda.stream(self._s_dec):
                            self._s_dec.wait_event(ev_roll[ui])
                            for h in range(nb):
                                self._decode(out[u0 + h], rc_all[u0 + h], sg_all[u0 + h])
                            ev_dec.record(self._s_dec)
                else:
                    # pinned host output: the downloads run on a torch-owned stream -- PyTorch's host allocator records an event
                    # on every stream a pinned block was used on when the block is freed, and the CU-masked streams of this
                    # object may be gone by then (close())
                    # The download stream must not sit PARKED in a wait for the unit's rollout (a queue stalled in a cross-queue
                    # wait slows the queues that are running on this platform: the masked encode lane ran at 5.7 instead of 4.05 ms
                    # per batch, 312 vs 390 k frames/s) -- the host enqueues the copies once the rollout is done (download_ready)
                    done = torch.cuda.Event()
                    done.record(s_roll)
                    downloads.append((ui, u0, nb, u, done))
                u.busy = ev_roll[ui]
                if steal and ui + lead < nu:
                    # their feature buffers were last read by the encodes of batches <= u0 + nb - 1 ... (NF = lead*G + G apart),
                    # which this stream has waited for
                    tu0, tnb = units[ui + lead][:2]
                    for jj in range(tu0, tu0 + tnb):
                        steal_for(jj, s_roll, ui % len(rolls))
        # The host waits for the last units HERE, before the calling stream is made to wait for the pipeline's streams: a
        # wait that sits pending on the calling stream (PyTorch's default stream is the legacy null stream) for the whole
        # run was measured to slow the kernels of the masked encode lane that shares shader engines with the rollout by
        # 30 % (7.7 instead of 6.4 ms per batch, tools/lane_probe.py COPY=1) -- so run() returns when the results are done.
        download_ready(everything=True)
        for e in ev_roll[-(len(rolls) + 4):]:   # (the drain units on the unmasked streams may overtake the units before them)
            e.synchronize()
        if decoded is not None:
            ev_dec.synchronize()   # (the last decode: the stage runs in order on one stream)
            cur.wait_stream(self._s_dec)
        for st, _, _ in lanes:
            cur.wait_stream(st)
        for st in rolls + list(self.s_free) + ([self._s_copy] if host_in else []) + ([self._s_out] if not out.is_cuda else []):
            cur.wait_stream(st)
        self.completion_events = ev_roll
        self.completion_batches = [nb for _, nb, _, _ in units]
        if trace:
            torch.cuda.synchronize(self.dev)
            self.timeline = {'encode_end_ms': [max(ev_t0.elapsed_time(e) for e in ev_enc[j][:1 if j < n_fill else nl]) for j in range(n)],
                             'rollout_start_ms': [ev_t0.elapsed_time(e) for e in ev_rstart],
                             'rollout_end_ms': [ev_t0.elapsed_time(e) for e in ev_roll],
                             'units': [(u0, nb) for u0, nb, _, _ in units]}
        self._check_seam()
        return out

    def _check_seam(self):
        """Seam launches (only when the rollout options turn them on) hand rows over inside a launch with a bounded wait; a
        consumer that gave up has poisoned its outputs with NaN and counted itself -- raise instead of returning them."""
        o = self.rollout_opts
        if o is not None and o.seam_fused == 0:
            return
        if self._lib.sf_rollout_uses_seam_opts(C.byref(self._plan.struct), self.G * self.B, None if self.rollout_opts is None else C.byref(self.rollout_opts)):
            t = self._lib.sf_seam_timeouts()
            if t != getattr(self, '_seam_seen', 0):
                self._seam_seen = t
                raise RuntimeError('slotformer_amd: a seam hand-off of the rollout timed out (a producer workgroup was not resident); '
                                   'the affected slots are NaN -- run the pipeline with rollout_opts={"seam": False}')
