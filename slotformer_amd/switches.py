"""Every SF_* environment variable the library, the pipeline and bench.py read, with what it does.  bench.py refuses to run with an SF_* variable that is
not in this table and prints the ones that are set in `config.switches` of its line; INTEGRATION.md section 5 is generated from it
(python -m slotformer_amd.switches).  kind: 'product' = selects a documented behaviour; 'bench' = bench.py / tools only; 'probe' = an experiment kept
for measurements (its default is what ships; the combinations are not tested); 'debug' = in-kernel time stamps."""

SWITCHES = {
    # ---- product: arithmetic and kernel forms ----
    'SF_PRECISION': ('product', "process default of the matrix arithmetic: 'bf16x3' (default; split-bf16, fp32-equivalent) | 'f32' (exact-f32 MFMA)"),
    'SF_LAYER_TOK': ('product', "process default of sf_rollout_opts.layer_tok for DIRECT rollout calls (0, default: the row-tile / latency forms; 1: the layers before the last as token-stationary launches). The pipeline passes its own choice per unit"),
    'SF_SEAM_FUSED': ('product', "process default of the seam launches of the latency forms (1; 0: two launches instead)"),
    'SF_CONV_FP16X2': ('product', "OPT-IN two-product fp16 form of the 5x5 convolutions (weights rounded to fp16: 3-6e-5 from the fixtures instead of 1e-5); default 0"),
    'SF_ENCODE_FORK': ('product', "direct encode calls as two branches on two streams (1, default) or one stream (0)"),
    # ---- product: the batch pipeline ----
    'SF_PIPE_TOK': ('product', "0: the pipeline never uses the token-stationary layer launches (every unit bit-identical to the serial module calls)"),
    'SF_PIPE_GROUP': ('product', "batches per rollout unit (default: 6 for token-stationary units of 32-video batches, else 4 / unit_batches_for)"),
    'SF_PIPE_HYBRID': ('product', "every k-th batch behind the fill is encoded on an unmasked stream beside the CU-masked lane (default 4 with token-stationary units, 3 / 5 / 0 otherwise)"),
    'SF_PIPE_HYBRID_TAIL': ('probe', "no hybrid-lane encode among the last k batches of a run (default min(hybrid, 3))"),
    'SF_PIPE_FILL': ('product', "batches encoded on the whole chip at the start of a run (default: three units' worth)"),
    'SF_PIPE_FILL_WS': ('probe', "workgroups of the weights-stationary convolution inside the whole-chip fill / hybrid encode graphs (default: the device's CU count in pipelines with token-stationary units, else 0 = the 4-row tiles)"),
    'SF_PIPE_FILL_PAR': ('probe', "whole-chip encodes side by side during the fill (default 2)"),
    'SF_PIPE_CU_SPLIT': ('product', "CU mask of the encode partition, e.g. rows4 (default: sized from the two sides' CU time)"),
    'SF_PIPE_ENCODE_GRAPH': ('product', "replay the encode of a batch from a hipGraph (1, default) or launch it eagerly (0)"),
    'SF_PIPE_TRACE': ('bench', "record the device timeline of a run (tools/pipe_timeline.py)"),
    'SF_PIPE_SIZES': ('probe', "explicit unit plan of a run, e.g. 6,6,6,2 (tools/sweep_tok.sh)"),
    'SF_PIPE_DRAIN_UNITS': ('probe', "units at the end of a run that take unmasked streams (default: 2 when a run of token-stationary units ends in a short unit, else 1)"),
    'SF_PIPE_DRAIN_LAT': ('probe', "the last full-size unit of a run in the latency forms (default: units below 128 videos)"),
    'SF_PIPE_STEAL': ('probe', "time steps of convolutions per batch computed on the rollout streams (default 0 for the pair partition)"),
    'SF_PIPE_PRE_STEAL': ('probe', "convolution steps of the next unit computed in front of a rollout (default none; measured neutral)"),
    'SF_PIPE_FILL_STEAL': ('probe', "stolen steps during the fill (default 0)"),
    'SF_PIPE_RAMP': ('probe', "ever smaller units at the end of a run (default 0; measured slower)"),
    'SF_PIPE_SPREAD': ('probe', "0: no splitting of the remainder between the last two units"),
    'SF_PIPE_SEAM': ('probe', "seam launches inside pipeline units (default 0)"),
    'SF_PIPE_FFN_ROWS': ('probe', "rows per FFN workgroup of the chunk-partial form inside pipeline units"),
    'SF_PIPE_ATTN_HEADS': ('probe', "heads per attention workgroup inside pipeline units"),
    'SF_PIPE_ATTN_ROWS': ('probe', "0: no row-tile attention form inside pipeline units"),
    'SF_PIPE_FFN_TILE': ('probe', "0 / 1 / 2: FFN tile form inside pipeline units"),
    'SF_PIPE_ENCODE_FORK': ('probe', "encode graphs with two branches inside the pipeline (default 0; measured slower there)"),
    'SF_PIPE_PLACEMENT': ('probe', "'measure': calibrate which streams the unmasked lanes take instead of the rule (default 'rule'; same throughput)"),
    'SF_PIPE_LOG_PLACEMENT': ('debug', "print the stream placement"),
    'SF_PIPE_FREE_SKIP': ('probe', "pool streams parked in front of the unmasked ones (default 1, 0 with a process group)"),
    'SF_PIPE_FREE_MASKED': ('probe', "unmasked lanes as full-mask CU-masked streams (default 0)"),
    'SF_PIPE_FREE_DUMMY': ('probe', "extra dummy streams created first (default 0)"),
    'SF_PIPE_STREAM_POOL': ('probe', "0: private CU-masked streams per pipeline object instead of the process-wide pool"),
    'SF_PIPE_UPLOAD_WAIT': ('probe', "'stream': the copy stream waits for a staging slot on the device instead of the host"),
    # ---- bench.py ----
    'SF_BENCH_GROUP': ('bench', "batches per rollout unit of the bench pipeline"),
    'SF_BENCH_ENC_GROUP': ('bench', "batches handed to the pipeline as one (default encode_group_for)"),
    'SF_BENCH_CU_SPLIT': ('bench', "encode CU mask of the bench pipeline (hex word or rows<R>)"),
    'SF_BENCH_PARTITION': ('bench', "'pair' (default) | 'three' | 'two' | 'none'"),
    'SF_BENCH_STEAL': ('bench', "steal_steps of the bench pipeline"),
    'SF_BENCH_FORCE_DIST': ('bench', "1: form the RCCL process group with one rank too (the multi-GPU path on one GPU)"),
    'SF_BENCH_SELF_LAUNCH': ('bench', "1: take the torch.distributed.run self-launch path with --gpus 1"),
    'SF_BENCH_LIVE_EVERY': ('bench', "bracket every n-th conv / Slot-Attention launch with events in the untimed live pass"),
    'SF_BENCH_LIVE_MASK': ('bench', "kernel classes of that pass"),
    'SF_BENCH_ALLOW_STALE_TRACE': ('bench', "1: use the committed rocprof summary although the kernel sources changed"),
    'SF_LIB_PATH': ('bench', "load another build of libslotformer_hip.so (tools: -D variants of a kernel)"),
    'SF_EXTRA_FLAG': ('bench', "one extra hipcc flag for slotformer_amd.build"),
    # ---- probes inside the library (defaults ship) ----
    'SF_ENC_FUSE_NEXT': ('probe', "0: the slot prologue of the next time step as its own launch instead of the tail of the slot update"),
    'SF_ENC_INTERLEAVE': ('probe', "1: slot updates riding in front of convolution tiles (measured +0.2 %)"),
    'SF_SA_TILE': ('probe', "0: the two-pass Slot-Attention kernel instead of the one-pass tile kernel"),
    'SF_SA_ONEPASS': ('probe', "1: the single-shot one-pass Slot-Attention kernel (measured slower)"),
    'SF_PIXEL_MLP_STREAM': ('probe', "0: the tile-at-a-time per-pixel chain"),
    'SF_PIXEL_TILE': ('probe', "pixels per workgroup of the per-pixel chain (64 default / 128)"),
    'SF_PIXEL_PIX': ('probe', "pixel tile of the 192-wide chain"),
    'SF_QKV_TILE_ROWS': ('probe', "128: one weight stream per 128 rows in the q|k|v row-tile kernel"),
    'SF_CONV_WS': ('probe', "0: the 4-row-tile convolution everywhere (default: the weights-stationary kernel on streams with CUs of their own)"),
    'SF_CONV_HALO': ('probe', "0: the generic implicit-GEMM convolution instead of the halo / row-tile kernels"),
    'SF_CONV_FIRST': ('probe', "0: the generic path for the first convolution"),
    'SF_CONV_CFG': ('probe', "tile configuration of the implicit-GEMM convolution"),
    'SF_GEMM_CFG': ('probe', "force a GEMM tile configuration (tools/gemm_bench.py)"),
    'SF_DECONV_CLASSES': ('probe', "0: the single-launch gather form of the transposed convolutions"),
    'SF_TRAIN_ATTN': ('probe', "'scalar': plain-FMA attention kernels in training"),
    'SF_ATTN_BWD48': ('probe', "0: the generic attention backward at head width 48"),
    'SF_WGRAD_WINDOW': ('probe', "0: the weight-gradient kernel without its LDS window"),
    # ---- in-kernel time stamps ----
    'SF_LF_DBG': ('debug', "phase stamps of the fused-layer / row-tile kernels (tools/lf_phase_probe.py)"),
    'SF_LT_DBG': ('debug', "phase stamps of the token-stationary layer kernel (tools/layer_tok_probe.py)"),
    'SF_CONV_DBG': ('debug', "phase stamps of the row-tile convolution"),
    'SF_DECONV_DBG': ('debug', "phase stamps of the transposed convolution"),
    'SF_SA_DBG': ('debug', "phase stamps of the Slot-Attention tile kernel"),
    'SF_GEMM_DBG': ('debug', "print the GEMM configuration picked per shape"),
}


def check_environment(environ=None):
    """(set, unknown): the SF_* variables that are set, and those among them this table does not know"""
    import os
    env = os.environ if environ is None else environ
    have = {k: env[k] for k in sorted(env) if k.startswith('SF_')}
    return have, [k for k in have if k not in SWITCHES]


def markdown_table():
    rows = ['| variable | kind | effect |', '|---|---|---|']
    for k, (kind, text) in SWITCHES.items():
        rows.append(f'| `{k}` | {kind} | {text} |')
    return '\n'.join(rows)


if __name__ == '__main__':
    print(markdown_table())
