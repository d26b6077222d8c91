"""Every SF_* environment variable the library, the pipeline and bench.py read, with what it does (fifteen; the round-4 tree had seventy).  bench.py
refuses to run with an SF_* variable that is not in this table and prints the ones that are set in `config.switches` of its line; INTEGRATION.md
section 5 is generated from it (python -m slotformer_amd.switches).  Everything else that used to be an environment probe is either gone with the form
it selected (measured dead: profiles/r02..r05_probes.txt) or a keyword argument / command-line flag of the object it belongs to
(EncodeRolloutPipeline(...), bench.py --group / --cu-split / --partition / --steal / --force-dist ...)."""
import os

SWITCHES = {
    # ---- arithmetic and kernel forms (process defaults; per call: sf_rollout_opts / engine.rollout_opts) ----
    'SF_PRECISION': ('product', "process default of the matrix arithmetic: 'bf16x3' (default; split-bf16, fp32-equivalent) | 'f32' (exact-f32 MFMA) | 'bf16' (single pass)"),
    'SF_LAYER_TOK': ('product', "process default of sf_rollout_opts.layer_tok for DIRECT rollout calls (0, default: the row-tile / latency forms; 1: the layers before the last as token-stationary launches). The pipeline passes its own choice per unit"),
    'SF_SEAM_FUSED': ('product', "process default of the seam launches of the latency forms (1; 0: two launches instead)"),
    'SF_CONV_FP16X2': ('product', "OPT-IN two-product fp16 form of the 5x5 convolutions (weights rounded to fp16: 3-6e-5 from the fixtures instead of 1e-5); default 0"),
    'SF_CONV_WS': ('product', "0: the 4-row-tile convolution everywhere (default 1: the weights-stationary kernel on streams with CUs of their own; the same bits)"),
    'SF_ENCODE_FORK': ('product', "direct encode calls as two branches on two streams (1, default) or one stream (0)"),
    # ---- the batch pipeline ----
    'SF_PIPE_TOK': ('product', "0: the pipeline never uses the token-stationary layer launches (every unit bit-identical to the serial module calls)"),
    'SF_PIPE_GROUP': ('product', "batches per rollout unit (default: 6 for token-stationary units of 32-video batches, else 4 / unit_batches_for)"),
    'SF_PIPE_HYBRID': ('product', "every k-th batch behind the fill is encoded on an unmasked stream beside the CU-masked lane (default 8 with token-stationary units, 3 / 5 / 0 otherwise)"),
    'SF_PIPE_FILL': ('product', "batches encoded on the whole chip at the start of a run (default: three units' worth)"),
    'SF_PIPE_CU_SPLIT': ('product', "CU mask of the encode partition, e.g. rows4 (default: sized from the two sides' CU time)"),
    'SF_PIPE_ENCODE_GRAPH': ('product', "replay the encode of a batch from a hipGraph (1, default) or launch it eagerly (0)"),
    # ---- tools ----
    'SF_PIPE_TRACE': ('tools', "record the device timeline of a run (tools/pipe_timeline.py)"),
    'SF_LIB_PATH': ('tools', "load another build of libslotformer_hip.so (tools/build_variant.sh: -D variants of a kernel)"),
    'SF_DBG': ('tools', "in-kernel time stamps and tuning overrides of the probes, comma separated: conv, lt, lf=<bits>, deconv, gemm, gemmcfg=<id>, convcfg=<id>, sab1 / flash1 (the packed-word attention backward / exact-f32 flash forward the bf16-plane kernels replaced), sabts (tools/*_probe.py)"),
}


def check_environment(environ=None):
    """(set, unknown): the SF_* variables of the environment that are in the table (name -> value), and the ones that are not."""
    environ = os.environ if environ is None else environ
    seen = {k: v for k, v in environ.items() if k.startswith('SF_')}
    return {k: v for k, v in seen.items() if k in SWITCHES}, sorted(k for k in seen if k not in SWITCHES)


def markdown_table():
    rows = ['| variable | kind | what it does |', '|---|---|---|']
    rows += [f'| `{k}` | {kind} | {text} |' for k, (kind, text) in SWITCHES.items()]
    return '\n'.join(rows)


if __name__ == '__main__':
    print(markdown_table())
