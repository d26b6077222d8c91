"""Build libslotformer_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake.

    python -m slotformer_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, 'build')
LIB = os.path.join(HERE, 'libslotformer_hip.so')
SOURCES = ['gemm.hip', 'slot_attn.hip', 'slot_update_mfma.hip', 'slot_update_wide.hip', 'slot_chain.hip', 'pred_step.hip', 'slot_attn_bwd.hip', 'slot_attn_train.hip', 'elementwise.hip', 'attn_fused.hip', 'conv_halo.hip', 'conv_rows4.hip', 'conv_ws.hip', 'deconv_s2.hip', 'conv_first.hip', 'pixel_mlp.hip', 'layer_fused.hip', 'attn_rows.hip', 'ffn_tile.hip', 'layer_tok.hip', 'rollout_train.hip', 'savi_decode_train.hip', 'savi_features_train.hip', 'steve_decoder.hip', 'slate_attn_bwd.hip', 'engine.hip', 'host_twins.hip', 'sf_runtime.cpp']
# every header of csrc/ (kernel bodies shared between translation units live in headers too: slot_update_body.h, stream_mfma.h) + the public one
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join('..', '..', 'include', 'slotformer_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


# per-file flags: the SLP vectoriser packs the LayerNorm sums of layer_tok.hip into v_pk_add_f32 behind dozens of register moves (spills)
FILE_FLAGS = {'layer_tok.hip': ['-fno-slp-vectorize']}


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_tree_hash():
    """sha256 over the sources the library is built from (names + contents, in SOURCES / HEADERS order): what a committed
    rocprof summary stores as `source_tree`, so that bench.py can tell whether its trace still describes these kernels."""
    import hashlib
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        path = os.path.join(CSRC, name)
        h.update(os.path.basename(name).encode())
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    hipcc = _hipcc()
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, os.path.splitext(s)[0] + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + FILE_FLAGS.get(s, []) + (['-x', 'hip'] if s.endswith('.cpp') else []) + ['-c', src, '-o', obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(' '.join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + ' '.join(cmd) + '\n' + r.stdout + r.stderr)
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB])
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
