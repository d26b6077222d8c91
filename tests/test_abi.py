"""CPU: the C-ABI library loads without a GPU and exports exactly what include/slotformer_hip.h
declares; the ctypes signature table covers every declaration."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'slotformer_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(sf_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from slotformer_amd import _lib
    lib = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in the header but not exported'
    assert set(syms) == set(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)
    assert lib.sf_version() >= 100
    assert isinstance(lib.sf_last_error_string(), bytes)


def test_struct_layouts_match_header():
    """Field order/count of the ctypes mirrors vs the typedefs in the header."""
    from slotformer_amd import _lib
    txt = open(os.path.join(ROOT, 'include', 'slotformer_hip.h')).read()

    def fields(name):
        end = txt.index('} ' + name + ';')
        body = txt[txt.rindex('typedef struct {', 0, end) + len('typedef struct {'):end]
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        out = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            decl = re.sub(r'^(const\s+)?(float|int|void|sf_tfm_layer)\s*', '', decl)
            for f in decl.split(','):
                out.append(re.sub(r'[\*\s]|\[\d+\]', '', f))
        return out

    for cls, name in ((_lib.sf_tfm_layer, 'sf_tfm_layer'), (_lib.sf_rollouter, 'sf_rollouter'),
                      (_lib.sf_savi_encoder, 'sf_savi_encoder'), (_lib.sf_savi_decoder, 'sf_savi_decoder')):
        assert [f[0] for f in cls._fields_] == fields(name), name
    assert ctypes.sizeof(_lib.sf_tfm_layer) == 17 * 8


def test_argument_errors_without_gpu():
    """Argument validation happens before any device work, so it is testable on CPU."""
    from slotformer_amd import _lib
    lib = _lib.lib()
    assert lib.sf_linear_f32(None, 4, None, None, None, None, 1e-5, None, 4, None, 4, 1, 4, 4, 0, None) < 0
    assert b'null pointer' in lib.sf_last_error_string()
    assert lib.sf_slot_attn_num_partials(4096) == 16
    assert lib.sf_rollout_workspace_bytes(None, 4) == 0


def test_training_entry_points_reject_bad_arguments():
    """Argument errors of the row-N1 entry points are negative return codes with a message, raised before any HIP call
    (so this runs without a GPU): the product path fails loudly instead of computing something else."""
    import ctypes as C
    from slotformer_amd import _lib
    lib = _lib.lib()

    def err():
        return lib.sf_last_error_string().decode()

    one = C.c_void_p(16)   # a non-null dummy pointer: every case below is rejected before it is dereferenced
    # linear backward: widths must be multiples of 64
    assert lib.sf_linear_bwd_f32(one, one, None, one, None, one, None, 8, 100, 64, 0, one, 1 << 30, None) < 0
    assert 'multiples of 64' in err()
    # Adam: the step count is 1-based
    assert lib.sf_adam_flat_f32(one, one, one, one, 10, 0, 1e-3, 0.9, 0.999, 1e-8, None) < 0
    assert 'Adam' in err()
    # dropout probability range
    assert lib.sf_dropout_f32(one, None, one, 8, 1.0, 0, None) < 0
    assert 'dropout_p' in err()
    # Slot-Attention backward: unsupported slot size, too many slots
    assert lib.sf_slot_attn_iter_bwd_f32(one, one, 100, 409600, one, one, one, 1, one, one, one, 0, one, 1, 4096, 7, 100, 0.1, 1e-6, one,
                                         1 << 30, None) < 0
    assert 'slot_size' in err()
    assert lib.sf_slot_attn_iter_bwd_f32(one, one, 128, 524288, one, one, one, 1, one, one, one, 0, one, 1, 4096, 9, 128, 0.1, 1e-6, one,
                                         1 << 30, None) < 0
    # rollout training: workspace query rejects a model it cannot train (no layers) by returning 0 bytes
    m = _lib.sf_rollouter()
    assert lib.sf_rollout_train_workspace_bytes(C.byref(m), 2, 3) == 0
    # precision modes
    assert lib.sf_set_precision(3) < 0 and 'precision' in err()
    old = lib.sf_get_precision()
    assert lib.sf_set_precision(2) == 0 and lib.sf_get_precision() == 2
    assert lib.sf_set_precision(old) == 0
