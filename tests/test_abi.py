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
    assert ctypes.sizeof(_lib.sf_tfm_layer) == 16 * 8


def test_argument_errors_without_gpu():
    """Argument validation happens before any device work, so it is testable on CPU."""
    from slotformer_amd import _lib
    lib = _lib.lib()
    assert lib.sf_linear_f32(None, 4, None, None, None, None, 1e-5, None, 4, None, 4, 1, 4, 4, 0, None) < 0
    assert b'null pointer' in lib.sf_last_error_string()
    assert lib.sf_slot_attn_num_partials(4096) == 16
    assert lib.sf_rollout_workspace_bytes(None, 4) == 0
