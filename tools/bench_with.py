"""bench.py with library process defaults set first (A/B of a kernel form on one box):  python tools/bench_with.py sa_planes=0 [slot_chain=1] -- <bench.py arguments>"""
import os
import runpy
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
torch.cuda.init()
from slotformer_amd import _lib  # noqa: E402
i = sys.argv.index('--')
lib = _lib.lib()
for kv in sys.argv[1:i]:
    k, v = kv.split('=')
    {'sa_planes': lib.sf_set_slot_attn_planes, 'slot_chain': lib.sf_set_slot_chain, 'pixel_tok': lib.sf_set_pixel_tok}[k](int(v))
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[i + 1:]
runpy.run_path(sys.argv[0], run_name='__main__')
