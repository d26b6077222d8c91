import faulthandler, sys, os
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import golden_util as gu
from slotformer_amd import harness
from slotformer_amd.base_slots import build_model
from slotformer_amd.video_prediction.models import SlotRollouter
dev = torch.device('cuda:0')
def P(*a):
    print(*a, flush=True)
torch.manual_seed(5)
savi = build_model(gu.ParamsView(gu.C2_SAVI)).eval().to(dev); savi.testing = True
roll = SlotRollouter(**gu.C2_ROLL['rollout_dict']).eval().to(dev)
T, H, bs, V = 6, 8, 4, 14
rs = np.random.RandomState(3)
videos = torch.from_numpy((rs.rand(V, T, 3, 128, 128) * 2 - 1).astype(np.float32))
noises = torch.from_numpy(rs.standard_normal((V, T, 7, 128)).astype(np.float32))
with torch.no_grad():
    P('host in'); out = harness.extract_and_rollout(savi, roll, videos, H, batch_size=bs, noises=noises); torch.cuda.synchronize(); P('ok', out.shape)
    P('no noise'); out_r = harness.extract_and_rollout(savi, roll, videos, H, batch_size=bs); torch.cuda.synchronize(); P('ok')
    P('dev in'); out_d = harness.extract_and_rollout(savi, roll, videos.to(dev), H, batch_size=bs, noises=noises); torch.cuda.synchronize(); P('ok', torch.equal(out_d, out))
    P('to host'); out_h = harness.extract_and_rollout(savi, roll, videos, H, batch_size=bs, noises=noises, to_host=True); P('ok', torch.equal(out_h, out.cpu()))
    P('release'); harness.release_pipelines(); P('ok')
P('del'); del out_h, out_d, out_r, out; P('ok')
import gc; gc.collect(); P('gc ok')
