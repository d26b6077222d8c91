import sys, time, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, bench
for th in (16, 32, 64):
    torch.set_num_threads(th)
    os.environ['SF_CPU_THREADS'] = str(th)
    t0 = time.time()
    import golden_util as gu, oracle
    scfg, rcfg = bench.c2_configs()
    from slotformer_amd.base_slots import build_model
    from slotformer_amd.video_prediction.models import SlotRollouter
    torch.manual_seed(0)
    savi = build_model(gu.ParamsView(scfg)); roll = SlotRollouter(**rcfg['rollout_dict'])
    ssd = {k: v.detach() for k, v in savi.state_dict().items()}
    rsd = {'rollouter.' + k: v.detach() for k, v in roll.state_dict().items()}
    img = bench.synthetic_img(4); noise = torch.randn(4, 6, 7, 128)
    with torch.no_grad():
        for rep in range(2):
            t1 = time.time()
            post = oracle.savi_encode(img, ssd, scfg, noise=noise)['post_slots']
            t2 = time.time()
            oracle.rollouter_forward(post, 50, rsd, rcfg['rollout_dict'])
            t3 = time.time()
            print(th, 'threads: encode %.2fs rollout %.2fs' % (t2-t1, t3-t2), flush=True)
