# round 4: soak of the replayed training steps at the benchmark configuration -- does the loss of a FIXED synthetic batch fall,
# do the device-side amax scalars keep up with drifting weights, does every loss stay finite, is the step time flat?
import sys, time, json
sys.path.insert(0, '.')
import torch
import master_thesis_amd as mt
from master_thesis_amd.graph import GraphedTrainStep
from master_thesis_amd.synthetic import make_batch
dev = torch.device('cuda', 0)
mt._lib.load()
kw = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)
torch.manual_seed(0)
vgg = mt.VGGFeatures(device=dev)
dfpn = mt.DFPN(vgg, **kw).to(dev)
batch = make_batch(8, 5, 256, 256, dev, seed=1234)
def soak(model, tag, steps, every=25):
    opt = model.configure_optimizers()['optimizer']
    g = GraphedTrainStep(model, opt, batch, include_optimizer=True)
    losses, times = [], []
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(steps):
        l = g.step()
        if i % every == 0 or i == steps - 1:
            torch.cuda.synchronize()
            times.append((i, time.time() - t0))
            losses.append((i, float(l)))
            print(tag, 'step', i, 'loss %.6f' % losses[-1][1], 'elapsed %.2f s' % times[-1][1], flush=True)
    flat = torch.cat([p.detach().reshape(-1) for p in model.trainable_parameters()])
    ok = all(l == l and abs(l) < 1e30 for _, l in losses) and bool(torch.isfinite(flat).all())
    ms = [(times[k + 1][1] - times[k][1]) / (times[k + 1][0] - times[k][0]) * 1e3 for k in range(len(times) - 1)]
    print(tag, json.dumps({'first': losses[0][1], 'last': losses[-1][1], 'min': min(l for _, l in losses), 'finite': ok,
                           'ms_per_step_by_window': [round(m, 2) for m in ms]}), flush=True)
    del g, opt
soak(dfpn, 'dfpn', int(sys.argv[1]) if len(sys.argv) > 1 else 400)
torch.cuda.empty_cache()
chn = mt.CHN(vgg, None, dfpn, **kw).to(dev)
soak(chn, 'chn', int(sys.argv[2]) if len(sys.argv) > 2 else 300)
