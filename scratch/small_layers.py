"""Which conv launches of one eager step run below 150 TFLOP/s?  (shapes + ms, for the tiny-channel layers)"""
import sys, collections, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import master_thesis_amd as mt
from master_thesis_amd import _lib as L, engine as E
from master_thesis_amd.synthetic import make_batch
which = sys.argv[1] if len(sys.argv) > 1 else 'dfpn'
dev = torch.device('cuda', 0)
kw = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)
torch.manual_seed(0)
vgg = mt.VGGFeatures(device=dev)
model = mt.DFPN(vgg, **kw).to(dev) if which == 'dfpn' else mt.CHN(vgg, None, mt.DFPN(vgg, **kw), **kw).to(dev)
opt = model.configure_optimizers()['optimizer']
batch = make_batch(8, 5, 256, 256, dev, seed=1234)
def step():
    opt.zero_grad(); loss = model.training_step(batch, 0); loss.backward(); opt.step()
step(); step()
log = []
orig = L.call
def call(name, *a):
    if name in ('mt_conv_run', 'mt_wgrad_run', 'mt_conv_tiny_in_run', 'mt_conv_tiny_out_run', 'mt_wgrad_tiny_run'):
        d = a[0]._obj
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(name, *a); e1.record()
        if name in ('mt_conv_run', 'mt_conv_tiny_in_run', 'mt_conv_tiny_out_run'):
            log.append((name, (d.x.N, d.x.H, d.x.W, d.x.C, d.y.H, d.y.W, d.y.C, d.KH, d.stride, d.transposed, d.ups), e0, e1))
        else:
            log.append((name, (d.d.N, d.d.H, d.d.W, d.d.C, d.g.H, d.g.W, d.g.C, d.KH, d.stride, 0, d.ups), e0, e1))
        return r
    return orig(name, *a)
L.call = call
E.L.call = call
step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, shp, e0, e1 in log:
    a = agg.setdefault((name, shp), [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = 0
for (name, shp), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    N, H, W, C, Ho, Wo, Co, K, st, tr, ups = shp
    fl = 2.0 * N * (Ho if 'conv' in name else H) * (Wo if 'conv' in name else W) * C * Co * K * K
    tf = fl * n / ms / 1e9
    if min(C, Co) <= 16:
        tot += ms
        byts = 4.0 * N * (H * W * C + Ho * Wo * Co)
        print(f'{name:20s} x{n:2d} N{N} {H}x{W}x{C} -> {Ho}x{Wo}x{Co} k{K} s{st} t{tr} u{ups}: {ms:7.3f} ms total, {ms / n * 1e3:7.1f} us each, {tf:6.1f} TF(padded), {byts * n / ms / 1e6:7.0f} GB/s')
print('tiny-channel total ms:', tot)
