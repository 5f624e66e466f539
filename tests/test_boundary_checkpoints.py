"""CPU-only checks of the checkpoint entry points the reference's ``__main__`` calls
(``__main__.py:33-35,38-42,52-54``: ``DFPN.load_from_checkpoint`` / ``CHN.load_from_checkpoint`` on
Lightning-format files; ``model_cpn.py:93-118``: ``CPN.init_model_with_state`` on a ``module.``-prefixed
file) and of the optimizer state of such a checkpoint, whose ``torch.optim.Adam`` state is numbered
over ALL parameters the reference hands to Adam -- the frozen VGG tensors among them."""
import pytest
import torch

import seeded as S

KW = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)


def _lightning_file(path, model, optimizer_state=None):
    """What pl.Trainer writes: the module's state_dict under 'state_dict' next to trainer bookkeeping."""
    ck = {'epoch': 3, 'global_step': 1234, 'pytorch-lightning_version': '1.5.2',
          'state_dict': {k: v.clone() for k, v in model.state_dict().items()},
          'optimizer_states': [optimizer_state] if optimizer_state is not None else [],
          'lr_schedulers': [{'step_size': 50, 'gamma': 0.5, 'last_epoch': 3, '_step_count': 4}]}
    torch.save(ck, path)
    return ck


def test_dfpn_load_from_checkpoint_lightning_format(tmp_path):
    import master_thesis_amd as mt
    vgg = mt.VGGFeatures()
    src = mt.DFPN(vgg, **KW)
    src.load_state_dict(S.add_buffers(S.seeded_state_dict(S.dfpn_shapes())))
    path = tmp_path / 'dfpn.ckpt'
    _lightning_file(path, src)
    # __main__.py:52-54: DFPN.load_from_checkpoint(path, model_vgg=model_vgg, **vars(args))
    got = mt.DFPN.load_from_checkpoint(str(path), model_vgg=mt.VGGFeatures(), **KW, unrelated_cli_flag=1)
    assert isinstance(got, mt.DFPN) and got.kwargs['lr'] == 1e-4
    a, b = src.state_dict(), got.state_dict()
    assert list(a) == list(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # a bare state_dict file (no 'state_dict' key) is accepted too
    torch.save(src.state_dict(), tmp_path / 'bare.pt')
    got2 = mt.DFPN.load_from_checkpoint(str(tmp_path / 'bare.pt'), model_vgg=mt.VGGFeatures(), **KW)
    assert torch.equal(got2.flow_256.nn[40].weight, src.flow_256.nn[40].weight)


def test_chn_load_from_checkpoint_lightning_format(tmp_path):
    import master_thesis_amd as mt
    vgg = mt.VGGFeatures()
    src = mt.CHN(vgg, None, mt.DFPN(vgg, **KW), **KW)
    src.load_state_dict(S.add_buffers(S.seeded_state_dict(S.chn_shapes()), ('', 'model_aligner.')))
    path = tmp_path / 'chn.ckpt'
    _lightning_file(path, src)
    vgg2 = mt.VGGFeatures()
    # __main__.py:38-42: CHN.load_from_checkpoint(path, model_vgg=..., model_lpips=..., model_aligner=..., **vars(args))
    got = mt.CHN.load_from_checkpoint(str(path), model_vgg=vgg2, model_lpips=None, model_aligner=mt.DFPN(vgg2, **KW), **KW)
    a, b = src.state_dict(), got.state_dict()
    assert list(a) == list(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # strict loading refuses a file written for another model
    with pytest.raises(RuntimeError):
        mt.DFPN.load_from_checkpoint(str(path), model_vgg=mt.VGGFeatures(), **KW)


def test_lightning_optimizer_state_with_frozen_parameters_in_between(tmp_path):
    """The reference gives Adam self.parameters() (model_dfpn.py:302).  DFPN.corr = CorrelationVGG(model_vgg) is the
    first submodule (model_dfpn.py:32,487-488): after its 12 SeparableConv4d tensors come the frozen VGG tensors (26
    feature tensors + torchvision's 6 classifier tensors), which never receive a gradient and so have no Adam state.
    The state of the trainable parameters is therefore keyed 0..11, 44.., not 0..n-1."""
    import master_thesis_amd as mt
    model = mt.DFPN(mt.VGGFeatures(), **KW)
    model.load_state_dict(S.add_buffers(S.seeded_state_dict(S.dfpn_shapes())))
    trainable = model.trainable_parameters()
    every, has_grad = [], []                                 # the list the reference's Adam sees, in its order
    for n, p in model.named_parameters():                    # (named_parameters() de-duplicates the shared VGG)
        every.append(p)
        has_grad.append('model_vgg' not in n)
        if n == 'corr.model_vgg.features.28.bias':           # torchvision's VGG carries its classifier along
            for shape in ((8, 16), (8,), (8, 8), (8,), (4, 8), (4,)):
                every.append(torch.zeros(shape))
                has_grad.append(False)
    assert has_grad[:12] == [True] * 12 and has_grad[12:44] == [False] * 32 and sum(has_grad) == len(trainable)
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in every], lr=2.5e-5)
    g = torch.Generator().manual_seed(1)
    for _ in range(2):
        for p, hg in zip(ref.param_groups[0]['params'], has_grad):
            p.grad = torch.randn(p.shape, generator=g) if hg else None    # VGG: no gradient, no state
        ref.step()
    state = ref.state_dict()
    keys = sorted(state['state'])
    assert keys == [i for i, hg in enumerate(has_grad) if hg] and keys[12] == 44
    path = tmp_path / 'dfpn.ckpt'
    _lightning_file(path, model, state)
    ck = torch.load(path)
    resumed = mt.DFPN.load_from_checkpoint(str(path), model_vgg=mt.VGGFeatures(), **KW)
    opt = resumed.configure_optimizers()['optimizer']
    opt.load_state_dict(ck['optimizer_states'][0])
    assert opt.step_count == 2 and opt.param_groups[0]['lr'] == 2.5e-5
    for k, (off, n) in enumerate(opt._slices):
        ent = state['state'][keys[k]]
        assert torch.equal(opt.exp_avg[off:off + n], ent['exp_avg'].reshape(-1)), k
        assert torch.equal(opt.exp_avg_sq[off:off + n], ent['exp_avg_sq'].reshape(-1)), k
    # a state whose k-th entry does not fit the k-th trainable parameter is refused before anything is copied
    bad = {'state': dict(state['state']), 'param_groups': state['param_groups']}
    last = keys[-1]
    bad['state'][last] = dict(bad['state'][last], exp_avg=torch.zeros(7), exp_avg_sq=torch.zeros(7))
    before = opt.exp_avg.clone()
    with pytest.raises(ValueError, match='does not fit'):
        opt.load_state_dict(bad)
    assert torch.equal(opt.exp_avg, before)
    # ... and so is an update rule the fused kernel does not implement
    wd = {'state': state['state'], 'param_groups': [dict(state['param_groups'][0], weight_decay=0.01)]}
    with pytest.raises(ValueError, match='plain Adam'):
        opt.load_state_dict(wd)


def test_cpn_init_model_with_state(tmp_path):
    """model_cpn.py:93-118: the published CPN weights are a DataParallel state dict ('module.' prefix) that also
    holds CPN's own inpainting network (Encoder / CM_Module / Decoder), which this path does not use."""
    import master_thesis_amd as mt
    sd = S.cpn_state_dict()
    file_sd = {'module.' + k: v for k, v in sd.items()}
    file_sd['module.Encoder.conv12.conv.0.weight'] = torch.zeros(64, 4, 5, 5)
    file_sd['module.CM_Module.dummy'] = torch.zeros(3)
    file_sd['module.Decoder.conv4.conv.0.bias'] = torch.zeros(257)
    path = tmp_path / 'cpn.pth'
    torch.save(file_sd, path)
    model = mt.CPN.init_model_with_state(str(path), 'cpu')
    assert isinstance(model, mt.CPN)
    own = model.state_dict()
    assert set(own) == set(sd)
    for k in sd:
        assert torch.equal(own[k], sd[k]), k
    assert all(not p.requires_grad for p in model.parameters())
