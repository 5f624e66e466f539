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


def _reference_adam_list(model):
    """The tensors the reference's ``torch.optim.Adam(self.parameters())`` sees for a DFPN (model_dfpn.py:302), in its
    order, with small stand-ins for torchvision's six VGG classifier tensors; and which of them receive gradients."""
    every, has_grad = [], []
    for n, p in model.named_parameters():                    # (named_parameters() de-duplicates the shared VGG)
        every.append(p)
        has_grad.append('model_vgg' not in n)
        if n == 'corr.model_vgg.features.28.bias':           # torchvision's VGG carries its classifier along
            for shape in ((8, 16), (8,), (8, 8), (8,), (4, 8), (4,)):
                every.append(torch.zeros(shape))
                has_grad.append(False)
    return every, has_grad


def test_optimizer_state_written_here_resumes_in_torch_adam(tmp_path):
    """The other direction (missing #4): ``FlatAdam.state_dict()`` IS the ``torch.optim.Adam`` layout
    (state keyed by the reference's parameter indices, frozen tensors have an index and no state, one group over the
    whole list), so a Lightning checkpoint written by this package resumes in the reference: a real torch Adam over
    the reference's parameter list loads it and its next step is bit-identical to that of the torch Adam the state
    came from.  Also: a state saved before the first step, and the flat layout of rounds 1-3, still load."""
    import master_thesis_amd as mt
    model = mt.DFPN(mt.VGGFeatures(), **KW)
    model.load_state_dict(S.add_buffers(S.seeded_state_dict(S.dfpn_shapes())))
    every, has_grad = _reference_adam_list(model)
    from master_thesis_amd.optim import reference_parameter_list
    assert len(reference_parameter_list(model)) == len(every)
    origin = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in every], lr=2.5e-5)
    g = torch.Generator().manual_seed(2)

    def give_grads(opt_, seed):
        gg = torch.Generator().manual_seed(seed)
        for p, hg in zip(opt_.param_groups[0]['params'], has_grad):
            p.grad = torch.randn(p.shape, generator=gg) if hg else None
    for s_ in (10, 11, 12):
        give_grads(origin, s_)
        origin.step()
    opt = model.configure_optimizers()['optimizer']
    fresh_state = opt.state_dict()                                       # before any step: no state, as torch
    assert fresh_state['state'] == {} and fresh_state['param_groups'][0]['params'] == list(range(len(every)))
    opt.load_state_dict(origin.state_dict())
    path = tmp_path / 'written_here.ckpt'
    _lightning_file(path, model, opt.state_dict())                       # what pl.Trainer would write from this package
    written = torch.load(path)['optimizer_states'][0]
    want = origin.state_dict()
    assert sorted(written['state']) == sorted(want['state']) == [i for i, hg in enumerate(has_grad) if hg]
    assert written['param_groups'][0]['params'] == want['param_groups'][0]['params']
    for k in ('lr', 'betas', 'eps', 'weight_decay', 'amsgrad'):
        assert written['param_groups'][0][k] == want['param_groups'][0][k], k
    for i, ent in want['state'].items():
        assert int(written['state'][i]['step']) == int(ent['step']) == 3
        assert written['state'][i]['exp_avg'].shape == ent['exp_avg'].shape
        assert torch.equal(written['state'][i]['exp_avg'], ent['exp_avg'])
        assert torch.equal(written['state'][i]['exp_avg_sq'], ent['exp_avg_sq'])
    # the reference side: a torch.optim.Adam over ITS parameter list resumes from the file and steps identically
    resumed = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in origin.param_groups[0]['params']], lr=1.0)
    resumed.load_state_dict(written)
    assert resumed.param_groups[0]['lr'] == 2.5e-5
    give_grads(origin, 13)
    give_grads(resumed, 13)
    origin.step()
    resumed.step()
    for a, b in zip(origin.param_groups[0]['params'], resumed.param_groups[0]['params']):
        assert torch.equal(a, b)
    # a scheduler's bookkeeping travels in the group; the flat layout (rounds 1-3) still loads
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=50, gamma=0.5)
    assert 'initial_lr' in opt.state_dict()['param_groups'][0] and sched is not None
    opt2 = mt.DFPN(mt.VGGFeatures(), **KW).configure_optimizers()['optimizer']
    opt2.load_state_dict(opt.flat_state_dict())
    assert opt2.step_count == 3 and torch.equal(opt2.exp_avg, opt.exp_avg)
    opt2.load_state_dict(fresh_state)
    assert opt2.step_count == 0 and float(opt2.exp_avg.abs().max()) == 0.0


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
