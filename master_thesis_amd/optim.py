"""Adam over ONE flat fp32 buffer (torch.optim.Adam semantics with default
betas/eps, model_dfpn.py:302 / model_chn.py:385) as a single fused HIP kernel,
plus the flat gradient buffer the data-parallel all-reduce works on."""
import math

import torch

from . import _lib as L
from . import engine as E


def reference_parameter_list(model):
    """The list ``model.parameters()`` yields in the REFERENCE, as far as its length and order go: this package's
    parameters in registration order (the module trees match: tests/test_abi_and_host.py), plus one placeholder per
    tensor the reference carries and this package does not -- the six ``classifier`` tensors of torchvision's VGG
    behind every VGGFeatures' ``features`` (model_vgg.py:11,20: the reference derives from ``torchvision.models.VGG``).
    ``torch.optim.Adam`` numbers its state by position in this list and refuses a state dict whose group has another
    length, so FlatAdam.state_dict() needs it to write checkpoints the reference can resume from."""
    from .model_vgg import VGGFeatures
    extra = {}
    for mod in model.modules():
        if isinstance(mod, VGGFeatures):
            ps = list(mod.parameters())
            if ps:
                extra[id(ps[-1])] = VGGFeatures.REFERENCE_CLASSIFIER_TENSORS
    out = []
    for p in model.parameters():                   # (de-duplicated: a shared VGG appears once, where it is met first)
        out.append(p)
        out.extend(object() for _ in range(extra.get(id(p), 0)))
    return out


class FlatAdam(torch.optim.Optimizer):
    """``all_params``: the parameter list the REFERENCE hands its ``torch.optim.Adam`` -- ``self.parameters()``
    (model_dfpn.py:302, model_chn.py:385), frozen tensors included.  ``state_dict()`` numbers the optimizer state by
    position in that list, as torch does, so that a checkpoint written here resumes in the reference."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, all_params=None):
        params = [p for p in params if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._plist = params
        if all_params is None:
            self._ref_index, self._ref_count = list(range(len(params))), len(params)
        else:
            pos = {id(p): i for i, p in enumerate(all_params)}
            missing = [i for i, p in enumerate(params) if id(p) not in pos]
            if missing:
                raise ValueError(f'FlatAdam: trainable parameters {missing[:8]} are not in all_params')
            self._ref_index, self._ref_count = [pos[id(p)] for p in params], len(pos)
            if self._ref_index != sorted(self._ref_index):
                raise ValueError('FlatAdam: the trainable parameters must keep the order of all_params')
        dev = params[0].device
        n = sum(p.numel() for p in params)
        self.flat_p = torch.empty((n,), device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros((n,), device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.step_count = 0
        self.grad_scale = 1.0          # 1/world_size after a SUM all-reduce
        # device-resident {step, lr}: lets a captured hipGraph replay the update
        self.dev_state = torch.tensor([0.0, lr], device=dev, dtype=torch.float32)
        off = 0
        self._slices = []
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_p[off:off + k].view(p.shape)
                p.grad = self._grad_view(off, k, p.shape)
                self._slices.append((off, k))
                off += k
        E.bump_weight_epoch()

    def _grad_view(self, off, k, shape):
        """A parameter's gradient = a window of the flat buffer.  The tag tells the conv engine's
        backward that it may ADD a weight / bias gradient straight into it (and hand autograd None)
        instead of returning a temporary for AccumulateGrad to add."""
        g = self.flat_g[off:off + k].view(shape)
        g._mt_flat = True
        return g

    def zero_grad(self, set_to_none=False):
        """Gradients stay views of the flat buffer (autograd accumulates in
        place), so they are zeroed, never dropped."""
        self.flat_g.zero_()
        for p, (off, k) in zip(self._plist, self._slices):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self._grad_view(off, k, p.shape)

    def _gather_foreign_grads(self):
        for p, (off, k) in zip(self._plist, self._slices):
            want = self.flat_g.data_ptr() + 4 * off
            if p.grad is None:
                self.flat_g[off:off + k].zero_()
                p.grad = self._grad_view(off, k, p.shape)
            elif p.grad.data_ptr() != want:
                self.flat_g[off:off + k].copy_(p.grad.reshape(-1))
                p.grad = self._grad_view(off, k, p.shape)

    def check_aliasing(self):
        """Every parameter must still BE its window of the flat buffer (``model.to(dtype)``,
        ``load_state_dict(assign=True)`` or ``p.data = ...`` silently break that: Adam would then
        update a buffer nobody reads).  A parameter that moved is re-adopted: its values are copied
        into the window, ``p.data`` points back at it and the packed weight images are refreshed."""
        moved = 0
        for p, (off, k) in zip(self._plist, self._slices):
            if p.data_ptr() != self.flat_p.data_ptr() + 4 * off:
                assert p.numel() == k, 'a parameter changed its size under FlatAdam'
                self.flat_p[off:off + k].copy_(p.detach().reshape(-1).to(self.flat_p))
                p.data = self.flat_p[off:off + k].view(p.shape)
                moved += 1
        if moved:
            E.bump_weight_epoch()
        return moved

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        capturing = self.flat_p.is_cuda and torch.cuda.is_current_stream_capturing()
        if not capturing:
            self.check_aliasing()
        self._gather_foreign_grads()
        g = self.param_groups[0]
        b1, b2 = g['betas']
        self.step_count += 1
        if capturing:
            # graph capture: step and lr are read from the device at replay time
            L.call('mt_adam_step_dev', L.ptr(self.flat_p), L.ptr(self.flat_g), L.ptr(self.exp_avg),
                   L.ptr(self.exp_avg_sq), self.flat_p.numel(), b1, b2, g['eps'], self.grad_scale,
                   L.ptr(self.dev_state), L.stream_ptr())
        else:
            self.dev_state[0] = float(self.step_count)
            self.sync_device_lr()
            bc1 = 1.0 - b1 ** self.step_count
            bc2 = 1.0 - b2 ** self.step_count
            L.call('mt_adam_step', L.ptr(self.flat_p), L.ptr(self.flat_g), L.ptr(self.exp_avg),
                   L.ptr(self.exp_avg_sq), self.flat_p.numel(), b1, b2, g['lr'] / bc1, 1.0 / math.sqrt(bc2), g['eps'],
                   self.grad_scale, L.stream_ptr())
        # only this buffer was rewritten: packs of weights outside it (frozen VGG / aligner) stay as they are
        E.bump_weight_epoch((self.flat_p.data_ptr(), self.flat_p.data_ptr() + 4 * self.flat_p.numel()))
        return loss

    def sync_device_lr(self):
        """Device copy of the learning rate (read by the graph-captured update).  Called by
        step() and by GraphedTrainStep before every replay, so an LR scheduler or a resumed
        checkpoint needs no extra call."""
        lr = float(self.param_groups[0]['lr'])
        if lr != getattr(self, '_dev_lr', None):
            self.dev_state[1] = lr
            self._dev_lr = lr

    def sync_device_state(self):
        """Device copies of {step, lr} from the host-side state (after load_state_dict)."""
        self.dev_state[0] = float(self.step_count)
        self._dev_lr = None
        self.sync_device_lr()

    def flat_state_dict(self):
        """The optimizer state as it lives here: two flat moment buffers (rounds 1-3 wrote this layout; still loadable)."""
        return {'step': self.step_count, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq,
                'param_groups': [{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups]}

    def state_dict(self):
        """The ``torch.optim.Adam`` layout (what the reference and Lightning write and read: model_dfpn.py:295-308,
        __main__.py:33-35): ``state`` = {index in the reference's parameter list: {step, exp_avg, exp_avg_sq}} for the
        parameters that have taken a step -- frozen tensors have an index and no entry, exactly as torch leaves them --
        and one param group whose ``params`` runs over ALL of that list.  The moments are views of the flat buffers (a
        checkpoint writer serialises them per parameter); ``step`` is a plain int, which torch 1.10 (the reference's pin)
        uses as it is and current torch converts on load.  load_state_dict() takes it back."""
        g = self.param_groups[0]
        group = {'lr': g['lr'], 'betas': tuple(g['betas']), 'eps': g['eps'], 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None}
        group.update({k: v for k, v in g.items() if k not in group and k != 'params'})      # (initial_lr of a scheduler, ...)
        group['params'] = list(range(self._ref_count))
        state = {}
        if self.step_count > 0:
            for idx, p, (off, k) in zip(self._ref_index, self._plist, self._slices):
                state[idx] = {'step': int(self.step_count), 'exp_avg': self.exp_avg[off:off + k].view(p.shape),
                              'exp_avg_sq': self.exp_avg_sq[off:off + k].view(p.shape)}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        """Accepts FlatAdam's own layout and the ``torch.optim.Adam`` layout of reference /
        Lightning checkpoints (``state`` = {index: {step, exp_avg, exp_avg_sq}} over the same
        parameter order, ``param_groups``).  The device-resident {step, lr} follow."""
        for s in sd['param_groups']:
            # the fused kernel is plain Adam (model_dfpn.py:302, model_chn.py:385): refuse a state written for another rule
            if s.get('weight_decay', 0) or s.get('amsgrad', False) or s.get('maximize', False):
                raise ValueError('checkpoint optimizer uses weight_decay / amsgrad / maximize '
                                 f'({ {k: s.get(k) for k in ("weight_decay", "amsgrad", "maximize")} }): '
                                 'FlatAdam implements plain Adam only')
        if 'state' in sd and 'exp_avg' not in sd:
            # torch.optim.Adam numbers the state by position in ITS parameter list.  The reference hands Adam
            # self.parameters() (model_dfpn.py:302), where the frozen VGG tensors sit between the SeparableConv4d and the
            # mixer parameters (model_dfpn.py:32,487-488) and never get state: the k-th STATEFUL parameter of the
            # checkpoint is the k-th trainable parameter here, whatever its index.
            st = sd['state']
            keys = sorted(st.keys(), key=int)
            if len(keys) == 0:                      # (a torch optimizer that has not taken a step yet holds no state)
                self.exp_avg.zero_()
                self.exp_avg_sq.zero_()
            if len(keys) not in (0, len(self._plist)):
                raise ValueError(f'optimizer state holds {len(keys)} parameters, this model trains {len(self._plist)}: '
                                 'the checkpoint was written for a different set of trainable parameters')
            ents = [st[k] for k in keys]
            for n_, (ent, (off, k)) in enumerate(zip(ents, self._slices)):      # validate everything before copying anything
                for name in ('exp_avg', 'exp_avg_sq'):
                    if ent[name].numel() != k:
                        raise ValueError(f'optimizer state entry {keys[n_]} ({name}: {ent[name].numel()} elements) does not '
                                         f'fit trainable parameter {n_} ({k} elements, shape {tuple(self._plist[n_].shape)})')
            steps = {int(ent['step']) for ent in ents}
            if len(steps) > 1:
                raise ValueError(f'per-parameter step counts differ ({sorted(steps)}): one fused update needs one')
            for ent, (off, k) in zip(ents, self._slices):
                self.exp_avg[off:off + k].copy_(ent['exp_avg'].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(ent['exp_avg_sq'].reshape(-1))
            self.step_count = steps.pop() if steps else 0
        else:
            if sd['exp_avg'].numel() != self.exp_avg.numel() or sd['exp_avg_sq'].numel() != self.exp_avg_sq.numel():
                raise ValueError(f'flat optimizer state holds {sd["exp_avg"].numel()} elements, this model trains '
                                 f'{self.exp_avg.numel()}')
            self.step_count = int(sd['step'])
            self.exp_avg.copy_(sd['exp_avg'])
            self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        for g, s in zip(self.param_groups, sd['param_groups']):
            g.update({k: v for k, v in s.items() if k in ('lr', 'betas', 'eps', 'initial_lr')})
        self.sync_device_state()
        E.bump_weight_epoch()
