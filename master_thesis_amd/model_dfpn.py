"""Dense Flow Prediction Network (DFPN) on the mt355 HIP kernels.

Drop-in for ``master_thesis/model_dfpn.py``: same class names, constructor
arguments, method signatures, return layouts, loss names/order and state-dict
keys (``corr.conv.conv_{1,2}.N``, ``corr_mixer.mixer.N``, ``flow_{64,256}.nn.N``,
``mean``, ``std``, ``model_vgg.*``, ``corr.model_vgg.*``).  The ``nn.Sequential``
containers only hold the parameters; the arithmetic runs in libmt355.
"""
import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E
from . import ops
from .lightning_shim import LightningModule
from .optim import FlatAdam, reference_parameter_list
from .utils import FlowsUtils, LossesUtils, TransformsUtils, take

RELU, NONE = L.ACT_RELU, L.ACT_NONE


def _specs_from(seq):
    """LayerSpecs + flat parameter list of an nn.Sequential of Conv2d /
    ConvTranspose2d / ReLU (the activation after a conv is fused into it)."""
    mods = list(seq)
    specs, params = [], []
    for i, m in enumerate(mods):
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            act = RELU if i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU) else NONE
            kind = 't' if isinstance(m, nn.ConvTranspose2d) else 'c'
            specs.append(E.LayerSpec(kind, m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0], act))
            params += [m.weight, m.bias]
    return specs, params


class DFPN(LightningModule):
    """model_dfpn.py:14-473."""
    LOSSES_NAMES = ['corr_loss', 'flow_16', 'flow_64', 'flow_256', 'alignment_recons_64', 'alignment_recons_256']

    def __init__(self, model_vgg, **kwargs):
        super().__init__()
        self.corr = CorrelationVGG(model_vgg)
        self.corr_mixer = AlignmentCorrelationMixer()
        self.flow_64 = FlowEstimator()
        self.flow_256 = FlowEstimator()
        self.register_buffer('mean', torch.as_tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1, 1))
        self.register_buffer('std', torch.as_tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1, 1))
        self.model_vgg = model_vgg
        self.kwargs = kwargs

    def _affine(self):
        mean, std = self.mean.reshape(3), self.std.reshape(3)
        return (1.0 / std).contiguous(), (-mean / std).contiguous()

    def forward(self, x_target, m_target, x_refs, m_refs):
        """model_dfpn.py:46-101.  The (x-mean)/std normalisation (:71-72) is
        fused into the kernels that build the network inputs; the resizes act
        on the raw frames (bilinear resize and the per-channel affine commute)."""
        b, c, ref_n, h, w = x_refs.size()
        scale, shift = self._affine()
        sq = TransformsUtils.resize_set_bis(x_target, m_target, x_refs, m_refs, (256, 256))
        s64 = TransformsUtils.resize_set_bis(x_target, m_target, x_refs, m_refs, (64, 64))

        corr = self.corr._forward_raw(*sq, scale, shift)
        flow_16 = self.corr_mixer(corr)

        flow_64_pre = FlowsUtils.resize_flow(flow_16, (64, 64), mode='bilinear')
        flow_64 = self.flow_64._forward_raw(*s64, flow_64_pre, scale, shift)
        flow_256_pre = FlowsUtils.resize_flow(flow_64, (256, 256), mode='bilinear')
        flow_256 = self.flow_256._forward_raw(*sq, flow_256_pre, scale, shift)
        if (h, w) != (256, 256):
            flow_256 = FlowsUtils.resize_flow(flow_256, (h, w), mode='bilinear')
        return corr, flow_16, flow_64, flow_256

    def align(self, x_target, m_target, x_refs, m_refs):
        """model_dfpn.py:103-133."""
        with torch.no_grad():
            *_, flow_256 = self(x_target, m_target, x_refs, m_refs)
        x_ref_aligned, v_ref_aligned = FlowsUtils.align_set(x_refs, (1 - m_refs), flow_256)
        v_map = (v_ref_aligned - (1 - m_target).unsqueeze(2)).clamp(0, 1)
        return x_ref_aligned, v_ref_aligned, v_map

    def _step(self, batch, flow_key):
        (x, m), y, info = batch
        flows_use, flow_gt = info[2], info[flow_key]
        t, r_list = DFPN.get_indexes(x.size(2))
        out = DFPN._train_val_wrapper(self, x, m, y, flow_gt, flows_use, t, r_list)
        return self.compute_loss(*out, t, r_list)

    def training_step(self, batch, batch_idx):
        """model_dfpn.py:135-158."""
        loss, loss_items = self._step(batch, 4)
        self._log_losses(loss, loss_items, 'training')
        return loss

    def validation_step(self, batch, batch_idx):
        """model_dfpn.py:160-183."""
        loss, loss_items = self._step(batch, 4)
        self._log_losses(loss, loss_items, 'validation')
        return loss

    def test_step(self, batch, batch_idx):
        """model_dfpn.py:185-208 (reads info[5] as the flow, as the reference
        does; frame logging needs TensorBoard and is out of scope)."""
        loss, _ = self._step(batch, 5)
        return loss

    def compute_loss(self, corr, xs, vs, ys, xs_aligned, flows, flows_gt, flows_use, t, r_list):
        """model_dfpn.py:210-293."""
        b, c, f, h, w = ys[2].size()
        with torch.no_grad():
            y_in = ys[2]
            if (h, w) != (256, 256):
                y_in = ops.resize_planar(y_in, (256, 256), 'bilinear')
            vgg = self.model_vgg
            scale, shift = vgg._affine(y_in.device)
            y4 = torch.empty((b * f, 256, 256, E.cpad(3)), device=y_in.device, dtype=torch.float32)
            E.pack_images(y4, b, f, [(y_in, 0, scale, shift)])
            feats = vgg.forward4(y4, n_taps=4)[3]                       # (b*f,16,16,512)
            feats = feats.reshape(b, f, 16, 16, feats.shape[3])
            ft = feats[:, t].contiguous()
            fr = take(feats, 1, r_list).reshape(b * len(r_list), 16, 16, feats.shape[4])
            corr_y = ops.correlation(ft, fr, None, None, b, len(r_list)).reshape(b, len(r_list), 16, 16, 16, 16)
        corr_loss = LossesUtils.l1(corr, corr_y)

        flow_loss_16 = LossesUtils.masked_l1(flows[0], flows_gt[0], torch.ones_like(flows[0]), flows_use)
        flow_loss_64 = LossesUtils.masked_l1(flows[1], flows_gt[1], torch.ones_like(flows[1]), flows_use)
        flow_loss_256 = LossesUtils.masked_l1(flows[2], flows_gt[2], torch.ones_like(flows[2]), flows_use)

        mask_out_64 = ((flows[1] < -1).float() + (flows[1] > 1).float()).sum(4).clamp(0, 1).unsqueeze(1)
        mask_out_256 = ((flows[2] < -1).float() + (flows[2] > 1).float()).sum(4).clamp(0, 1).unsqueeze(1)

        r = len(r_list)
        alignment_recons_64 = LossesUtils.masked_l1(
            xs[1][:, :, t].unsqueeze(2).repeat(1, 1, r, 1, 1), xs_aligned[1],
            vs[1][:, :, t].unsqueeze(2).repeat(1, 1, r, 1, 1) * (1 - mask_out_64), reduction='sum')
        alignment_recons_256 = LossesUtils.masked_l1(
            xs[2][:, :, t].unsqueeze(2).repeat(1, 1, r, 1, 1), xs_aligned[2],
            vs[2][:, :, t].unsqueeze(2).repeat(1, 1, r, 1, 1) * (1 - mask_out_256), reduction='sum')

        total_loss = corr_loss + flow_loss_16 + flow_loss_64 + flow_loss_256
        total_loss = total_loss + alignment_recons_64 + alignment_recons_256
        return total_loss, [corr_loss, flow_loss_16, flow_loss_64, flow_loss_256, alignment_recons_64,
                            alignment_recons_256]

    def trainable_parameters(self):
        """The parameters that receive gradients (everything but the VGG)."""
        return [p for n, p in self.named_parameters() if 'model_vgg' not in n]

    def configure_optimizers(self):
        """model_dfpn.py:295-308: Adam + StepLR; the Adam update is one fused
        kernel over a flat parameter buffer."""
        optimizer = FlatAdam(self.trainable_parameters(), lr=self.kwargs['lr'], all_params=reference_parameter_list(self))
        lr_scheduler = torch.optim.lr_scheduler.StepLR(
            optimizer, step_size=self.kwargs['lr_scheduler_step_size'], gamma=self.kwargs['lr_scheduler_gamma'])
        return {'optimizer': optimizer, 'lr_scheduler': lr_scheduler}

    def _train_val_wrapper(self, x, m, y, flow_gt, flows_use, t, r_list):
        """model_dfpn.py:310-394.  The reference's GT-flow warps and v_map_gt
        (:358-375) never reach the outputs and are skipped; the 16x16 warp
        (:377-379) and the 16/64 resizes of y feed nothing in compute_loss and
        are returned as None."""
        x_refs, m_refs = take(x, 2, r_list), take(m, 2, r_list)
        corr, flow_16, flow_64, flow_256 = self(x[:, :, t], m[:, :, t], x_refs, m_refs)
        v = 1 - m
        x_64 = ops.resize_planar(x, (64, 64), 'bilinear')
        v_64 = ops.resize_planar(v, (64, 64), 'nearest')
        flow_gt_r = take(flow_gt, 1, r_list)
        flow_16_gt = FlowsUtils.resize_flow(flow_gt_r, (16, 16))
        flow_64_gt = FlowsUtils.resize_flow(flow_gt_r, (64, 64))
        x_64_aligned, _ = FlowsUtils.align_set(take(x_64, 2, r_list), take(v_64, 2, r_list), flow_64)
        x_256_aligned, _ = FlowsUtils.align_set(x_refs, 1 - m_refs, flow_256)
        xs, vs, ys = (None, x_64, x), (None, v_64, v), (None, None, y)
        xs_aligned = (None, x_64_aligned, x_256_aligned)
        flows = (flow_16, flow_64, flow_256)
        flows_gt = (flow_16_gt, flow_64_gt, flow_gt_r)
        return corr, xs, vs, ys, xs_aligned, flows, flows_gt, flows_use

    def _log_losses(self, loss, loss_items, split):
        """model_dfpn.py:396-409."""
        self.log('loss_{}'.format(split), loss)
        for i, loss_item_id in enumerate(DFPN.LOSSES_NAMES):
            self.log('loss_{}_{}'.format(loss_item_id, split), loss_items[i])

    @staticmethod
    def get_indexes(size):
        """model_dfpn.py:458-473."""
        t, r_list = size // 2, list(range(size))
        r_list.pop(t)
        return t, r_list


class CorrelationVGG(nn.Module):
    """model_dfpn.py:476-583."""

    def __init__(self, model_vgg, use_softmax=False):
        super().__init__()
        self.conv = SeparableConv4d()
        self.model_vgg = model_vgg
        self.use_softmax = use_softmax

    def forward(self, x_target, m_target, x_refs, m_refs):
        """model_dfpn.py:491-532: inputs are already normalised."""
        return self._forward_raw(x_target, m_target, x_refs, m_refs, None, None)

    def _forward_raw(self, x_target, m_target, x_refs, m_refs, scale, shift):
        b, c, ref_n, h, w = x_refs.size()
        with torch.no_grad():
            # one VGG pass over the target and all reference frames (N = B + B*R)
            x4 = torch.empty((b + b * ref_n, h, w, E.cpad(3)), device=x_refs.device, dtype=torch.float32)
            E.pack_images(x4[:b], b, 1, [(x_target, 0, scale, shift)])
            E.pack_images(x4[b:], b, ref_n, [(x_refs, 0, scale, shift)])
            feats = self.model_vgg.forward4(x4, n_taps=4)[3]
            fh, fw = feats.shape[1:3]
            v_target = ops.resize_planar(1 - m_target, (fh, fw), 'nearest').reshape(b, fh * fw)
            v_ref = ops.resize_planar(1 - m_refs, (fh, fw), 'nearest')          # (B,1,R,fh,fw)
            v_ref = v_ref.reshape(b * ref_n, fh * fw)
            corr = ops.correlation(feats[:b], feats[b:], v_target, v_ref, b, ref_n)
        corr = self.conv._forward_pairs(corr, b, ref_n)
        return CorrelationVGG.softmax_3d(corr) if self.use_softmax else corr

    @staticmethod
    def correlation_masked_4d(x_target_feats, v_target, x_ref_feats, v_ref):
        """model_dfpn.py:534-565 on (B,C,H,W) / (B,C,F,H,W) tensors."""
        b, c, ref_n, h, w = x_ref_feats.size()
        ft = E.to_nhwc4(x_target_feats)
        fr = torch.empty((b * ref_n, h, w, E.cpad(c)), device=ft.device, dtype=torch.float32)
        if E.cpad(c) != c:
            fr.zero_()
        E.pack_images(fr, b, ref_n, [(x_ref_feats, 0, None, None)])
        vt = v_target.reshape(b, h * w) if v_target is not None else None
        vr = v_ref.transpose(1, 2).reshape(b * ref_n, h * w) if v_ref is not None else None
        return ops.correlation(ft, fr, vt, vr, b, ref_n).reshape(b, ref_n, h, w, h, w)

    @staticmethod
    def softmax_3d(x):
        """model_dfpn.py:567-583 (off by default): per target position a softmax over
        (reference frame, reference position) jointly -- one wave per row in HIP, the
        maximum and the sum by wave shuffles, no permuted copies.  No gradient (the
        volume comes from no-grad features)."""
        return ops.softmax_3d(x)


class SeparableConv4d(nn.Module):
    """model_dfpn.py:586-626."""

    def __init__(self):
        super().__init__()
        self.conv_1 = nn.Sequential(
            nn.Conv2d(1, 128, (3, 3), padding=1), nn.ReLU(),
            nn.Conv2d(128, 256, (3, 3), padding=1), nn.ReLU(),
            nn.Conv2d(256, 256, (3, 3), padding=1))
        self.conv_2 = nn.Sequential(
            nn.Conv2d(256, 256, (3, 3), padding=1), nn.ReLU(),
            nn.Conv2d(256, 128, (3, 3), padding=1), nn.ReLU(),
            nn.Conv2d(128, 1, (3, 3), padding=1))

    def forward(self, corr):
        """corr (B,F,16,16,16,16) -> filled volume of the same shape."""
        b, t, h, w = corr.shape[:4]
        return self._forward_pairs(corr.reshape(b * t, h * w, h * w), b, t)

    def _forward_pairs(self, corr, b, r):
        """corr (B*R, P_t, P_r).  conv_1 runs over the reference dims per target
        position; the (:621-623) permute is the swap_view addressing of conv_2's
        first layer; conv_2 runs over the target dims per reference position."""
        g, p = b * r, corr.shape[1]
        side = int(round(p ** 0.5))
        x4 = E.ToNHWC4Fn.apply(corr.reshape(g * p, 1, side, side))
        s1, p1 = _specs_from(self.conv_1)
        s2, p2 = _specs_from(self.conv_2)
        s2[0].swap_groups = g
        out = E.conv_stack(x4, s1 + s2, p1 + p2)              # [(g, p_r)][p_t][Cp]
        vol = out.reshape(g, p, p, out.shape[3])[..., 0]       # [g][p_r][p_t]
        # (:625-626) [b,r,h_r,w_r,h_t,w_t] -> [b,r,h_t,w_t,h_r,w_r]
        return vol.reshape(b, r, side, side, side, side).permute(0, 1, 4, 5, 2, 3)


class AlignmentCorrelationMixer(nn.Module):
    """model_dfpn.py:629-674."""

    def __init__(self, corr_size=16):
        super().__init__()
        cs = corr_size
        self.mixer = nn.Sequential(
            nn.Conv2d(cs ** 2, cs ** 2, (5, 5), padding=2), nn.ReLU(),
            nn.Conv2d(cs ** 2, cs ** 2, (3, 3), padding=1), nn.ReLU(),
            nn.Conv2d(cs ** 2, cs, (3, 3), padding=1), nn.ReLU(),
            nn.Conv2d(cs, cs, (5, 5), padding=2), nn.ReLU(),
            nn.Conv2d(cs, cs, (3, 3), padding=1), nn.ReLU(),
            nn.Conv2d(cs, cs // 2, (3, 3), padding=1), nn.ReLU(),
            nn.Conv2d(cs // 2, cs // 2, (5, 5), padding=2), nn.ReLU(),
            nn.Conv2d(cs // 2, cs // 2, (3, 3), padding=1), nn.ReLU(),
            nn.Conv2d(cs // 2, cs // 4, (3, 3), padding=1), nn.ReLU(),
            nn.Conv2d(cs // 4, cs // 4, (5, 5), padding=2), nn.ReLU(),
            nn.Conv2d(cs // 4, cs // 4, (3, 3), padding=1), nn.ReLU(),
            nn.Conv2d(cs // 4, cs // 8, (3, 3), padding=1),
            nn.Conv2d(cs // 8, cs // 8, (5, 5), padding=2),
            nn.Conv2d(cs // 8, cs // 8, (3, 3), padding=1))

    def forward(self, corr):
        """corr (B,F,16,16,16,16) [b,f,h_t,w_t,h_r,w_r] -> flow (B,F,16,16,2).
        The reference feeds reshape(b*f, 256, 16, 16) = channels over the target
        position, pixels over the reference position: in NHWC that is
        [b,f,h_r,w_r,(h_t,w_t)] = the layout SeparableConv4d produced."""
        b, f, h, w = corr.shape[:4]
        x4 = corr.permute(0, 1, 4, 5, 2, 3).reshape(b * f, 16, 16, h * w).contiguous()
        specs, params = _specs_from(self.mixer)
        out = E.conv_stack(x4, specs, params)                  # (b*f,16,16,4)
        return out[..., :2].reshape(b, f, h, w, 2)


class FlowEstimator(nn.Module):
    """model_dfpn.py:677-743."""

    def __init__(self, in_c=10):
        super().__init__()
        self.nn = nn.Sequential(
            nn.Conv2d(in_c, 128, (5, 5), (1, 1), 2), nn.ReLU(),
            nn.Conv2d(128, 128, (3, 3), (1, 1), 1), nn.ReLU(),
            nn.Conv2d(128, 128, (5, 5), (2, 2), 2), nn.ReLU(),
            nn.Conv2d(128, 128, (5, 5), (1, 1), 2), nn.ReLU(),
            nn.Conv2d(128, 128, (3, 3), (1, 1), 1), nn.ReLU(),
            nn.Conv2d(128, 128, (3, 3), (2, 2), 1), nn.ReLU(),
            nn.Conv2d(128, 128, (5, 5), (1, 1), 2), nn.ReLU(),
            nn.Conv2d(128, 128, (3, 3), (1, 1), 1), nn.ReLU(),
            nn.Conv2d(128, 128, (3, 3), (2, 2), 1), nn.ReLU(),
            nn.Conv2d(128, 128, (5, 5), (1, 1), 2), nn.ReLU(),
            nn.Conv2d(128, 128, (3, 3), (1, 1), 1), nn.ReLU(),
            nn.ConvTranspose2d(128, 128, (3, 3), (2, 2), (1, 1), (1, 1)), nn.ReLU(),
            nn.Conv2d(128, 128, (5, 5), (1, 1), 2), nn.ReLU(),
            nn.Conv2d(128, 128, (3, 3), (1, 1), 1), nn.ReLU(),
            nn.ConvTranspose2d(128, 128, (3, 3), (2, 2), (1, 1), (1, 1)), nn.ReLU(),
            nn.Conv2d(128, 128, (5, 5), (1, 1), 2), nn.ReLU(),
            nn.Conv2d(128, 128, (3, 3), (1, 1), 1), nn.ReLU(),
            nn.ConvTranspose2d(128, 128, (5, 5), (2, 2), (2, 2), (1, 1)), nn.ReLU(),
            nn.Conv2d(128, 128, (5, 5), (1, 1), 2), nn.ReLU(),
            nn.Conv2d(128, 128, (3, 3), (1, 1), 1), nn.ReLU(),
            nn.Conv2d(128, 2, (3, 3), (1, 1), 1))

    def forward(self, x_target, m_target, x_refs, m_refs, flow_pre):
        """model_dfpn.py:713-743: inputs are already normalised."""
        return self._forward_raw(x_target, m_target, x_refs, m_refs, flow_pre, None, None)

    def _forward_raw(self, x_target, m_target, x_refs, m_refs, flow_pre, scale, shift):
        b, c, ref_n, h, w = x_refs.size()
        nn_input = ops.PackFlowEstimatorFn.apply(x_target, m_target, x_refs, m_refs, flow_pre, scale, shift)
        specs, params = _specs_from(self.nn)
        specs[0].dx_window = (8, 2)       # PackFlowEstimatorFn reads the gradient of the two flow channels only
        out = E.conv_stack(nn_input, specs, params)            # (b*r,h,w,4)
        return out[..., :2].reshape(b, ref_n, h, w, 2)
