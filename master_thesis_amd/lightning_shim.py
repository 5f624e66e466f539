"""``pl.LightningModule`` when pytorch_lightning is importable, otherwise a
minimal stand-in with the members the reference modules use
(model_dfpn.py:14,406-409,452-456): ``log``, ``logger``, ``current_epoch``,
``load_from_checkpoint``."""
import torch

try:  # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl
    LightningModule = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    HAVE_LIGHTNING = False

    class LightningModule(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.logged = {}
            self.logger = None
            self.current_epoch = 0

        def log(self, name, value, *args, **kwargs):
            # detached: a logged loss must not keep its autograd graph alive
            self.logged[name] = value.detach() if torch.is_tensor(value) else value

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **kwargs):
            ckpt = torch.load(checkpoint_path, map_location=map_location or 'cpu')
            model = cls(**kwargs)
            model.load_state_dict(ckpt.get('state_dict', ckpt), strict=strict)
            return model
