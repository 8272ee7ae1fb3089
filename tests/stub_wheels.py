"""Stand-ins for the wheels the pinning kit (tools/pin_oracles.py) drives — TEST INFRASTRUCTURE for the kit's own plumbing test.

ultralytics / spandrel / diffusers cannot be installed in the build image, so the kit's code path (build the library model from the
oracle's seeded state dict -> run it -> store its outputs -> replay the oracle against the stored arrays) is exercised with objects
that have the wheels' call shapes and are backed by the oracle classes themselves.  That says nothing about parity (it is circular by
construction); it proves the kit runs end to end, writes loadable fixtures and that the replay reads them.  The OpenCV stand-in is
different: tests/independent_cv.py holds independent implementations."""
import types
from collections import namedtuple

import torch

_Keys = namedtuple("_IncompatibleKeys", ["missing_keys", "unexpected_keys"])


def _load(net, sd):
    own = net.state_dict()
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own]
    net.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    return _Keys(missing, unexpected)


def ultralytics():
    from oracle import yolo11_ref, yolo_ref

    class _Model(torch.nn.Module):
        def __init__(self, cfg, ch=3, nc=1, verbose=False):
            super().__init__()
            name = str(cfg)
            seg = "-seg" in name
            if name.startswith("yolov8"):
                self.net = yolo_ref.make_model(name[6], nc, seed=12345)               # its own random start; the kit loads the oracle's weights over it
            else:
                fam = "12" if name.startswith("yolo12") else "11"
                self.net = yolo11_ref.make_model(fam, name[6], nc, seg, seed=12345)
            self.seg = seg

        def fuse(self):
            return self

        def load_state_dict(self, sd, strict=True):
            return _load(self.net, sd)

        def forward(self, x):
            y = self.net(x)
            ys = list(y) if isinstance(y, (tuple, list)) else [y]
            return (ys[0], (ys[1:], None)) if len(ys) > 1 else (ys[0], None)          # nested like ultralytics' eval output

    tasks = types.SimpleNamespace(DetectionModel=_Model, SegmentationModel=_Model)
    return types.SimpleNamespace(__name__="ultralytics_stub", __version__="stub-0", nn=types.SimpleNamespace(tasks=tasks))


def spandrel():
    from oracle import rcan_ref

    class _Desc:
        def __init__(self, sd):
            self.model = rcan_ref.load_ref(sd)
            self.scale = rcan_ref.rcan_hparams(sd)["scale"]
            self.architecture = types.SimpleNamespace(name="RCAN")

        def __call__(self, x):
            return self.model(x)

    class ModelLoader:
        def load_from_state_dict(self, sd):
            return _Desc(sd)

    return types.SimpleNamespace(__name__="spandrel_stub", __version__="stub-0", ModelLoader=ModelLoader)


def diffusers():
    from oracle import flux_ref as fr

    class FluxTransformer2DModel(torch.nn.Module):
        def __init__(self, patch_size=1, in_channels=64, num_layers=19, num_single_layers=38, attention_head_dim=128, num_attention_heads=24,
                     joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True, axes_dims_rope=(16, 56, 56)):
            super().__init__()
            self.net = fr.FluxTransformer(d=attention_head_dim * num_attention_heads, heads=num_attention_heads, layers=num_layers,
                                          single_layers=num_single_layers, in_channels=in_channels, joint_dim=joint_attention_dim,
                                          pooled_dim=pooled_projection_dim, axes_dim=tuple(axes_dims_rope))

        def load_state_dict(self, sd, strict=True):
            return _load(self.net, sd)

        def forward(self, hidden_states, timestep, guidance, pooled_projections, encoder_hidden_states, txt_ids, img_ids, return_dict=False):
            y = self.net(hidden_states[0], float(timestep[0]), float(guidance[0]), pooled_projections[0], encoder_hidden_states[0], txt_ids, img_ids)
            return (y[None],)

    class AutoencoderKL(torch.nn.Module):
        def __init__(self, block_out_channels=(128, 256, 512, 512), latent_channels=16, norm_num_groups=32, scaling_factor=0.3611,
                     shift_factor=0.1159, **_):
            super().__init__()
            self.net = fr.VAE(ch=tuple(block_out_channels), latent=latent_channels, groups=norm_num_groups, scaling_factor=scaling_factor,
                              shift_factor=shift_factor)

        def load_state_dict(self, sd, strict=True):
            return _load(self.net, sd)

        def encode(self, x):
            mean = self.net.encode_mode(x)
            return types.SimpleNamespace(latent_dist=types.SimpleNamespace(mode=lambda: mean))

        def decode(self, z, return_dict=False):
            return (self.net.decode(z),)

    return types.SimpleNamespace(__name__="diffusers_stub", __version__="stub-0", FluxTransformer2DModel=FluxTransformer2DModel,
                                 AutoencoderKL=AutoencoderKL)
