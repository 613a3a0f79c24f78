"""Drop-in ``InpaintGenerator`` modules (reference: model/e2fgvi.py:133-263, model/e2fgvi_hq.py:134-263).

The module tree below exists to own the parameters under exactly the reference's checkpoint names
(243 / 244 ``state_dict`` entries) so that ``load_state_dict`` of released or
reference-made checkpoints works unchanged and ``test.py``-style callers
(``importlib.import_module('model.' + name).InpaintGenerator()``, ``.to(device)``, ``.eval()``,
``model(masked_imgs, n_local)``) need no edits.  The arithmetic is NOT done by these torch modules:
``forward`` hands the parameters to ``engine.Engine``, which runs the hand-written HIP kernels.
Inference only (the reference trains with autograd; training is out of scope here).
"""
import math

import torch
import torch.nn as nn

from .synth import rolled_valid_index


# ----------------------------------------------------------------------------- parameter containers
class _Holder(nn.Module):
    def forward(self, *a, **k):   # pragma: no cover - containers are not callable on their own
        raise RuntimeError("parameter container: the forward pass runs in e2fgvi_amd.engine.Engine")


class Encoder(_Holder):
    def __init__(self):
        super().__init__()
        cfg = ((3, 64, 2, 1), (64, 64, 1, 1), (64, 128, 2, 1), (128, 256, 1, 1), (256, 384, 1, 1),
               (640, 512, 1, 2), (768, 384, 1, 4), (640, 256, 1, 8), (512, 128, 1, 1))
        layers = []
        for cin, cout, s, g in cfg:
            layers += [nn.Conv2d(cin, cout, 3, s, 1, groups=g), nn.LeakyReLU(0.2, inplace=True)]
        self.layers = nn.ModuleList(layers)


class deconv(_Holder):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, 1, 1)


class SecondOrderDeformableAlignment(_Holder):
    """weight/bias of the modulated deformable conv + the conv_offset stack (feat_prop.py:13-33)."""

    def __init__(self, cin=256, cout=128, deform_groups=16):
        super().__init__()
        self.in_channels, self.out_channels, self.deform_groups = cin, cout, deform_groups
        self.kernel_size, self.stride, self.padding, self.dilation, self.groups = (3, 3), 1, 1, 1, 1
        self.max_residue_magnitude = 10
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3))
        self.bias = nn.Parameter(torch.zeros(cout))
        stdv = 1.0 / math.sqrt(cin * 9)
        self.weight.data.uniform_(-stdv, stdv)
        self.conv_offset = nn.Sequential(
            nn.Conv2d(3 * cout + 4, cout, 3, 1, 1), nn.LeakyReLU(0.1, inplace=True),
            nn.Conv2d(cout, cout, 3, 1, 1), nn.LeakyReLU(0.1, inplace=True),
            nn.Conv2d(cout, cout, 3, 1, 1), nn.LeakyReLU(0.1, inplace=True),
            nn.Conv2d(cout, 27 * deform_groups, 3, 1, 1))
        self.init_offset()

    def init_offset(self):
        nn.init.constant_(self.conv_offset[-1].weight, 0)
        nn.init.constant_(self.conv_offset[-1].bias, 0)


class BidirectionalPropagation(_Holder):
    def __init__(self, channel=128):
        super().__init__()
        self.channel = channel
        self.deform_align = nn.ModuleDict()
        self.backbone = nn.ModuleDict()
        for i, name in enumerate(("backward_", "forward_")):
            self.deform_align[name] = SecondOrderDeformableAlignment(2 * channel, channel, 16)
            self.backbone[name] = nn.Sequential(nn.Conv2d((2 + i) * channel, channel, 3, 1, 1),
                                                nn.LeakyReLU(0.1, inplace=True),
                                                nn.Conv2d(channel, channel, 3, 1, 1))
        self.fusion = nn.Conv2d(2 * channel, channel, 1, 1, 0)


class SoftSplit(_Holder):
    def __init__(self):
        super().__init__()
        self.embedding = nn.Linear(49 * 128, 512)


class SoftComp(_Holder):
    def __init__(self, hq):
        super().__init__()
        self.embedding = nn.Linear(512, 49 * 128)
        if hq:
            self.bias_conv = nn.Conv2d(128, 128, 3, 1, 1)
        else:
            self.bias = nn.Parameter(torch.zeros(128, 60, 108))


class WindowAttention(_Holder):
    def __init__(self):
        super().__init__()
        self.register_buffer("valid_ind_rolled", rolled_valid_index())
        self.qkv = nn.Linear(512, 1536)
        self.proj = nn.Linear(512, 512)


class FusionFeedForward(_Holder):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Linear(512, 1960))
        self.conv2 = nn.Sequential(nn.GELU(), nn.Linear(1960, 512))


class TemporalFocalTransformerBlock(_Holder):
    def __init__(self):
        super().__init__()
        self.pool_layers = nn.ModuleList([nn.Linear(45, 1)])
        self.pool_layers[0].weight.data.fill_(1.0 / 45)
        self.pool_layers[0].bias.data.fill_(0)
        self.norm1 = nn.LayerNorm(512)
        self.attn = WindowAttention()
        self.norm2 = nn.LayerNorm(512)
        self.mlp = FusionFeedForward()


class _ConvModule(_Holder):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 7, 1, 3)
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode="fan_out", nonlinearity="relu")
        nn.init.constant_(self.conv.bias, 0)


class SPyNetBasicModule(_Holder):
    def __init__(self):
        super().__init__()
        self.basic_module = nn.Sequential(*[_ConvModule(a, b) for a, b in ((8, 32), (32, 64), (64, 32), (32, 16), (16, 2))])


class SPyNet(_Holder):
    """Weights only.  The reference downloads pretrained SPyNet weights in its constructor
    (flow_comp.py:59-72); there is no network here, so they come with the checkpoint."""

    def __init__(self):
        super().__init__()
        self.basic_module = nn.ModuleList([SPyNetBasicModule() for _ in range(6)])
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))


# ----------------------------------------------------------------------------- generator
class _InpaintGeneratorBase(nn.Module):
    MODEL = "e2fgvi"

    def __init__(self, init_weights=True):
        super().__init__()
        hq = self.MODEL == "e2fgvi_hq"
        self.encoder = Encoder()
        self.decoder = nn.Sequential(deconv(128, 128), nn.LeakyReLU(0.2, inplace=True), nn.Conv2d(128, 64, 3, 1, 1),
                                     nn.LeakyReLU(0.2, inplace=True), deconv(64, 64), nn.LeakyReLU(0.2, inplace=True),
                                     nn.Conv2d(64, 3, 3, 1, 1))
        self.feat_prop_module = BidirectionalPropagation(128)
        self.ss = SoftSplit()
        self.sc = SoftComp(hq)
        self.transformer = nn.Sequential(*[TemporalFocalTransformerBlock() for _ in range(8)])
        self._engine = None
        self._engine_key = None
        if init_weights:
            self.init_weights()
            for m in self.modules():         # e2fgvi.py:202-205
                if isinstance(m, SecondOrderDeformableAlignment):
                    m.init_offset()
        self.update_spynet = SPyNet()        # built after init_weights, like the reference (e2fgvi.py:208)
        # "fp32" (default, the parity configuration) or "bf16" (optional bf16-MFMA mode for the HQ configurations)
        self.precision = "fp32"

    def init_weights(self, init_type="normal", gain=0.02):
        """The reference's BaseNetwork.init_weights (e2fgvi.py:29-68): every Conv*/Linear* weight is re-drawn with
        ``init_type`` in normal | xavier | xavier_uniform | kaiming | orthogonal | none and its bias zeroed; the
        deformable conv's main weight is not touched (the reference matches on class names containing 'Conv' /
        'Linear', SecondOrderDeformableAlignment has neither).  The constructor then re-applies init_offset
        (e2fgvi.py:200-205): conv_offset[-1] is zero."""
        for m in self.modules():
            if not isinstance(m, (nn.Conv2d, nn.Linear)):
                continue
            if init_type == "normal":
                nn.init.normal_(m.weight.data, 0.0, gain)
            elif init_type == "xavier":
                nn.init.xavier_normal_(m.weight.data, gain=gain)
            elif init_type == "xavier_uniform":
                nn.init.xavier_uniform_(m.weight.data, gain=1.0)
            elif init_type == "kaiming":
                nn.init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
            elif init_type == "orthogonal":
                nn.init.orthogonal_(m.weight.data, gain=gain)
            elif init_type == "none":
                m.reset_parameters()
            else:
                raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            if m.bias is not None:
                nn.init.constant_(m.bias.data, 0.0)
        self.refresh_engine()

    def print_network(self):
        n = sum(p.numel() for p in self.parameters())
        print("Network [%s] was created. Total number of parameters: %.1f million." % (type(self).__name__, n / 1e6))

    # -- engine cache -------------------------------------------------------------------------
    # The engine holds re-laid-out COPIES of the weights.  It is dropped whenever this module's parameters can have
    # changed through the nn.Module API (load_state_dict, .to()/.half()/.cuda() -> _apply, init_weights) and whenever
    # the fingerprint below moves (optimizer-style in-place updates bump Parameter._version; `.data` swaps change a
    # data_ptr).  Writes through `p.data.copy_()` are invisible to both: call refresh_engine() after such edits.
    def refresh_engine(self):
        """Drop the cached device plan; the next forward re-reads every parameter."""
        self._engine = None
        self._engine_key = None

    def _apply(self, fn, *a, **k):
        self.refresh_engine()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict=True, **k):
        self.refresh_engine()
        return super().load_state_dict(state_dict, strict=strict, **k)

    def load_checkpoint(self, path, map_location=None, strict=True):
        """Load a checkpoint FILE in any of the layouts the reference ecosystem writes: a bare generator state_dict
        (release_model/E2FGVI-CVPR22.pth, E2FGVI-HQ-CVPR22.pth -- test.py:119-120 does torch.load + load_state_dict),
        the trainer's gen_*.pth (also bare, core/trainer.py:240), a ``{'state_dict': ...}`` wrapper (mmcv style -- the
        SPyNet file flow_comp.py:59-72 downloads), and keys carrying a DataParallel / DDP ``module.`` prefix.  A
        SPyNet-only file (keys ``basic_module.*``) is loaded into ``update_spynet``."""
        sd = torch.load(path, map_location=map_location or "cpu")
        if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
            sd = sd["state_dict"]
        if not isinstance(sd, dict) or not sd:
            raise ValueError("%s does not contain a state_dict" % path)
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        if all(k.startswith(("basic_module.", "mean", "std")) for k in sd):
            self.refresh_engine()
            return self.update_spynet.load_state_dict(sd, strict=strict)
        return self.load_state_dict(sd, strict=strict)

    def train(self, mode=True):
        """nn.Module.train() for generic tooling (wrappers that restore the training flag after an eval pass, Lightning ...):
        this module is the INFERENCE forward -- the parameter containers have no autograd path -- so the flag
        stays False; asking for training mode warns once instead of raising."""
        if mode and not getattr(self, "_warned_train", False):
            import warnings
            warnings.warn("this InpaintGenerator is the MI355X inference forward: .train() keeps it in eval mode (train with "
                          "the reference implementation and load the checkpoint here: load_state_dict / load_checkpoint)")
            self._warned_train = True
        return super().train(False)

    def _fingerprint(self):
        ps = list(self.parameters()) + list(self.buffers())
        return (str(ps[0].device), tuple(p._version for p in ps), tuple(p.data_ptr() for p in ps), self.precision)

    def engine(self):
        from .engine import Engine
        key = self._fingerprint()
        if self._engine is None or key != self._engine_key:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("InpaintGenerator runs only on an MI355X (ROCm 'cuda') device: move the module "
                                   "with .to('cuda'); there is no CPU path")
            self._engine = Engine(self.state_dict(), self.MODEL, dev, precision=self.precision)
            self._engine_key = key
        return self._engine

    def forward_bidirect_flow(self, masked_local_frames):
        """[b,l_t,3,H,W] in [0,1] -> (flows_forward, flows_backward), each [b,l_t-1,2,H/4,W/4]."""
        from . import ops
        eng = self.engine()
        b, l_t, c, H, W = masked_local_frames.shape
        with torch.no_grad():
            fwd, bwd = eng.flows(masked_local_frames.float() * 2 - 1, l_t)
            h, w = H // 4, W // 4
            return (ops.nhwc_to_nchw(fwd.reshape(-1, h, w, 2)).view(b, l_t - 1, 2, h, w),
                    ops.nhwc_to_nchw(bwd.reshape(-1, h, w, 2)).view(b, l_t - 1, 2, h, w))

    def forward(self, masked_frames, num_local_frames):
        """masked_frames: float32 [b,t,3,H,W] in [-1,1]; returns (frames [b*t,3,H,W], (flow_fwd, flow_bwd))."""
        if not masked_frames.is_cuda:
            raise RuntimeError("masked_frames must live on the MI355X (cuda) device; there is no CPU path")
        with torch.no_grad():
            return self.engine().forward(masked_frames, int(num_local_frames))


class InpaintGenerator(_InpaintGeneratorBase):
    """Fixed-resolution 432x240 model (reference model/e2fgvi.py:133)."""
    MODEL = "e2fgvi"


class InpaintGeneratorHQ(_InpaintGeneratorBase):
    """Arbitrary-resolution model (reference model/e2fgvi_hq.py:134)."""
    MODEL = "e2fgvi_hq"


# ----------------------------------------------------------------------------- training-side names (stubs)
class Discriminator(nn.Module):
    """The reference's T-PatchGAN discriminator (model/e2fgvi.py:271-344) belongs to training (core/trainer.py),
    which is out of scope of this inference implementation."""

    def __init__(self, *a, **k):
        raise NotImplementedError("model.e2fgvi.Discriminator is training-only; this package implements the MI355X "
                                  "inference forward (InpaintGenerator).  Train with the reference implementation.")


def spectral_norm(module, mode=True):
    """model/e2fgvi.py:347-350 -- only the discriminator uses it (training-only, see Discriminator)."""
    raise NotImplementedError("spectral_norm is training-only; see Discriminator")
