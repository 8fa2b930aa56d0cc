"""Model factory with the reference's contract: ``MODELS[cfg.MODEL.NAME](cfg, is_train)``.

Mirrors deep_hrnet/lib/models/__init__.py:28-41 and pose_hrnet.py:508-514
(``get_pose_net``): the returned object takes the reference's state_dict
(``load_state_dict``, ``module.`` prefixes stripped as pose_engine.py:107-117
does), ``.to(device)``, ``.eval()`` and is called on an NCHW fp32 batch,
returning the heat-map tensor ``[N, J*(3 if offset else 1), H/4, W/4]`` fp32
on the same device.  The forward runs entirely in libudp_pose_hip.so.
"""
import ctypes as C

import torch

from . import _lib
from .hrnet_plan import HRNetProgram, storage_bytes
from .synth import hrnet_param_shapes


def _get(cfg, *path):
    cur = cfg
    for p in path:
        cur = cur[p] if isinstance(cur, dict) else getattr(cur, p)
    return cur


class PoseHighResolutionNetHip:
    """HRNet (UDP variant) inference on MI355X through the C ABI."""

    MAX_IO_SHAPES = 8      # persistent (input, output) buffer pairs kept, least recently used evicted (= the
                           # library's hipGraph cache size: one graph per buffer set)

    def __init__(self, cfg, dtype="f32", psa=False):
        self.psa = bool(psa)
        self.cfg = cfg
        self.training = False
        self._trainer = None         # train.HRNetTrainer over the same weights (created by .train())
        self._stale = False          # legacy flag (function.train sets it); the trainer's version counter decides
        self._seen_version = 0       # trainer.version the inference program / state_dict were last synced at
        self.extra = _get(cfg, "MODEL", "EXTRA")
        self.num_joints = int(_get(cfg, "MODEL", "NUM_JOINTS"))
        self.target_type = _get(cfg, "MODEL", "TARGET_TYPE")
        self.dtype = dtype
        self.device = None
        self.use_graph = True
        self.max_images_per_launch = None     # None: only the 2 GiB-per-tensor limit of one launch applies
        self._sd = None
        self._compiled = {}      # (h, w) -> (handle, blob tensor, program)
        self._ws = None
        self._io = {}
        # same checks as HighResolutionModule._check_branches (pose_hrnet.py:121-139)
        hrnet_param_shapes(self.extra, self.num_joints, self.target_type)

    # ---- nn.Module-like surface used by the reference's callers
    def load_state_dict(self, state_dict, strict=True):
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        want = hrnet_param_shapes(self.extra, self.num_joints, self.target_type, psa=self.psa)
        missing = [k for k in want if k not in sd and not k.endswith("num_batches_tracked")]
        unexpected = [k for k in sd if k not in want]
        if strict and (missing or unexpected):
            raise RuntimeError("state_dict mismatch: missing %s unexpected %s" % (missing[:5], unexpected[:5]))
        if missing:
            raise RuntimeError("state_dict lacks %d tensors the forward needs, e.g. %s" % (len(missing), missing[:3]))
        for k, shape in want.items():
            if k in sd and tuple(sd[k].shape) != tuple(shape):
                raise RuntimeError("size mismatch for %s: %s vs %s" % (k, tuple(sd[k].shape), tuple(shape)))
        self._sd = sd
        self._release()
        return self

    def state_dict(self):
        """Reference-format state_dict; after training steps it is read back from the trainer's flat buffer."""
        if self._trainer_moved():
            self._sync_from_trainer()
        return dict(self._sd or {})

    def init_weights(self, pretrained=""):
        """pose_hrnet.py:473-505: conv weights ~ N(0, 0.001), conv biases 0, BatchNorm (weight 1, bias 0; running
        statistics at their nn.BatchNorm2d defaults 0 / 1), then the layers of a pretrained file named by
        MODEL.EXTRA.PRETRAINED_LAYERS ('*' = all; 'stage4.2.fuse_layers' keys skipped as :492-493 does).  Draws come
        from torch's global generator like nn.init.normal_ (module iteration order is not reproduced, so the
        values differ from the reference's for the same seed; the distribution is the same)."""
        import os
        shapes = hrnet_param_shapes(self.extra, self.num_joints, self.target_type, psa=self.psa)
        sd = {}
        for k, shape in shapes.items():
            if k.endswith("num_batches_tracked"):
                sd[k] = torch.zeros((), dtype=torch.int64)
            elif k.endswith("running_var"):
                sd[k] = torch.ones(shape)
            elif k.endswith("running_mean"):
                sd[k] = torch.zeros(shape)
            elif len(shape) == 4:
                sd[k] = torch.randn(shape) * 0.001
            elif k.endswith(".weight") and (k[:-7] + ".running_mean") in shapes:
                sd[k] = torch.ones(shape)                 # BatchNorm weight
            else:
                sd[k] = torch.zeros(shape)                # BatchNorm / conv bias
        if pretrained and os.path.isfile(pretrained):
            layers = list(self.extra.get("PRETRAINED_LAYERS", ["*"]))
            for name, v in torch.load(pretrained, map_location="cpu", weights_only=True).items():
                name = name[7:] if name.startswith("module.") else name
                if "stage4.2.fuse_layers" in name or name not in sd:
                    continue
                if layers[0] == "*" or name.split(".")[0] in layers:
                    if tuple(v.shape) != tuple(sd[name].shape):
                        raise RuntimeError("size mismatch for %s: %s vs %s" % (name, tuple(v.shape), tuple(sd[name].shape)))
                    sd[name] = v
        elif pretrained:
            raise ValueError("%s is not exist!" % pretrained)       # pose_hrnet.py:503-505
        self._sd = sd
        self._trainer, self._seen_version = None, 0
        self._release()
        return self

    # ---- training surface (function.train drives it; tools/train.py:91,116-125)
    def trainer(self):
        """The HRNetTrainer over this model's weights (created on first use from the current state_dict)."""
        if self._trainer is None:
            if self.psa:
                raise NotImplementedError("pose_hrnet_psa: the training step does not cover the attention ops")
            if self._sd is None:
                raise RuntimeError("load_state_dict() or init_weights() first")
            from .train import HRNetTrainer
            if self.device is None:
                self.to("cuda")
            self._trainer = HRNetTrainer(self.cfg, self._sd, device=self.device, dtype="bf16" if self.dtype == "bf16" else "f32")
        return self._trainer

    def parameters(self):
        """What ``optim.Adam(model.parameters(), lr=...)`` (utils.py:70-74) takes: ONE flat fp32 device tensor
        holding every parameter in named_parameters() order (the trainer's master copy)."""
        t = self.trainer()
        return [t.flat[:t._n_param]]

    def _trainer_moved(self):
        """True when the trainer has stepped (by whatever route: function.train, train_step, adam_step) since the
        compiled inference program and ``_sd`` were last read from it."""
        return self._trainer is not None and (self._stale or self._trainer.version != self._seen_version)

    def _sync_from_trainer(self):
        self._sd = self._trainer.state_dict()
        self._stale = False
        self._seen_version = self._trainer.version
        self._release()

    def to(self, device):
        self.device = torch.device(device)
        return self

    def cuda(self):
        return self.to("cuda")

    def eval(self):
        return self.train(False)

    def train(self, mode=True):
        """nn.Module.train(): in training mode ``model(x)`` is the train-mode forward of the trainer (batch
        statistics, tape kept for the backward); in eval mode the compiled inference program."""
        if mode:
            self.trainer()
        self.training = bool(mode)
        return self

    # ---- compile / run
    def _release(self):
        for h, _, _ in self._compiled.values():
            _lib.lib().udp_hrnet_destroy(h)
        self._compiled = {}
        self._io = {}

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _compile(self, h, w):
        if self._sd is None:
            raise RuntimeError("load_state_dict() first")
        prog = self._make_program(h, w)
        blob = torch.from_numpy(prog.weight_blob()).to(self.device)
        ops = prog.ops_array()
        bufs = (C.c_int64 * len(prog.buf_elems))(*prog.buf_elems)
        handle = C.c_void_p()
        _lib.check(_lib.lib().udp_hrnet_create(ops, len(ops), bufs, len(prog.buf_elems), _lib.ptr(blob),
                                               blob.numel(), _lib.DTYPES[self.dtype],
                                               h, w, prog.out_channels, C.byref(handle)))
        self._compiled[(h, w)] = (handle, blob, prog)
        return self._compiled[(h, w)]

    def _make_program(self, h, w):
        return HRNetProgram(self._sd, self.extra, h, w, self.dtype)

    def program(self, h, w):
        if self.device is None:
            self.to("cuda")
        return (self._compiled.get((h, w)) or self._compile(h, w))[2]

    def io_buffers(self, n, h, w, flip_test=False):
        """Persistent (input [n,3,h,w], heat-maps [n*(1|2),C,h/4,w/4]) device buffers for this
        shape.  Launch sequences are replayed as hipGraphs keyed on these addresses, so the
        forward reads the network input from / writes heat-maps to the same memory every call."""
        if self.device is None:
            self.to("cuda")
        key = (n, h, w, bool(flip_test))
        io = self._io.pop(key, None)
        if io is None:
            prog = self.program(h, w)
            b = n * (2 if flip_test else 1)
            io = (torch.empty(n, 3, h, w, dtype=torch.float32, device=self.device),
                  torch.empty(b, prog.out_channels, h // 4, w // 4, dtype=torch.float32, device=self.device))
            while len(self._io) >= self.MAX_IO_SHAPES:     # least recently used first (dicts keep insertion order)
                self._io.pop(next(iter(self._io)))
        self._io[key] = io                                  # (re-)inserted last = most recently used
        return io

    def raw_forward(self, x, flip_test=False):
        """x: cuda fp32 [N,3,H,W] -> heat-maps [N*(2 if flip_test else 1), C, H/4, W/4]
        (rows N.. are the raw outputs for the W-mirrored inputs).  The returned tensor is the
        model's persistent output buffer: it is overwritten by the next call of the same shape
        (the reference's callers ``.clone()`` it, pose_engine.py:125)."""
        if self.device is None:
            self.to(x.device)
        if not x.is_cuda:
            raise RuntimeError("udp-pose_amd has no CPU path: the input must live on the GPU")
        if self._trainer_moved():
            self._sync_from_trainer()                      # weights moved since the program was compiled
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected fp32 [N,3,H,W], got %s %s" % (x.dtype, tuple(x.shape)))
        n, _, h, w = x.shape
        if n < 1:
            raise ValueError("empty batch (the reference's torch.stack of no crops raises too)")
        handle, _, prog = self._compiled.get((h, w)) or self._compile(h, w)
        # one launch addresses a tensor with 32-bit byte offsets: split batches whose largest activation
        # ([2N, H/2, W/2, 64]) would pass 2 GiB
        cap = max(1, (2 ** 31 - 1) // (max(t.elems for t in prog._tensors) * storage_bytes(self.dtype)
                                      * (2 if flip_test else 1)))
        cap = min(cap, self.max_images_per_launch or cap)
        if n > cap:
            parts = [self.raw_forward(x[i:i + cap].contiguous(), flip_test).clone() for i in range(0, n, cap)]
            if not flip_test:
                return torch.cat(parts)
            sizes = [min(cap, n - i) for i in range(0, n, cap)]
            return torch.cat([p[:k] for p, k in zip(parts, sizes)] + [p[k:] for p, k in zip(parts, sizes)])
        xin, out = self.io_buffers(n, h, w, flip_test)
        if x.data_ptr() != xin.data_ptr():
            xin.copy_(x)
        lib = _lib.lib()
        need = lib.udp_hrnet_workspace_bytes(handle, n, int(flip_test))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        _lib.check(lib.udp_hrnet_forward(handle, _lib.ptr(xin), n, int(flip_test), _lib.ptr(self._ws),
                                         self._ws.numel(), _lib.ptr(out), int(self.use_graph), _lib.stream_ptr()))
        return out

    def profile(self, x, flip_test=False):
        """Per-op elapsed milliseconds (hipEvents around every launch, eager) and the op table:
        returns (ms ndarray [n_ops], describe() list)."""
        import numpy as np
        n, _, h, w = x.shape
        handle, _, prog = self._compiled.get((h, w)) or self._compile(h, w)
        xin, out = self.io_buffers(n, h, w, flip_test)
        if x.data_ptr() != xin.data_ptr():
            xin.copy_(x)
        lib = _lib.lib()
        need = lib.udp_hrnet_workspace_bytes(handle, n, int(flip_test))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        nops = lib.udp_hrnet_num_launches(handle)
        ms = (C.c_float * nops)()
        _lib.check(lib.udp_hrnet_profile(handle, _lib.ptr(xin), n, int(flip_test), _lib.ptr(self._ws),
                                         self._ws.numel(), _lib.ptr(out), ms, _lib.stream_ptr()))
        return np.frombuffer(ms, dtype=np.float32).copy(), prog.describe()

    def __call__(self, x):
        if self.training:
            return self.trainer().forward(x.contiguous())
        return self.raw_forward(x, flip_test=False)

    forward = __call__


class RSN18Hip(PoseHighResolutionNetHip):
    """RSN-18 (RSN/exps/RSN18.coco/network.py, STAGE_NUM = 1) inference through the same C ABI;
    ``state_dict`` in the reference module's key format; returns ``outputs[-1][-1]`` ([N,C,H/4,W/4])."""

    def __init__(self, out_channels=17, dtype="f32", chl_num=256):
        self.out_channels = int(out_channels)
        self.chl_num = chl_num
        self.dtype = dtype
        self.psa, self.cfg = False, None
        self.training, self._trainer, self._stale = False, None, False
        self.device = None
        self.use_graph = True
        self.max_images_per_launch = None     # None: only the 2 GiB-per-tensor limit of one launch applies
        self._sd = None
        self._compiled = {}
        self._ws = None
        self._io = {}

    def load_state_dict(self, state_dict, strict=True):
        from .synth import rsn18_param_shapes
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        want = rsn18_param_shapes(self.out_channels, self.chl_num)
        missing = [k for k in want if k not in sd and not k.endswith("num_batches_tracked")]
        unexpected = [k for k in sd if k not in want]
        if missing or (strict and unexpected):
            raise RuntimeError("state_dict mismatch: missing %s unexpected %s" % (missing[:5], unexpected[:5]))
        for k, shape in want.items():
            if k in sd and tuple(sd[k].shape) != tuple(shape):
                raise RuntimeError("size mismatch for %s: %s vs %s" % (k, tuple(sd[k].shape), tuple(shape)))
        self._sd = sd
        self._release()
        return self

    def trainer(self):
        raise NotImplementedError("RSN-18: the training step covers pose_hrnet only")

    def _make_program(self, h, w):
        from .rsn_plan import RSNProgram
        return RSNProgram(self._sd, h, w, self.dtype, self.chl_num)


def get_pose_net(cfg, is_train, **kwargs):
    """pose_hrnet.py:508-514: ``init_weights(cfg.MODEL.PRETRAINED)`` when ``is_train and cfg.MODEL.INIT_WEIGHTS``."""
    model = PoseHighResolutionNetHip(cfg, **kwargs)
    if is_train:
        try:
            init = bool(_get(cfg, "MODEL", "INIT_WEIGHTS"))
        except (KeyError, AttributeError):
            init = False
        if init:
            try:
                pretrained = _get(cfg, "MODEL", "PRETRAINED")
            except (KeyError, AttributeError):
                pretrained = ""
            model.init_weights(pretrained or "")
    return model


def get_pose_net_psa(cfg, is_train, **kwargs):
    """pose_hrnet_psa.py get_pose_net: same plan, every BasicBlock carries "<block>.deattn.*" (PSA_s)
    parameters and the planner emits the attention ops for them."""
    return get_pose_net(cfg, is_train, psa=True, **kwargs)


MODELS = {"pose_hrnet": get_pose_net, "pose_hrnet_psa": get_pose_net_psa}
