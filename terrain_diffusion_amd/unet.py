"""EDMUnet2D with the reference's call signature, executed by the HIP engine.

Mirrors terrain_diffusion/models/edm_unet.py:15-184: same constructor kwargs, same state-dict names
(diffusers layout, SURVEY.md §8b face 3), `model(x, noise_labels=..., conditional_inputs=[...])`.
"""
import ctypes as C
import json
import os

import numpy as np
import torch

from ._lib import UnetConfig, lib, check
from .engine import get_engine, ptr, f32

DTYPES = {"fp32": 0, "float32": 0, None: 0, torch.float32: 0, "bf16": 1, "bfloat16": 1, torch.bfloat16: 1, "fp16": 2, "float16": 2, torch.float16: 2}


class EDMUnet2D:
    def __init__(self, image_size, in_channels, out_channels=None, model_channels=128, model_channel_mults=None,
                 layers_per_block=2, emb_channels=None, noise_emb_dims=None, attn_resolutions=None,
                 midblock_attention=True, concat_balance=0.3, logvar_channels=128, block_kwargs=None,
                 conditional_inputs=(), encode_only=False, disable_out_gain=False, fourier_scale=1, n_logvar=1,
                 *, dtype="bf16", device="cuda"):
        if block_kwargs or encode_only or disable_out_gain:
            raise NotImplementedError("block_kwargs / encode_only / disable_out_gain are not on the accelerated path")
        if fourier_scale != "pos":
            raise NotImplementedError("only fourier_scale='pos' (the released configs) is supported")
        if noise_emb_dims is not None and int(noise_emb_dims) == 0:
            # edm_unet.py:49,87-90: 0 DISABLES the noise input; the engine's embedding kernel always has one (None = model_channels)
            raise NotImplementedError("noise_emb_dims=0 (noise input disabled) is not on the accelerated path")
        mults = list(model_channel_mults or [1, 2, 3, 4])
        lpb = [layers_per_block] * len(mults) if isinstance(layers_per_block, int) else list(layers_per_block)
        conds = [list(c) for c in conditional_inputs]
        if len(conds) > 8 or any(c[0] not in ("tensor", "float") for c in conds):
            raise NotImplementedError("conditional inputs: up to 8 of type 'tensor' / 'float' ('embedding' tables are not on the accelerated path)")
        self.config = dict(image_size=image_size, in_channels=in_channels, out_channels=out_channels or in_channels,
                           model_channels=model_channels, model_channel_mults=mults, layers_per_block=lpb,
                           emb_channels=emb_channels, noise_emb_dims=noise_emb_dims, attn_resolutions=list(attn_resolutions or []),
                           midblock_attention=midblock_attention, concat_balance=concat_balance, conditional_inputs=conds,
                           fourier_scale=fourier_scale)
        self.dtype = dtype
        self.engine = get_engine(device)
        self.device = torch.device("cuda", self.engine.device_id)
        cfg = UnetConfig()
        cfg.image_size, cfg.in_channels, cfg.out_channels = image_size, in_channels, out_channels or in_channels
        cfg.model_channels, cfg.n_levels = model_channels, len(mults)
        for i, m in enumerate(mults):
            cfg.channel_mults[i] = m
            cfg.layers_per_block[i] = lpb[i]
        ar = list(attn_resolutions or [])
        cfg.n_attn_resolutions = len(ar)
        for i, r in enumerate(ar):
            cfg.attn_resolutions[i] = r
        cfg.midblock_attention = int(bool(midblock_attention))
        cfg.concat_balance = float(concat_balance)
        cfg.noise_emb_dims = int(noise_emb_dims or 0)
        cfg.emb_channels = int(emb_channels or 0)
        cfg.n_cond = len(conds)
        for i, (typ, dim, wt) in enumerate(conds):
            cfg.cond_type[i] = 0 if typ == "tensor" else 1
            cfg.cond_dims[i] = int(dim)
            cfg.cond_weights[i] = float(wt)
        self._h = C.c_void_p()
        check(lib().td_unet_create(self.engine._h, C.byref(cfg), DTYPES[dtype], C.byref(self._h)))
        self._finalized = False

    # ---- checkpoint interface
    def expected_parameters(self):
        out = {}
        for i in range(lib().td_unet_num_params(self._h)):
            name, ndim, shape = C.c_char_p(), C.c_int32(), (C.c_int64 * 4)()
            check(lib().td_unet_param_info(self._h, i, C.byref(name), C.byref(ndim), C.byref(shape)))
            out[name.value.decode()] = tuple(shape[k] for k in range(ndim.value))
        return out

    @staticmethod
    def _fold_reference(w, gain):
        """MPConv.forward's weight arithmetic in eval mode (mp_layers.py:9-12, 203-213), in fp32 torch ops exactly as the
        reference executes them, so the engine multiplies by the same numbers the reference does."""
        w = w.to(torch.float32)
        norm = torch.linalg.vector_norm(w, dim=None, keepdim=True)
        norm = torch.add(1e-4, norm, alpha=np.sqrt(norm.numel() / w.numel()))
        w = w / norm
        return w * (gain / np.sqrt(w[0].numel()))

    def load_state_dict(self, state_dict, strict=True, fold="reference"):
        """Reference parameter names (edm_unet.py state dict).  logvar_* (training only) are ignored.
        fold="reference": weights are normalised on the host with the reference's fp32 arithmetic (bit-identical weights);
        fold="engine": raw weights are handed over and folded in fp64 inside the engine (closer to exact maths; differs from
        the reference by the rounding of torch's fp32 vector_norm, ~3e-6 per large conv)."""
        exp = self.expected_parameters()
        missing = [k for k in exp if k not in state_dict]
        unexpected = [k for k in state_dict if k not in exp and not k.startswith("logvar_")]
        if strict and (missing or unexpected):
            raise KeyError(f"state dict mismatch: missing {missing[:5]} unexpected {unexpected[:5]}")
        if missing:
            # strict=False tolerates UNEXPECTED keys (as torch does); the engine has no initialiser for a parameter the checkpoint lacks
            raise ValueError(f"state dict lacks {len(missing)} parameter(s) the model needs, e.g. {missing[:5]}: the engine cannot run on uninitialised weights")
        check(lib().td_unet_set_prefolded(self._h, int(fold == "reference")))
        for k, shape in exp.items():
            w = torch.as_tensor(state_dict[k]).detach().to("cpu", torch.float32).contiguous()
            if tuple(w.shape) != tuple(shape):
                raise ValueError(f"{k}: shape {tuple(w.shape)} != {shape}")
            if fold == "reference" and k.endswith(".weight"):
                if k.endswith(".emb_linear.weight"):
                    gain = torch.as_tensor(state_dict[k[:-len("emb_linear.weight")] + "emb_gain"]).to(torch.float32)
                elif k == "out_conv.weight":
                    gain = torch.as_tensor(state_dict["out_gain"]).to(torch.float32)
                else:
                    gain = 1
                w = self._fold_reference(w, gain).contiguous()
            check(lib().td_unet_set_param(self._h, k.encode(), C.c_void_p(w.data_ptr()), w.numel()))
        check(lib().td_unet_finalize(self._h))
        self._finalized = True
        # the caller's (raw, un-folded) tensors are kept by reference for state_dict() / save_pretrained(): the engine only holds packed weights
        self._state = {k: state_dict[k] for k in state_dict if k in exp or k.startswith("logvar_")}
        return self

    def state_dict(self):
        """The parameters this model was loaded from, under the reference's names (raw values, as a checkpoint holds them)."""
        if not self._finalized:
            raise RuntimeError("load_state_dict first")
        return dict(self._state)

    def save_pretrained(self, save_directory, **_ignored):
        """diffusers ModelMixin layout (what world_pipeline.py:500-518 writes per sub-model and from_pretrained reads back):
        <dir>/config.json with the constructor arguments + <dir>/diffusion_pytorch_model.safetensors."""
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(dict(self.config, _class_name="EDMUnet2D"), f, indent=2, sort_keys=True)
        save_file({k: torch.as_tensor(v).detach().to("cpu", torch.float32).contiguous() for k, v in self.state_dict().items()},
                  os.path.join(save_directory, "diffusion_pytorch_model.safetensors"))

    @classmethod
    def from_pretrained(cls, path, *, dtype="bf16", device="cuda"):
        """diffusers ModelMixin layout: <path>/config.json + one *.safetensors (world_pipeline.py:541-565)."""
        from safetensors.torch import load_file
        with open(os.path.join(path, "config.json")) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        files = [f for f in os.listdir(path) if f.endswith(".safetensors")]
        if len(files) != 1:
            raise FileNotFoundError(f"expected exactly one .safetensors in {path}")
        m = cls(**cfg, dtype=dtype, device=device)
        return m.load_state_dict(load_file(os.path.join(path, files[0])))

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    # ---- model(x, noise_labels, conditional_inputs)  (edm_unet.py:161-184)
    def __call__(self, x, noise_labels, conditional_inputs=None):
        if not self._finalized:
            raise RuntimeError("load_state_dict first")
        x = f32(x)
        n, _, H, W = x.shape
        t = f32(noise_labels, "cpu").flatten()
        if t.numel() == 1 and n > 1:
            t = t.expand(n).contiguous()
        if t.numel() != n:
            raise ValueError(f"noise_labels has {t.numel()} entries for a batch of {n} (expected 1 or {n})")
        if x.shape[1] != self.config["in_channels"]:
            raise ValueError(f"x has {x.shape[1]} channels, the model takes {self.config['in_channels']}")
        cond = self.cond_rows(conditional_inputs, n, x.device)
        out = torch.empty((n, self.config["out_channels"], H, W), dtype=torch.float32, device=x.device)
        check(lib().td_unet_forward(self._h, n, H, W, ptr(x), ptr(t), ptr(cond), ptr(out)))
        return out

    forward = __call__

    def cond_rows(self, conditional_inputs, n, device=None):
        """conditional_inputs (list, reference order: (n,dim) tensors for 'tensor' inputs, (n,) or (1,) for 'float' inputs)
        -> one contiguous (n, row_len) fp32 conditioning matrix as the C-ABI takes it, or None."""
        conds = self.config["conditional_inputs"]
        conditional_inputs = list(conditional_inputs or [])
        if len(conditional_inputs) != len(conds):
            raise ValueError("Invalid number of conditional inputs")
        if not conds:
            return None
        cols = []
        for (typ, dim, _w), v in zip(conds, conditional_inputs):
            v = torch.as_tensor(v, dtype=torch.float32)
            if typ == "tensor":
                v = v.reshape(-1, dim)
            else:
                v = v.reshape(-1, 1)
            cols.append(v.expand(n, -1) if v.shape[0] == 1 else v)
        out = torch.cat([c.to(cols[0].device) for c in cols], dim=1).contiguous()
        return out.to(device) if device is not None else out

    def read_activation(self, n, H, W, label, max_elems=1 << 26):
        """Debug/test: output of fused conv op `label` from the last forward with this (n,H,W), as NCHW fp32 (host)."""
        buf = np.empty(max_elems, dtype=np.float32)
        dims = (C.c_int32 * 4)()
        check(lib().td_unet_read_activation(self._h, n, H, W, label.encode(), C.c_void_p(buf.ctypes.data), buf.size, C.byref(dims)))
        shape = tuple(dims)
        return torch.from_numpy(buf[:int(np.prod(shape))].reshape(shape).copy())

    def close(self):
        if self._h:
            lib().td_unet_destroy(self._h)
            self._h = C.c_void_p()
