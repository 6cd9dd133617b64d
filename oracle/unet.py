"""ORACLE: CPU restatement of EDMUnet2D (magnitude-preserving U-Net) in plain torch fp32/fp64.

Follows (behaviour, re-written): terrain_diffusion/models/edm_unet.py:17-184,
terrain_diffusion/models/unet_block.py:12-156, terrain_diffusion/models/mp_layers.py:9-221.
Functional form: a config dict + a flat state dict (reference parameter names) -> forward.
Weights are folded once (mp_layers.py:203-213) instead of on every call.
"""
from collections import OrderedDict
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import rng

COARSE_CONFIG = dict(image_size=16, in_channels=11, out_channels=6, model_channels=128, model_channel_mults=[1], layers_per_block=2,
                     attn_resolutions=[], midblock_attention=False, concat_balance=0.5, conditional_inputs=[["float", 64, 0.2]] * 5,
                     fourier_scale="pos")   # configs/diffusion_coarse/diffusion_coarse_30m.cfg:47-62
DECODER_CONFIG = dict(image_size=512, in_channels=5, out_channels=1, model_channels=64, model_channel_mults=[1, 2, 3, 4], layers_per_block=3,
                      attn_resolutions=[], midblock_attention=False, concat_balance=0.5, conditional_inputs=[],
                      fourier_scale="pos")  # configs/diffusion_decoder/diffusion_decoder_64-3.cfg:51-65

BASE_CONFIG = dict(image_size=512, in_channels=5, out_channels=5, model_channels=192,
                   model_channel_mults=[1, 2, 3, 4], layers_per_block=3, attn_resolutions=[8, 16],
                   midblock_attention=True, concat_balance=0.5, conditional_inputs=[["tensor", 58, 1.0]],
                   fourier_scale="pos")


def tiny_config(model_channels=64, layers_per_block=1, image_size=512, **kw):
    cfg = dict(BASE_CONFIG)
    cfg.update(model_channels=model_channels, layers_per_block=layers_per_block, image_size=image_size)
    cfg.update(kw)
    return cfg


# ------------------------------------------------------------------------------ architecture plan
def build_plan(cfg):
    """Block list in execution order, mirroring edm_unet.py:105-139 (names are the reference's)."""
    mc = cfg["model_channels"]
    mults = cfg.get("model_channel_mults") or [1, 2, 3, 4]
    lpb = cfg.get("layers_per_block", 2)
    lpb = [lpb] * len(mults) if isinstance(lpb, int) else list(lpb)
    attn_res = cfg.get("attn_resolutions") or []
    emb_ch = cfg.get("emb_channels") or mc * max(mults)
    image_size = cfg["image_size"]
    chans = [mc * m for m in mults]
    enc, dec = [], []
    cout = cfg["in_channels"] + 1
    for level, (ch, nb) in enumerate(zip(chans, lpb)):
        res = image_size // 2 ** level
        if level == 0:
            enc.append(dict(name=f"enc.{res}x{res}_conv", kind="conv", cin=cout, cout=ch))
            cout = ch
        else:
            enc.append(dict(name=f"enc.{res}x{res}_down", kind="block", mode="enc", cin=cout, cout=cout,
                            resample="down", attn=False))
        for idx in range(nb):
            cin, cout = cout, ch
            enc.append(dict(name=f"enc.{res}x{res}_block{idx}", kind="block", mode="enc", cin=cin, cout=cout,
                            resample="keep", attn=(res in attn_res)))
    skips = [b["cout"] for b in enc]
    for level, (ch, nb) in reversed(list(enumerate(zip(chans, lpb)))):
        res = image_size // 2 ** level
        if level == len(chans) - 1:
            dec.append(dict(name=f"dec.{res}x{res}_in0", kind="block", mode="dec", cin=cout, cout=cout,
                            resample="keep", attn=bool(cfg.get("midblock_attention", True)), concat=False))
            dec.append(dict(name=f"dec.{res}x{res}_in1", kind="block", mode="dec", cin=cout, cout=cout,
                            resample="keep", attn=False, concat=False))
        else:
            dec.append(dict(name=f"dec.{res}x{res}_up", kind="block", mode="dec", cin=cout, cout=cout,
                            resample="up", attn=False, concat=False))
        for idx in range(nb + 1):
            skip = skips.pop()
            cin, cout = cout + skip, ch
            dec.append(dict(name=f"dec.{res}x{res}_block{idx}", kind="block", mode="dec", cin=cin, cout=cout,
                            resample="keep", attn=(res in attn_res), concat=True, skip_c=skip))
    return dict(enc=enc, dec=dec, emb_channels=emb_ch, final_c=cout)


def param_shapes(cfg):
    """Ordered {name: shape} of every tensor forward() reads (reference state-dict names)."""
    plan = build_plan(cfg)
    mc = cfg["model_channels"]
    emb = plan["emb_channels"]
    noise_dims = mc if cfg.get("noise_emb_dims") is None else cfg["noise_emb_dims"]
    sh = OrderedDict()
    sh["out_gain"] = ()
    sh["noise_linear.weight"] = (emb, noise_dims)
    for i, (typ, x, _w) in enumerate(cfg.get("conditional_inputs", [])):
        if typ == "tensor":
            sh[f"conditional_layers.{i}.weight"] = (emb, x)
        elif typ == "float":
            sh[f"conditional_layers.{i}.0.freqs"] = (x,)
            sh[f"conditional_layers.{i}.0.phases"] = (x,)
            sh[f"conditional_layers.{i}.1.weight"] = (emb, x)
        else:
            raise NotImplementedError(typ)
    for b in plan["enc"] + plan["dec"]:
        n = b["name"]
        if b["kind"] == "conv":
            sh[n + ".weight"] = (b["cout"], b["cin"], 3, 3)
            continue
        sh[n + ".emb_gain"] = ()
        c0_in = b["cout"] if b["mode"] == "enc" else b["cin"]
        sh[n + ".conv_res0.weight"] = (b["cout"], c0_in, 3, 3)
        sh[n + ".emb_linear.weight"] = (b["cout"], emb)
        sh[n + ".conv_res1.weight"] = (b["cout"], b["cout"], 3, 3)
        if b["cin"] != b["cout"]:
            sh[n + ".conv_skip.weight"] = (b["cout"], b["cin"], 1, 1)
        if b["attn"]:
            sh[n + ".attn_qkv.weight"] = (3 * b["cout"], b["cout"], 1, 1)
            sh[n + ".attn_proj.weight"] = (b["cout"], b["cout"], 1, 1)
    sh["out_conv.weight"] = (cfg.get("out_channels") or cfg["in_channels"], plan["final_c"], 3, 3)
    return sh


def _name_seed(name: str, seed: int) -> int:
    h = 0xCBF29CE484222325  # FNV-1a 64
    for ch in name.encode():
        h = ((h ^ ch) * 0x100000001B3) & rng.M64
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & rng.M64 or 1


def synth_state_dict(cfg, seed=1234, out_gain=1.0, emb_gain=0.5):
    """Deterministic synthetic weights from the PORTABLE rng (no torch RNG dependence): weight tensors are
    standard normals seeded by FNV(name)^seed; gains are set non-zero (fresh gains are 0 and
    would make every parity check vacuous).  Also emits the positional-embedding `freqs` buffer exactly
    as mp_layers.py:89-94 builds it."""
    sd = OrderedDict()
    for name, shape in param_shapes(cfg).items():
        if name == "out_gain":
            sd[name] = torch.tensor(float(out_gain))
        elif name.endswith("emb_gain"):
            sd[name] = torch.tensor(float(emb_gain))
        elif name.endswith(".freqs"):
            sd[name] = torch.from_numpy(rng.standard_normal(_name_seed(name, seed), shape)) * (2 * np.pi)
        elif name.endswith(".phases"):
            u = rng.pcg_stream(_name_seed(name, seed), int(np.prod(shape))).astype(np.float64) / 4294967296.0
            sd[name] = torch.from_numpy((2 * np.pi * u).astype(np.float32)).reshape(shape)
        else:
            sd[name] = torch.from_numpy(rng.standard_normal(_name_seed(name, seed), shape))
    if cfg.get("fourier_scale", 1) == "pos":
        mc = cfg["model_channels"]
        noise_dims = mc if cfg.get("noise_emb_dims") is None else cfg["noise_emb_dims"]
        half = noise_dims // 2
        sd["noise_fourier.freqs"] = torch.exp(torch.arange(half) * -(math.log(10) / (half - 1)))
    else:
        raise NotImplementedError("only the positional noise embedding is on the hot path")
    return sd


# ------------------------------------------------------------------------------ numerics
def normalize(x, dim=None, eps=1e-4):
    """mp_layers.py:9-12."""
    norm = torch.linalg.vector_norm(x, dim=dim, keepdim=True)
    norm = eps + norm * math.sqrt(norm.numel() / x.numel())
    return x / norm


def fold_weight(w, gain=1.0):
    """mp_layers.py:203-213 (eval mode): W/(eps+||W||/sqrt(numel)) * gain/sqrt(fan_in)."""
    w = w.to(torch.float32)
    w = normalize(w)
    return w * (float(gain) / math.sqrt(w[0].numel()))


def mp_silu(x):
    return F.silu(x) / 0.596


def mp_sum2(a, b, t):
    """mp_layers.py:47-62 with w = [1-t, t]."""
    wa, wb = 1.0 - t, t
    return (a * wa + b * wb) / math.sqrt(wa * wa + wb * wb)


def mp_concat_scales(na, nb, t):
    """mp_layers.py:65-86 with w = [1-t, t]: per-source scalars."""
    wa, wb = 1.0 - t, t
    c = math.sqrt((na + nb) / (wa * wa + wb * wb))
    return c / math.sqrt(na) * wa, c / math.sqrt(nb) * wb


def pos_embedding(t, freqs):
    """mp_layers.py:96-107."""
    y = t.to(torch.float32).outer(freqs.to(torch.float32))
    return torch.cat([torch.sin(y), torch.cos(y)], dim=1) * math.sqrt(2)


class OracleUnet:
    """Folded-weight functional EDMUnet2D.  dtype float32 (parity) or float64 (error-floor studies)."""

    def __init__(self, cfg, state_dict, dtype=torch.float32):
        self.cfg = cfg
        self.plan = build_plan(cfg)
        self.dtype = dtype
        sd = state_dict
        f = lambda name, gain=1.0: fold_weight(sd[name], gain).to(dtype)
        self.w = {}
        self.w["noise_linear"] = f("noise_linear.weight")
        self.freqs = sd["noise_fourier.freqs"].to(torch.float32)
        self.cond_w = []       # (type, folded weight, freqs, phases)
        self.cond_weights = [1.0]
        for i, (typ, x, wt) in enumerate(cfg.get("conditional_inputs", [])):
            if typ == "tensor":
                self.cond_w.append(("tensor", f(f"conditional_layers.{i}.weight"), None, None))
            elif typ == "float":   # nn.Sequential(MPFourier(x), MPConv(x, emb)) — edm_unet.py:96-97, mp_layers.py:109-131
                self.cond_w.append(("float", f(f"conditional_layers.{i}.1.weight"), sd[f"conditional_layers.{i}.0.freqs"].to(torch.float32),
                                    sd[f"conditional_layers.{i}.0.phases"].to(torch.float32)))
            else:
                raise NotImplementedError(typ)
            self.cond_weights.append(float(wt))
        for b in self.plan["enc"] + self.plan["dec"]:
            n = b["name"]
            if b["kind"] == "conv":
                self.w[n] = f(n + ".weight")
                continue
            self.w[n + ".conv_res0"] = f(n + ".conv_res0.weight")
            self.w[n + ".emb_linear"] = f(n + ".emb_linear.weight", float(sd[n + ".emb_gain"]))
            self.w[n + ".conv_res1"] = f(n + ".conv_res1.weight")
            if b["cin"] != b["cout"]:
                self.w[n + ".conv_skip"] = f(n + ".conv_skip.weight")
            if b["attn"]:
                self.w[n + ".attn_qkv"] = f(n + ".attn_qkv.weight")
                self.w[n + ".attn_proj"] = f(n + ".attn_proj.weight")
        self.w["out_conv"] = f("out_conv.weight", float(sd["out_gain"]))

    # edm_unet.py:145-159
    def embeddings(self, noise_labels, conditional_inputs):
        embeds = [F.linear(pos_embedding(noise_labels, self.freqs).to(self.dtype), self.w["noise_linear"])]
        for (typ, w, fr, ph), c in zip(self.cond_w, conditional_inputs):
            if typ == "tensor":
                embeds.append(mp_silu(F.linear(c.to(self.dtype), w)))          # MPConv inputs get mp_silu (edm_unet.py:152-153)
            else:
                y = c.to(torch.float32).outer(fr) + ph                          # MPFourier (mp_layers.py:115-131), no mp_silu
                embeds.append(F.linear((y.cos() * math.sqrt(2)).to(self.dtype), w))
        ws = torch.tensor(self.cond_weights, dtype=self.dtype)
        emb = sum(e * wi for e, wi in zip(embeds, ws)) / torch.linalg.vector_norm(ws)
        return mp_silu(emb)

    # unet_block.py:102-108
    def _attn(self, x, n):
        heads = x.shape[1] // 64
        y = F.conv2d(x, self.w[n + ".attn_qkv"])
        y = y.reshape(y.shape[0], heads, -1, 3, y.shape[2] * y.shape[3])
        q, k, v = normalize(y, dim=2).unbind(3)
        w = torch.einsum("nhcq,nhck->nhqk", q, k / math.sqrt(q.shape[2])).softmax(dim=3)
        y = torch.einsum("nhqk,nhck->nhcq", w, v)
        return F.conv2d(y.reshape(*x.shape), self.w[n + ".attn_proj"])

    # unet_block.py:116-156
    def _block(self, b, x, emb, taps=None):
        n = b["name"]
        if b["resample"] == "down":
            x = x[:, :, ::2, ::2]
        elif b["resample"] == "up":
            x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
        if b["mode"] == "enc":
            if b["cin"] != b["cout"]:
                x = F.conv2d(x, self.w[n + ".conv_skip"])
            x = normalize(x, dim=1)
        y = F.conv2d(mp_silu(x), self.w[n + ".conv_res0"], padding=1)
        c = F.linear(emb, self.w[n + ".emb_linear"]) + 1
        c = c / torch.sqrt(torch.mean(c ** 2, dim=1, keepdim=True) + 1e-8)
        y = mp_silu(y * c[:, :, None, None])
        if taps is not None:
            taps[n + ".y1"] = y
        y = F.conv2d(y, self.w[n + ".conv_res1"], padding=1)
        if b["mode"] == "dec" and b["cin"] != b["cout"]:
            x = F.conv2d(x, self.w[n + ".conv_skip"])
        x = mp_sum2(x, y, 0.3)
        if b["attn"]:
            x = mp_sum2(x, self._attn(x, n), 0.3)
        return torch.clip(x, -256, 256)

    # edm_unet.py:161-184
    def forward(self, x, noise_labels, conditional_inputs, taps=None):
        x = x.to(self.dtype)
        emb = self.embeddings(noise_labels, conditional_inputs)
        x = torch.cat([x, torch.ones_like(x[:, :1])], dim=1)
        skips = []
        for b in self.plan["enc"]:
            x = F.conv2d(x, self.w[b["name"]], padding=1) if b["kind"] == "conv" else self._block(b, x, emb, taps)
            if taps is not None:
                taps[b["name"]] = x
            skips.append(x)
        t = self.cfg.get("concat_balance", 0.3)
        for b in self.plan["dec"]:
            if b.get("concat"):
                s = skips.pop()
                sa, sb = mp_concat_scales(x.shape[1], s.shape[1], t)
                x = torch.cat([x * sa, s * sb], dim=1)
            x = self._block(b, x, emb, taps)
            if taps is not None:
                taps[b["name"]] = x
        return F.conv2d(x, self.w["out_conv"], padding=1)

    __call__ = forward
