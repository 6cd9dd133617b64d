"""sha256 of the base U-Net's output on fixed inputs (batch 64 / 5 / 1, bf16): compares two builds of the library bit for bit."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import terrain_diffusion_amd as td
from terrain_diffusion_amd._lib import lib
from oracle.unet import BASE_CONFIG, synth_state_dict
from oracle import rng
for kv in os.environ.get("TD_OPTS", "").split(","):
    if kv:
        k, v = kv.split("="); td.engine.get_engine("cuda").set_option(k, int(v))
cfg = dict(BASE_CONFIG)
m = td.EDMUnet2D(**cfg, dtype="bf16").load_state_dict(synth_state_dict(cfg, seed=1234))
hs = []
for n in (64, 5, 1):
    x = torch.from_numpy(rng.standard_normal(7, (n, 5, 64, 64))).cuda()
    c = torch.from_numpy(rng.standard_normal(8, (n, 58))).cuda()
    y = m(x, torch.full((n,), 1.1), [c])
    hs.append(hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16])
lib().td_build_id.restype = __import__("ctypes").c_char_p
print("build", lib().td_build_id().decode(), "output hashes", hs)
