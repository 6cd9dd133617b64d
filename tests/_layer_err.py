import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import terrain_diffusion_amd as td
from conftest import rel_rms
from oracle import rng
from oracle.unet import OracleUnet, synth_state_dict, tiny_config, BASE_CONFIG
which = sys.argv[1] if len(sys.argv) > 1 else 'tiny'
if which == 'tiny':
    cfg = tiny_config(64, 1); seed = 77; shape = (2, 5, 16, 16); t = torch.tensor([1.2, 0.3])
else:
    cfg = dict(BASE_CONFIG); seed = 1234; shape = (1, 5, 64, 64); t = torch.tensor([1.1])
sd = synth_state_dict(cfg, seed=seed)
x = torch.from_numpy(rng.standard_normal(7, shape)); cond = torch.from_numpy(rng.standard_normal(8, (shape[0], 58)))
taps = {}
o32 = OracleUnet(cfg, sd); y32 = o32(x, t, [cond], taps=taps)
taps64 = {}
o64 = OracleUnet(cfg, sd, dtype=torch.float64); y64 = o64(x.double(), t, [cond.double()], taps=taps64)
m = td.EDMUnet2D(**cfg, dtype='fp32').load_state_dict(sd)
import os
for kv in os.environ.get('TD_OPTS','').split(','):
    if kv: k,v=kv.split('='); m.engine.set_option(k,int(v)); print('opt',k,v)
y = m(x.cuda(), t, [cond.cuda()]).cpu()
print('final: hip-vs-o64 %.3e  o32-vs-o64 %.3e  hip-vs-o32 %.3e' % (rel_rms(y, y64), rel_rms(y32, y64), rel_rms(y, y32)))
import torch.nn.functional as F
emb_h = m.read_activation(shape[0], shape[2], shape[3], '@emb').reshape(shape[0], -1)
e64 = o64.embeddings(t, [cond.double()]); e32 = o32.embeddings(t, [cond])
print('emb: hip-vs-o64 %.3e o32-vs-o64 %.3e' % (rel_rms(emb_h, e64), rel_rms(e32, e64)))
cv = m.read_activation(shape[0], shape[2], shape[3], '@cvec').reshape(shape[0], -1)
off = 0
for b in o32.plan['enc'] + o32.plan['dec']:
    if b['kind'] == 'conv': continue
    c64 = F.linear(e64, o64.w[b['name'] + '.emb_linear']) + 1; c64 = c64 / torch.sqrt(torch.mean(c64 ** 2, dim=1, keepdim=True) + 1e-8)
    print('c %-28s %.3e' % (b['name'], rel_rms(cv[:, off:off + b['cout']], c64))); off += b['cout']
    if off > 600: break
for b in o32.plan['enc'] + o32.plan['dec']:
    n = b['name']
    lab = n if b['kind'] == 'conv' else (n + ('.attn_proj' if b['attn'] else '.conv_res1'))
    a = m.read_activation(shape[0], shape[2], shape[3], lab)
    line = '%-28s hip-vs-o64 %.3e  o32-vs-o64 %.3e' % (n, rel_rms(a, taps64[n]), rel_rms(taps[n], taps64[n]))
    if b['kind'] != 'conv':
        a1 = m.read_activation(shape[0], shape[2], shape[3], n + '.conv_res0')
        line += '   y1: hip %.3e o32 %.3e' % (rel_rms(a1, taps64[n + '.y1']), rel_rms(taps[n + '.y1'], taps64[n + '.y1']))
    print(line)
