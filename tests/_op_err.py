import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import terrain_diffusion_amd as td
from conftest import rel_rms
from oracle import rng
from oracle.unet import OracleUnet, synth_state_dict, BASE_CONFIG, normalize, mp_silu, mp_sum2
cfg = dict(BASE_CONFIG); sd = synth_state_dict(cfg, seed=1234); shape = (1, 5, 64, 64); t = torch.tensor([1.1])
x = torch.from_numpy(rng.standard_normal(7, shape)); cond = torch.from_numpy(rng.standard_normal(8, (1, 58)))
o64 = OracleUnet(cfg, sd, dtype=torch.float64)
m = td.EDMUnet2D(**cfg, dtype='fp32').load_state_dict(sd)
y = m(x.cuda(), t, [cond.cuda()]).cpu()
R = lambda lab: m.read_activation(1, 64, 64, lab).double()
cv = R('@cvec').reshape(1, -1)
x0 = R('enc.512x512_conv')
xin = torch.cat([x.double(), torch.ones(1, 1, 64, 64, dtype=torch.float64)], 1)
print('first conv (from exact input)      %.3e' % rel_rms(x0, F.conv2d(xin, o64.w['enc.512x512_conv'], padding=1)))
n = 'enc.512x512_block0'
c = cv[:, 0:192]
xn = normalize(x0, dim=1)
y1_exp = mp_silu(F.conv2d(mp_silu(xn), o64.w[n + '.conv_res0'], padding=1) * c[:, :, None, None])
y1 = R(n + '.conv_res0')
print('block0 conv_res0 (own inputs)      %.3e' % rel_rms(y1, y1_exp))
e = (y1 - y1_exp)
print('   err by rows (first 4 / middle / last):', [float(e[0, :, r].pow(2).mean().sqrt()) for r in (0, 1, 2, 3, 31, 32, 62, 63)])
print('   err by channel blocks of 32:', [float(e[0, cb:cb + 32].pow(2).mean().sqrt()) for cb in range(0, 192, 32)])
out_exp = torch.clip(mp_sum2(xn, F.conv2d(y1, o64.w[n + '.conv_res1'], padding=1), 0.3), -256, 256)
print('block0 conv_res1+res (own inputs)  %.3e' % rel_rms(R(n + '.conv_res1'), out_exp))
# pixel norm alone: compare rn implied
print('xn magnitude', float(xn.pow(2).mean().sqrt()))

ss = R('sumsq:enc.512x512_conv')
print('sumsq parts shape', tuple(ss.shape))
tot = ss.sum(0)[0]
ref = x0.pow(2).sum(1)[0]
print('sumsq rel err %.3e' % rel_rms(tot, ref), ' max rel %.3e' % float(((tot-ref).abs()/ref).max()))
per = torch.stack([x0[0, 32*i:32*i+32].pow(2).sum(0) for i in range(6)])
print('per-part rel err', [float(rel_rms(ss[i,0], per[i])) for i in range(6)])
out = R(n + '.conv_res1'); e = (out - out_exp)[0]; xn0 = xn[0]
alpha = (e * xn0).sum(0) / xn0.pow(2).sum(0)
resid = e - alpha[None] * xn0
print('res1: err rms %.3e ; per-pixel alpha mean %.3e std %.3e ; residual rms after removing alpha*xn %.3e' % (float(e.pow(2).mean().sqrt()), float(alpha.mean()), float(alpha.std()), float(resid.pow(2).mean().sqrt())))
# per-channel coherent component
beta = (e * xn0).sum((1, 2)) / xn0.pow(2).sum((1, 2))
print('per-channel beta mean %.3e std %.3e' % (float(beta.mean()), float(beta.std())))
xs = R('enc.256x256_block0.conv_skip'); xd = R('enc.256x256_down.conv_res1')
print('conv_skip 1x1 (own inputs) %.3e' % rel_rms(xs, F.conv2d(xd, o64.w['enc.256x256_block0.conv_skip'])))
y1m = R(n + '.conv_res0')
conv_exp = F.conv2d(y1m, o64.w[n + '.conv_res1'], padding=1)          # fp64 exact
conv_f32 = F.conv2d(y1m.float(), o64.w[n + '.conv_res1'].float(), padding=1).double()
conv_hip = (out - mp_sum2(xn, torch.zeros_like(xn), 0.3) ) / (0.3 / 0.58 ** 0.5)   # subtract residual part (fp64 xn)
ee = (conv_hip - conv_exp)[0]
print('conv-only: hip rel %.3e  cpu-fp32 rel %.3e' % (rel_rms(conv_hip, conv_exp), rel_rms(conv_f32, conv_exp)))
q = torch.quantile(ee.abs().flatten()[:1000000], torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], dtype=torch.float64))
print('abs err quantiles 50/90/99/99.9/max:', [float(v) for v in q], ' rms', float(ee.pow(2).mean().sqrt()))
print('mean signed err %.3e ; corr(err, value) %.3e' % (float(ee.mean()), float((ee * conv_exp[0]).mean() / conv_exp[0].pow(2).mean())))
