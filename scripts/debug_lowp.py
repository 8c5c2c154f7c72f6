import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from oracle import pixrefer_lowp_ref as lowp
from voicepuppet_amd.engine import PixReferEngine
import gpu_util as gu
from test_gpu_step import synth, make_params
ngf = ndf = 8; n, h = 2, 256
p = make_params(ngf, ndf, 3); batch = synth(n, h, 11)
nodes = lowp.forward_backward({k: v.astype(np.float64) for k, v in p.items()}, *[b.astype(np.float64) for b in batch], ngf=ngf, ndf=ndf)
eng = PixReferEngine(n, h, ngf, ndf, dtype='bf16', training=True); eng.load_params(p)
eng.forward(*[torch.tensor(b, device='cuda') for b in batch]); eng.backward_d(); torch.cuda.synchronize()
T = lambda name: eng.tensor(name).float().cpu().numpy()
D = nodes['D']
for sc in ['layer_1', 'layer_2', 'layer_3', 'layer_4']:
  y = T('d/' + sc); r = D.y[sc]
  print(sc, 'y relL2 %.2e  exact-match frac %.4f' % (gu.rel_l2(y, r), np.mean(y == r)))
print('logits', gu.rel_l2(T('logits'), D.y['layer_5']))
for sc in ['layer_4', 'layer_3', 'layer_2', 'layer_1']:
  dy = T('d/%s:dy' % sc); r = nodes['d_dy_dloss'][sc]
  print(sc, 'dy(D loss) relL2 %.2e exact %.4f' % (gu.rel_l2(dy, r), np.mean(dy == r)))
y = T('d/layer_1'); r = D.y['layer_1']
for g in range(3): print('layer_1 group', g, 'exact frac %.4f relL2 %.2e' % (np.mean(y[g*n:(g+1)*n] == r[g*n:(g+1)*n]), gu.rel_l2(y[g*n:(g+1)*n], r[g*n:(g+1)*n])))
din = T('d/d_inputs'); rd = D.y['d_inputs']
for g in range(3): print('d_inputs group', g, 'exact frac %.4f' % np.mean(din[g*n:(g+1)*n, ..., :6] == rd[g*n:(g+1)*n]))
G = nodes['G']
for sc in ['encoder_1', 'encoder_2', 'encoder_fg_1']:
  y = T('g/' + sc); r = G.y[sc]
  print(sc, 'y relL2 %.2e exact %.4f' % (gu.rel_l2(y, r), np.mean(y == r)))
gin = T('g/inputs'); print('gin exact', np.mean(gin[..., :6] == G.y['inputs']))
# direct check of one conv: recompute layer_1 group 0 on host from the device's own stored inputs and packed weights
from oracle import nn_ops as ops
w = lowp.round_bf16(p['discriminator/layer_1/conv2d/kernel']); b = p['discriminator/layer_1/conv2d/bias'].astype(np.float64)
yy = ops.conv2d_fwd(din[:n, ..., :6].astype(np.float64), w, b, 2, 1)
print('host conv from device inputs: pre-round vs device y: max abs diff / ulp', np.abs(lowp.round_bf16(yy) - T('d/layer_1')[:n]).max())
dd = np.abs(yy - T('d/layer_1')[:n]); print('|pre-round host - device stored| max %.3e mean %.3e; typical |y| %.3e' % (dd.max(), dd.mean(), np.abs(yy).mean()))
