import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from oracle import pixrefer_ref as ref
from voicepuppet_amd.engine import PixReferEngine
import gpu_util as gu
from test_gpu_step import synth, make_params

dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
ngf = ndf = 8; n, h = 2, 256
p = make_params(ngf, ndf, 3); batch = synth(n, h, 11)
p64 = {k: v.astype(np.float64) for k, v in p.items()}
nodes = ref.forward_backward(p64, *[b.astype(np.float64) for b in batch], ngf=ngf, ndf=ndf)
eng = PixReferEngine(n, h, ngf, ndf, dtype=dtype, training=True)
eng.load_params(p)
eng.forward(*[torch.tensor(b, device='cuda') for b in batch]); eng.backward(); torch.cuda.synchronize()
T = lambda name: eng.tensor(name).float().cpu().numpy()
print('losses', eng.losses(), {k: nodes[k] for k in ('Discrim_loss','Gen_loss_GAN','Gen_loss_L1','Gen_loss','Perceptual_loss')})
dofg = T('d_din')[..., 3:6] + T('d_vin')[..., 0:3]
print('d_din part', gu.rel_l2(T('d_din')[..., 3:6] + 0*dofg, nodes['d_outputs_fg']), 'norms', np.linalg.norm(T('d_din')[...,3:6]), np.linalg.norm(T('d_vin')[...,0:3]))
print('d_din vs oracle', gu.rel_l2(T('d_din')[..., :6], nodes['d_dinput']), 'd_vin vs oracle', gu.rel_l2(T('d_vin')[..., :3], nodes['d_vin']))
for nm in ['conv3/conv3_3','conv3/conv3_2','conv3/conv3_1','pool2','conv2/conv2_2','conv2/conv2_1','pool1','conv1/conv1_2','conv1/conv1_1']:
  print(nm, 'dy norm', np.linalg.norm(T('v/'+nm+':dy')))
print('d_outputs_fg', gu.rel_l2(dofg, nodes['d_outputs_fg']))
o4 = nodes['gen_out4']
print('d_gen_out4(pre-tanh)', gu.rel_l2(T('d_gen_out4')[..., :4], nodes['d_gen_out4'] * (1 - o4 ** 2)))
acts, dacts = nodes['g_acts'], nodes['g_dacts']
for scope, kind, srcs, cout, bn, pre in ref.generator_spec(ngf):
  if scope == 'decoder_1': continue
  y = T('g/' + scope)
  msg = '%-20s' % scope
  if bn:
    sc, sh = T('g/%s:scale' % scope).reshape(-1), T('g/%s:shift' % scope).reshape(-1)
    z = sc * y + sh
  else:
    z = y
  msg += ' act %.2e' % gu.rel_l2(z, acts[scope])
  print(msg)
worst = {}
for which, key in ((1, 'Discrim_grads'), (0, 'Gen_grads')):
  grads = eng.get_params(which, src=eng.grads_d if which == 1 else eng.grads_g)
  for name, g in grads.items():
    r = nodes[key][name]
    if np.all(r == 0): continue
    print('%-60s %.3e' % (name, gu.rel_l2(g, r)))
