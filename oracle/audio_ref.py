"""Oracle: log-mel front-end + MfccNet + BFMNet head (inference).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (TF1.x cannot run here and the
reference has no vectors for this path); semantics stated from the TF 1.14 API contract and pinned by
identity tests (naive DFT, Parseval, mel-matrix properties, a torch-CPU float64 second opinion).

Restates, in plain numpy:
  generator/generator.py:40-80          DataGenerator.set_params / extract_mfcc  (log-mel, NOT a cepstrum)
  voicepuppet/bfmnet/tinynet.py:7-215   mobilenet_v2_func_blocks / MfccNet (inference: BN = stored affine)
  voicepuppet/bfmnet/bfmnet.py:20-122   MfccEncoder, RNNModule (GRUCell via dynamic_rnn), BFMCoeffDecoder
  voicepuppet/bfmnet/bfmnet.py:189-213  BFMNet.build_network(trainable=False)
"""
import numpy as np


# ----------------------------------------------------------------------------
# log-mel  (generator.py:60-80)
# ----------------------------------------------------------------------------
def hann_periodic(n):
  """tf.signal.hann_window(periodic=True), the default window of tf.signal.stft."""
  return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft_mag(pcm, frame_length=512, frame_step=128, fft_length=512):
  """tf.abs(tf.signal.stft(...)): no centring, no end padding; frames = 1 + (L - frame_length)//frame_step."""
  pcm = np.asarray(pcm)
  L = pcm.shape[-1]
  nf = 1 + (L - frame_length) // frame_step
  idx = np.arange(frame_length)[None, :] + frame_step * np.arange(nf)[:, None]
  frames = pcm[..., idx] * hann_periodic(frame_length)
  return np.abs(np.fft.rfft(frames, n=fft_length, axis=-1))


def _hz_to_mel(f):
  return 1127.0 * np.log1p(np.asarray(f, dtype=np.float64) / 700.0)   # HTK mel (TF _hertz_to_mel)


def linear_to_mel_weight_matrix(num_mel_bins=80, num_spectrogram_bins=257, sample_rate=16000,
                                lower_edge_hertz=80.0, upper_edge_hertz=7600.0):
  """tf.signal.linear_to_mel_weight_matrix: triangles in the mel domain, DC bin zeroed, no normalisation."""
  nyquist = sample_rate / 2.0
  lin = np.linspace(0.0, nyquist, num_spectrogram_bins)[1:]            # bin 0 (DC) is dropped ...
  spec_mel = _hz_to_mel(lin)[:, None]
  edges = np.linspace(_hz_to_mel(lower_edge_hertz), _hz_to_mel(upper_edge_hertz), num_mel_bins + 2)
  lower, center, upper = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
  lower_slopes = (spec_mel - lower) / (center - lower)
  upper_slopes = (upper - spec_mel) / (upper - center)
  w = np.maximum(0.0, np.minimum(lower_slopes, upper_slopes))
  return np.pad(w, ((1, 0), (0, 0)))                                   # ... and re-inserted as a zero row


def extract_mfcc(pcm, sample_rate=16000, num_mel_bins=80, win_length=512, hop_step=128, fft_length=512):
  """[B, L] -> [B, frames, 80] = log(|STFT| . mel + 1e-6).  The mel matrix is float32 in TF."""
  mag = stft_mag(pcm, win_length, hop_step, fft_length)
  mel = linear_to_mel_weight_matrix(num_mel_bins, fft_length // 2 + 1, sample_rate).astype(np.float32).astype(np.float64)
  return np.log(mag @ mel + 1e-6)


def pcm_length_for(pad_len, hop_step=128, win_length=512, frame_mfcc_scale=5):
  """infer_bfmvid.py:164: samples so that exactly pad_len*frame_mfcc_scale STFT frames come out."""
  return hop_step * (pad_len * frame_mfcc_scale - 1) + win_length


# ----------------------------------------------------------------------------
# MfccNet (tinynet.py:159-212), NHWC with H = time, W = mel
# ----------------------------------------------------------------------------
BN_EPS = 1e-3   # tf.contrib.layers.batch_norm default; scale=False (no gamma), center=True


def same_pads(size, k, s):
  """TF 'SAME': total = max((ceil(size/s)-1)*s + k - size, 0); extra goes at the end."""
  out = -(-size // s)
  total = max((out - 1) * s + k - size, 0)
  return total // 2, total - total // 2, out


def conv2d_same(x, w, stride):
  """tf.layers.conv2d(padding='same', use_bias=False); w HWIO; stride (sh, sw)."""
  n, h, wd, cin = x.shape
  kh, kw, _, cout = w.shape
  pt, pb, ho = same_pads(h, kh, stride[0])
  pl, pr, wo = same_pads(wd, kw, stride[1])
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
  y = np.zeros((n, ho, wo, cout), dtype=x.dtype)
  for i in range(kh):
    for j in range(kw):
      y += xp[:, i:i + stride[0] * ho:stride[0], j:j + stride[1] * wo:stride[1], :] @ w[i, j]
  return y


def depthwise_same(x, w):
  """tf.contrib.layers.separable_conv2d(num_outputs=None, depth_multiplier=1, stride 1, SAME); w [kh,kw,C,1]."""
  n, h, wd, c = x.shape
  kh, kw = w.shape[:2]
  pt, pb, _ = same_pads(h, kh, 1)
  pl, pr, _ = same_pads(wd, kw, 1)
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
  y = np.zeros_like(x)
  for i in range(kh):
    for j in range(kw):
      y += xp[:, i:i + h, j:j + wd, :] * w[i, j, :, 0]
  return y


def maxpool_same(x, k, s):
  """tf.layers.max_pooling2d(padding='same'): padded cells never win (-inf)."""
  n, h, wd, c = x.shape
  pt, pb, ho = same_pads(h, k[0], s[0])
  pl, pr, wo = same_pads(wd, k[1], s[1])
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf)
  y = np.full((n, ho, wo, c), -np.inf, dtype=x.dtype)
  for i in range(k[0]):
    for j in range(k[1]):
      y = np.maximum(y, xp[:, i:i + s[0] * ho:s[0], j:j + s[1] * wo:s[1], :])
  return y


def bn_infer(x, p, scope):
  mean, var, beta = p[scope + '/BatchNorm/moving_mean'], p[scope + '/BatchNorm/moving_variance'], p[scope + '/BatchNorm/beta']
  return (x - mean) / np.sqrt(var + BN_EPS) + beta


def relu6(x):
  return np.minimum(np.maximum(x, 0), 6)


# (scope, out_channels, expansion, pool_after)   tinynet.py:172-203
MFCCNET_BLOCKS = [('block1_0', 64, 1, False), ('block2_0', 64, 6, True), ('block2_1', 64, 6, False),
                  ('block3_0', 128, 6, True), ('block3_1', 128, 6, False), ('block3_2', 128, 6, False),
                  ('block4_0', 192, 6, True), ('block4_1', 192, 6, False), ('block4_2', 192, 6, False), ('block4_3', 192, 6, False),
                  ('block5_0', 256, 6, False), ('block5_1', 256, 6, False), ('block5_2', 256, 6, False),
                  ('block6_0', 256, 6, True), ('block6_1', 256, 6, False), ('block6_2', 256, 6, False),
                  ('block7_0', 256, 6, False)]


def mfccnet_manifest(out_channels=256, prefix='mfcc_encoder/MfccNet/'):
  """Ordered [(tf_name, shape)] (contrib batch_norm has no gamma)."""
  m = []

  def conv_bn(scope, shape):
    m.append((scope + '/kernel', shape))
    for v in ('beta', 'moving_mean', 'moving_variance'):
      m.append((scope.rsplit('/', 1)[0] + '/BatchNorm/' + v, (shape[-1],)))
  conv_bn(prefix + 'block0_0/conv2d/conv2d', (9, 5, 1, 32))
  cin = 32
  for scope, cout, exp, _ in MFCCNET_BLOCKS:
    b = prefix + scope
    conv_bn(b + '/expansion_1x1_conv2d/conv2d', (1, 1, cin, cin * exp))
    m.append((b + '/depthwise_conv2d/SeparableConv2d/depthwise_weights', (7, 3, cin * exp, 1)))
    for v in ('beta', 'moving_mean', 'moving_variance'):
      m.append((b + '/depthwise_conv2d/BatchNorm/' + v, (cin * exp,)))
    conv_bn(b + '/projection_1x1_conv2d/conv2d', (1, 1, cin * exp, cout))
    if cout != cin:
      conv_bn(b + '/1x1_conv2d/conv2d', (1, 1, cin, cout))
    cin = cout
  conv_bn(prefix + 'block8_0/conv2d/conv2d', (1, 1, cin, out_channels))
  return m


def bfmnet_manifest():
  """MfccNet + dense / GRU / decoder variables (bfmnet.py:189-213)."""
  m = mfccnet_manifest()
  m += [('mfcc_encoder/dense/kernel', (256, 256)), ('mfcc_encoder/dense/bias', (256,)),
        ('rnn_module/dense/kernel', (256, 256)), ('rnn_module/dense/bias', (256,)),
        ('rnn_module/rnn/multi_rnn_cell/cell_0/gru_cell/gates/kernel', (512, 512)),
        ('rnn_module/rnn/multi_rnn_cell/cell_0/gru_cell/gates/bias', (512,)),
        ('rnn_module/rnn/multi_rnn_cell/cell_0/gru_cell/candidate/kernel', (512, 256)),
        ('rnn_module/rnn/multi_rnn_cell/cell_0/gru_cell/candidate/bias', (256,)),
        ('bfm_coeff_decoder/dense/kernel', (256, 128)), ('bfm_coeff_decoder/dense/bias', (128,)),
        ('bfm_coeff_decoder/dense_1/kernel', (128, 64)), ('bfm_coeff_decoder/dense_1/bias', (64,)),
        ('bfm_coeff_decoder/dense_2/kernel', (64, 64)), ('bfm_coeff_decoder/dense_2/bias', (64,))]
  return m


def init_bfmnet_params(seed=0, dtype=np.float64):
  """Synthetic stand-in for ckpt_bfmnet/bfmnet-65000 (external download): xavier-ish kernels, moving
  statistics that look trained (mean ~ N(0,0.1), variance ~ U[0.5,1.5]), GRU gate bias 1.0."""
  rng = np.random.default_rng(seed)
  p = {}
  for name, shape in bfmnet_manifest():
    if name.endswith('moving_variance'):
      p[name] = rng.uniform(0.5, 1.5, shape).astype(dtype)
    elif name.endswith('moving_mean') or name.endswith('beta'):
      p[name] = rng.normal(0, 0.1, shape).astype(dtype)
    elif name.endswith('gates/bias'):
      p[name] = np.ones(shape, dtype)
    elif name.endswith('bias'):
      p[name] = rng.normal(0, 0.05, shape).astype(dtype)
    else:
      fan_in = int(np.prod(shape[:-1])) if 'depthwise' not in name else shape[0] * shape[1]
      p[name] = rng.normal(0, np.sqrt(2.0 / max(fan_in, 1)), shape).astype(dtype)
  return p


def mfccnet_fwd(p, x, prefix='mfcc_encoder/MfccNet/'):
  """x [B, T5, 80, 1] -> [B, T5, 3, 256]  (tinynet.py:159-212, is_training=False)."""
  s = prefix + 'block0_0/conv2d'
  net = np.maximum(bn_infer(conv2d_same(x, p[s + '/conv2d/kernel'], (1, 2)), p, s), 0)
  for scope, cout, exp, pool in MFCCNET_BLOCKS:
    b = prefix + scope
    inp = net
    net = relu6(bn_infer(conv2d_same(net, p[b + '/expansion_1x1_conv2d/conv2d/kernel'], (1, 1)), p, b + '/expansion_1x1_conv2d'))
    net = relu6(bn_infer(depthwise_same(net, p[b + '/depthwise_conv2d/SeparableConv2d/depthwise_weights']), p, b + '/depthwise_conv2d'))
    net = bn_infer(conv2d_same(net, p[b + '/projection_1x1_conv2d/conv2d/kernel'], (1, 1)), p, b + '/projection_1x1_conv2d')
    if net.shape[3] != inp.shape[3]:
      inp = bn_infer(conv2d_same(inp, p[b + '/1x1_conv2d/conv2d/kernel'], (1, 1)), p, b + '/1x1_conv2d')
    net = net + inp
    if pool:
      net = maxpool_same(net, (2, 2), (1, 2))
  s = prefix + 'block8_0/conv2d'
  return np.maximum(bn_infer(conv2d_same(net, p[s + '/conv2d/kernel'], (1, 1)), p, s), 0)


def leaky_relu(x, a=0.2):
  return np.where(x >= 0, x, a * x)    # tf.nn.leaky_relu = max(x, a*x)


def sigmoid(x):
  return 1.0 / (1.0 + np.exp(-x))


def gru_seq(x, seq_len, wg, bg, wc, bc):
  """tf.contrib.rnn.GRUCell through dynamic_rnn: gates [r,u] = sigmoid([x,h] Wg + bg),
  c = tanh([x, r*h] Wc + bc), h' = u*h + (1-u)*c; past seq_len: output 0, state frozen."""
  B, T, _ = x.shape
  H = wc.shape[1]
  h = np.zeros((B, H), dtype=x.dtype)
  out = np.zeros((B, T, H), dtype=x.dtype)
  for t in range(T):
    g = sigmoid(np.concatenate([x[:, t], h], axis=1) @ wg + bg)
    r, u = g[:, :H], g[:, H:]
    c = np.tanh(np.concatenate([x[:, t], r * h], axis=1) @ wc + bc)
    hn = u * h + (1 - u) * c
    live = (t < np.asarray(seq_len))[:, None]
    h = np.where(live, hn, h)
    out[:, t] = np.where(live, hn, 0)
  return out


def bfmnet_fwd(p, ears, mfccs, seq_len, decoder_masks=None):
  """BFMNet.build_inference_op (bfmnet.py:325-333): ears [B,T,1], mfccs [B,5T,80] -> [B,T,64].
  decoder_masks: None = deterministic (both tf.nn.dropout of BFMCoeffDecoder.build_network, bfmnet.py:114,116, omitted: DESIGN.md
  section 4); (m0 [B,T,128], m1 [B,T,64]) with entries 0 or 1 / keep_prob = one draw of those two dropouts, which the reference
  applies at inference too."""
  B = mfccs.shape[0]
  feat = mfccnet_fwd(p, mfccs[..., None])
  enc = maxpool_same(feat, (5, 3), (5, 3)).reshape(B, -1, 256)                      # bfmnet.py:35-36
  enc = leaky_relu(enc @ p['mfcc_encoder/dense/kernel'] + p['mfcc_encoder/dense/bias'])
  c1 = leaky_relu(enc @ p['rnn_module/dense/kernel'] + p['rnn_module/dense/bias'])
  g = 'rnn_module/rnn/multi_rnn_cell/cell_0/gru_cell/'
  rnn = gru_seq(c1, seq_len, p[g + 'gates/kernel'], p[g + 'gates/bias'], p[g + 'candidate/kernel'], p[g + 'candidate/bias'])
  d = leaky_relu(rnn @ p['bfm_coeff_decoder/dense/kernel'] + p['bfm_coeff_decoder/dense/bias'])
  if decoder_masks is not None:
    d = d * decoder_masks[0]                                                         # bfmnet.py:114
  d = leaky_relu(d @ p['bfm_coeff_decoder/dense_1/kernel'] + p['bfm_coeff_decoder/dense_1/bias'])
  if decoder_masks is not None:
    d = d * decoder_masks[1]                                                         # bfmnet.py:116
  out = d @ p['bfm_coeff_decoder/dense_2/kernel'] + p['bfm_coeff_decoder/dense_2/bias']
  e = ears * np.array([-2.0, -2.0, -2.0, -4.0])                                      # bfmnet.py:209
  out = out.copy()
  out[..., 16:20] += e                                                               # tf.pad(ears, [16, 44])
  return {'MfccEncoder': enc, 'RNNModule': rnn, 'BFMCoeffDecoder': out}
