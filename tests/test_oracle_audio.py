"""Pins oracle/audio_ref.py with independent identities and a torch-CPU float64 second opinion."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import audio_ref as ar


def test_stft_matches_naive_dft_and_frame_count():
  rng = np.random.default_rng(0)
  pcm = rng.normal(size=(2, 1000))
  mag = ar.stft_mag(pcm, 512, 128, 512)
  assert mag.shape == (2, 1 + (1000 - 512) // 128, 257)
  n = np.arange(512)
  w = 0.5 - 0.5 * np.cos(2 * np.pi * n / 512)          # periodic Hann
  for f in (0, 3):
    x = pcm[1, f * 128:f * 128 + 512] * w
    k = np.arange(257)[:, None]
    naive = np.abs((x[None, :] * np.exp(-2j * np.pi * k * n[None, :] / 512)).sum(axis=1))
    np.testing.assert_allclose(mag[1, f], naive, rtol=1e-9, atol=1e-9)


def test_stft_against_two_third_party_implementations():
  """A third-party second opinion for the one piece of the audio front-end the reference takes from a library (tf.signal.stft,
  generator/generator.py:63): scipy.signal.stft and torch.stft configured the way TF's documentation defines the op - periodic Hann
  window of 512, hop 128, 512-point rFFT, no centring, no padding at either end.  (scipy divides by the window sum.)"""
  import scipy.signal as ss
  rng = np.random.default_rng(3)
  pcm = np.clip(0.1 * rng.normal(size=(3, 16384)) + 0.3 * np.sin(2 * np.pi * 440 * np.arange(16384) / 16000.0), -1, 1)
  mag = ar.stft_mag(pcm, 512, 128, 512)
  win = ss.get_window("hann", 512, fftbins=True)                     # fftbins=True: the PERIODIC window
  np.testing.assert_allclose(win, ar.hann_periodic(512), atol=1e-15)
  _, _, z = ss.stft(pcm, window=win, nperseg=512, noverlap=384, nfft=512, boundary=None, padded=False, axis=-1)
  sp = np.abs(z).transpose(0, 2, 1) * win.sum()
  assert sp.shape == mag.shape == (3, 125, 257)
  np.testing.assert_allclose(mag, sp, rtol=1e-9, atol=1e-9)
  tz = torch.stft(torch.tensor(pcm), n_fft=512, hop_length=128, win_length=512, window=torch.hann_window(512, periodic=True, dtype=torch.float64),
                  center=False, return_complex=True)
  np.testing.assert_allclose(mag, tz.abs().numpy().transpose(0, 2, 1), rtol=1e-9, atol=1e-9)


def test_mel_matrix_properties():
  m = ar.linear_to_mel_weight_matrix()
  assert m.shape == (257, 80)
  assert np.all(m[0] == 0)                               # DC bin zeroed
  assert m.min() >= 0 and m.max() <= 1.0
  centers = (m * np.arange(257)[:, None]).sum(0) / m.sum(0)
  assert np.all(np.diff(centers) > 0)                    # monotone band centres
  hz = centers * 8000 / 256
  assert 80 < hz[0] < 200 and 7000 < hz[-1] < 7600       # 80..7600 Hz edges (generator.py:68)
  assert np.all((m > 0).sum(0) >= 1)


def test_logmel_of_sine_peaks_in_the_right_band_and_pcm_length():
  t = np.arange(16384) / 16000.0
  pcm = 0.5 * np.sin(2 * np.pi * 1000.0 * t)[None, :]
  lm = ar.extract_mfcc(pcm)
  assert lm.shape == (1, 125, 80)
  m = ar.linear_to_mel_weight_matrix()
  band = np.argmax(m[int(round(1000 / (8000 / 256)))])
  assert abs(int(np.argmax(lm[0, 60])) - band) <= 1
  assert ar.pcm_length_for(25) == 16384                   # infer_bfmvid.py:164: 1 s -> 125 frames


def test_same_padding_rules():
  assert ar.same_pads(80, 5, 2) == (1, 2, 40)             # SURVEY 8a padding caveat
  assert ar.same_pads(125, 2, 1) == (0, 1, 125)
  assert ar.same_pads(5, 2, 2) == (0, 1, 3)
  assert ar.same_pads(125, 5, 5) == (0, 0, 25)


def test_mfccnet_pieces_vs_torch():
  rng = np.random.default_rng(1)
  x = rng.normal(size=(2, 9, 10, 6))
  w = rng.normal(size=(7, 3, 6, 1))
  y = ar.depthwise_same(x, w)
  yt = F.conv2d(torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(w).permute(2, 3, 0, 1).contiguous(), padding=(3, 1), groups=6)
  np.testing.assert_allclose(y, yt.permute(0, 2, 3, 1).numpy(), rtol=1e-10, atol=1e-12)
  x = rng.normal(size=(2, 7, 5, 4))
  y = ar.maxpool_same(x, (2, 2), (1, 2))
  xt = F.pad(torch.tensor(x).permute(0, 3, 1, 2), (0, 1, 0, 1), value=float("-inf"))
  np.testing.assert_allclose(y, F.max_pool2d(xt, (2, 2), (1, 2)).permute(0, 2, 3, 1).numpy())
  x = rng.normal(size=(1, 6, 80, 1))
  w = rng.normal(size=(9, 5, 1, 4))
  y = ar.conv2d_same(x, w, (1, 2))
  xt = F.pad(torch.tensor(x).permute(0, 3, 1, 2), (1, 2, 4, 4))
  yt = F.conv2d(xt, torch.tensor(w).permute(3, 2, 0, 1).contiguous(), stride=(1, 2))
  np.testing.assert_allclose(y, yt.permute(0, 2, 3, 1).numpy(), rtol=1e-10, atol=1e-12)


def test_gru_vs_torch_grucell_and_sequence_masking():
  rng = np.random.default_rng(2)
  B, T, H = 3, 6, 256
  x = rng.normal(size=(B, T, H)) * 0.5
  wg, bg = rng.normal(size=(2 * H, 2 * H)) * 0.05, rng.normal(size=2 * H) * 0.1
  wc, bc = rng.normal(size=(2 * H, H)) * 0.05, rng.normal(size=H) * 0.1
  out = ar.gru_seq(x, [6, 4, 1], wg, bg, wc, bc)
  # torch GRUCell: r,z,n with n = tanh(Wx + b + r*(Uh + b')) differs from TF (r*h inside the matmul):
  # restate the TF cell directly in torch float64 instead
  h = torch.zeros(B, H, dtype=torch.float64)
  for t in range(T):
    xt = torch.tensor(x[:, t])
    g = torch.sigmoid(torch.cat([xt, h], 1) @ torch.tensor(wg) + torch.tensor(bg))
    r, u = g[:, :H], g[:, H:]
    c = torch.tanh(torch.cat([xt, r * h], 1) @ torch.tensor(wc) + torch.tensor(bc))
    hn = u * h + (1 - u) * c
    live = torch.tensor([t < 6, t < 4, t < 1])[:, None]
    h = torch.where(live, hn, h)
    np.testing.assert_allclose(out[:, t], torch.where(live, hn, torch.zeros_like(hn)).numpy(), rtol=1e-10, atol=1e-12)
  assert np.all(out[2, 1:] == 0)


def test_bfmnet_shapes_and_manifest():
  m = ar.bfmnet_manifest()
  names = [n for n, _ in m]
  assert len(set(names)) == len(names)
  assert "mfcc_encoder/MfccNet/block0_0/conv2d/conv2d/kernel" in names
  assert "mfcc_encoder/MfccNet/block3_0/depthwise_conv2d/SeparableConv2d/depthwise_weights" in names
  assert "mfcc_encoder/MfccNet/block3_0/1x1_conv2d/BatchNorm/moving_variance" in names
  conv_w = sum(int(np.prod(s)) for n, s in m if n.endswith("kernel") and "MfccNet" in n or n.endswith("depthwise_weights"))
  assert 7.0e6 < conv_w < 8.5e6                            # 7.89 M conv weights
  p = ar.init_bfmnet_params(0)
  rng = np.random.default_rng(0)
  out = ar.bfmnet_fwd(p, np.full((1, 5, 1), 0.3), rng.normal(size=(1, 25, 80)), [5])
  assert out["BFMCoeffDecoder"].shape == (1, 5, 64) and out["MfccEncoder"].shape == (1, 5, 256)
