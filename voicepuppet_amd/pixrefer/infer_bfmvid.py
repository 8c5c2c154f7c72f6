#!/usr/bin/env python
# -*- encoding: utf-8 -*-
"""End-to-end inference, same CLI as the reference (voicepuppet/pixrefer/infer_bfmvid.py):

    python voicepuppet/pixrefer/infer_bfmvid.py --config_path config/params.yml <image 1536x512> <audio.wav>

wav -> log-mel -> BFMNet -> 64 expression coefficients per video frame -> (3-D face conditioning image) ->
PixReferNet -> output/<i>.jpg (-> ffmpeg mux when ffmpeg exists).

The per-frame conditioning image is the BFM reconstruction + rasteriser (utils/reconstruct_mesh.py,
utils/cython/mesh_core.cpp), which run on the device for the whole clip at once (voicepuppet_amd.utils.reconstruct_mesh.
ClipRenderer).  They need external assets: BFM/BFM_model_front.mat and the photo's own 257 coefficients + alignment, which
the reference obtains from FaceReconModel.pb and the dlib/MXNet aligners (infer_bfmvid.py:47-74; out of scope).  Pass those
as `--bfmcoeff <npz with bfmcoeff[1,257], transform_params[5], center_x, center_y, ratio>`; without them every frame is
conditioned on the 3-D face panel of the input image (a warning is printed), which still exercises audio -> coefficients
-> generator.
Frames are batched on the device with per-sample batch-norm statistics, which is arithmetically the
reference's batch-1 loop (infer_bfmvid.py:231-243).
"""
import logging
import os
import shutil
import subprocess
import sys
from optparse import OptionParser

import numpy as np

sys.path.append(os.getcwd())

from voicepuppet_amd.bfmnet.bfmnet import BFMNet
from voicepuppet_amd.generator.generator import DataGenerator
from voicepuppet_amd.generator.loader import ImageLoader, WavLoader
from voicepuppet_amd.pixrefer.pixrefer import PixReferNet
from voicepuppet_amd.runtime import Session, convert_to_tensor, placeholder

logging.basicConfig(level=logging.INFO, format='%(asctime)s - %(name)s - %(levelname)s - %(message)s')
logger = logging.getLogger(__name__)


def prepare_pcm(pcm, gen):
  """infer_bfmvid.py:162-167: pad so that the mel sequence is exactly pad_len * frame_mfcc_scale frames."""
  pad_len = int(1 + pcm.shape[0] / gen.frame_wav_scale)
  pcm_length = gen.hop_step * (pad_len * gen.frame_mfcc_scale - 1) + gen.win_length
  if (pcm.shape[0] < pcm_length):
    pcm = np.pad(pcm, (0, pcm_length - pcm.shape[0]), 'constant', constant_values=(0))
  return pcm[:pcm_length][np.newaxis, :], pad_len


def splice_coeff(bfmcoeff, expr_seq):
  """infer_bfmvid.py:223-224: identity 0:80 and 144:257 from the photo, expression 80:144 predicted."""
  tiled = np.tile(bfmcoeff[:, np.newaxis, :], [1, expr_seq.shape[1], 1])
  return np.concatenate([tiled[:, :, :80], expr_seq, tiled[:, :, 144:]], axis=2)


def angle_sequence(frames, start=(0.0, 0.0, 0.0), shift=0.005):
  """The head-sway state machine of render_face (infer_bfmvid.py:76-90), unrolled for a clip: all three angles advance by
  `shift` per frame (float32 accumulation, like the reference's global array) and the direction flips after |angle_y| > 0.03."""
  angles = np.array([start], dtype=np.float32)
  out = np.zeros((frames, 3), np.float32)
  for i in range(frames):
    angles[0][0] += shift
    angles[0][1] += shift
    angles[0][2] += shift
    if (angles[0][1] > 0.03 or angles[0][1] < -0.03):
      shift = -shift
    out[i] = angles[0]
  return out


_GENERATORS = {}      # see main(): generator networks kept between clips of one process
_RENDERERS = {}


def render_faces(renderer, center_x, center_y, ratio, bfm_coeff_seq, img_shape, transform_params, on_device=False):
  """render_face (infer_bfmvid.py:79-122) for every frame of the clip: one device pass for reconstruction + rasterisation, then the
  reference's channel swap / cv2.resize / paste for all frames in one more launch (csrc/resize.hip: OpenCV's fixed-point bilinear,
  byte for byte; voicepuppet_amd/utils/cv_resize.py)."""
  from voicepuppet_amd.utils.cv_resize import resize_paste_u8
  ratio = ratio * transform_params[2]
  tx = -int((transform_params[3] / ratio))
  ty = -int((transform_params[4] / ratio))
  T = bfm_coeff_seq.shape[0]
  images, _ = renderer(bfm_coeff_seq.astype(np.float32), angle_sequence(T))      # [T, 224, 224, 3] uint8 on the device, rasteriser order
  side = int(round(images.shape[1] / ratio))
  cx, cy = side // 2, side // 2
  out = resize_paste_u8(images, side, side, (img_shape[0], img_shape[1]), center_y - cy - ty, center_x - cx - tx, swap_rb=True)   # :110-121
  return out if on_device else out.cpu().numpy()      # on_device: the clip loop consumes the frames where they are (no PCIe round trip)


def main(argv=None):
  cmd_parser = OptionParser(usage="usage: %prog [options] --config_path <> image audio")
  cmd_parser.add_option('--config_path', type="string", dest="config_path", help='the config yaml file')
  cmd_parser.add_option('--frame_batch', type="int", dest="frame_batch", default=8, help='frames per device batch')
  cmd_parser.add_option('--bfmcoeff', type="string", dest="bfmcoeff", default=None,
                        help='npz with the photo\'s bfmcoeff [1,257], transform_params [5], center_x, center_y, ratio')
  cmd_parser.add_option('--output_dir', type="string", dest="output_dir", default='output',
                        help='frame directory (the reference always writes output/; infer_clips.py gives every clip its own)')
  opts, argv = cmd_parser.parse_args(argv)

  if (opts.config_path is None):
    logger.error('Please check your parameters.')
    exit(0)
  config_path = opts.config_path
  if (not os.path.exists(config_path)):
    logger.error('config_path not exists')
    exit(0)

  image_file, audio_file = argv

  out_dir = opts.output_dir
  if not os.path.exists(out_dir):
    os.makedirs(out_dir)
  for file in os.listdir(out_dir):
    p = os.path.join(out_dir, file)
    shutil.rmtree(p) if os.path.isdir(p) else os.remove(p)

  batch_size = 1
  ### Generator for inference setting
  infer_generator = DataGenerator(config_path)
  params = infer_generator.params
  params.batch_size = batch_size
  infer_generator.set_params(params)
  pcm = WavLoader(sr=infer_generator.sample_rate).get_data(audio_file)
  pcm_slice, pad_len = prepare_pcm(pcm, infer_generator)
  mfcc = infer_generator.extract_mfcc(pcm_slice)

  img_size = 512
  img = ImageLoader().get_data(image_file)[:, :, ::-1]      # RGB float in [0,1], 512 x 1536
  face3d_refer = img[:, 512:512 * 2, :]
  fg_refer = img[:, :512, :] * img[:, 512 * 2:, :]
  img = img[:, :512, :]

  with Session() as sess:
    seq_len = convert_to_tensor(np.array([pad_len], dtype=np.int32))
    ear = convert_to_tensor(np.random.rand(1, pad_len, 1).astype(np.float32) / 100)

    ### BFMNet setting
    bfmnet = BFMNet(config_path)
    params = bfmnet.params
    params.batch_size = 1
    bfmnet.set_params(params)
    bfmnet_nodes = bfmnet.build_inference_op(ear, mfcc, seq_len)

    ### Vid2VidNet setting
    # A process that runs many clips (infer_clips.py) keeps the generator - its plan, its 35 M restored parameters - between calls:
    # building and restoring it is 0.35 s, the frames of an 8 s clip take 0.25 s.  Keyed by what defines it (config, frame batch, image
    # size, the checkpoint file and its modification time)
    nb = max(1, min(opts.frame_batch, pad_len))
    pix_ckpt = 'ckpt_pixrefer/pixrefernet-20000'
    pix_file = next((f for f in (pix_ckpt + '.index', pix_ckpt + '.npz') if os.path.exists(f)), None)
    pix_key = (os.path.abspath(config_path), nb, img_size, os.path.abspath(pix_file) if pix_file else None,
               os.path.getmtime(pix_file) if pix_file else None)
    cached = _GENERATORS.get(pix_key)
    if cached is None:
      vid2vidnet = PixReferNet(config_path)
      params = vid2vidnet.params
      params.batch_size = nb
      params.add_hparam('is_training', False)
      vid2vidnet.set_params(params)
      inputs_holder = placeholder([None, img_size, img_size, 6])
      fg_inputs_holder = placeholder([None, img_size, img_size, 3])
      targets_holder = placeholder([None, img_size, img_size, 3])
      vid2vid_nodes = vid2vidnet.build_inference_op(inputs_holder, fg_inputs_holder, targets_holder)
    else:
      vid2vidnet, inputs_holder, fg_inputs_holder, targets_holder, vid2vid_nodes = cached

    # infer_bfmvid.py:217-218: the TensorFlow checkpoints themselves (or .npz archives with the same variable names)
    for net, ckpt in ((bfmnet, 'ckpt_bfmnet/bfmnet-65000'), (vid2vidnet, pix_ckpt)):
      if net is vid2vidnet and cached is not None:
        continue                                     # restored when it was built
      if os.path.exists(ckpt + '.index'):
        net.restore(ckpt)
      elif os.path.exists(ckpt + '.npz'):
        net.restore(ckpt + '.npz')
      else:
        logger.warning('%s not found: running with randomly initialised weights', ckpt)
        if net is bfmnet:
          bfmnet.init_variables()

    if cached is None:
      if len(_GENERATORS) >= 4:
        _GENERATORS.clear()                          # (a handful of shapes at most: do not grow without bound)
      _GENERATORS[pix_key] = (vid2vidnet, inputs_holder, fg_inputs_holder, targets_holder, vid2vid_nodes)

    ### Run inference
    bfm_coeff_seq = sess.run(bfmnet_nodes['BFMCoeffDecoder'])
    face3d_seq = None
    if opts.bfmcoeff and os.path.exists(os.path.join('BFM', 'BFM_model_front.mat')):
      from scipy.io import loadmat
      from voicepuppet_amd.utils.reconstruct_mesh import ClipRenderer

      class _BFM(object):      # utils/bfm_load_data.py:9-21
        def __init__(self, model):
          for k in ('meanshape', 'idBase', 'exBase', 'meantex', 'texBase', 'point_buf', 'tri'):
            setattr(self, k, model[k])
          self.keypoints = np.squeeze(model['keypoints']).astype(np.int32) - 1
      photo = np.load(opts.bfmcoeff)
      coeff_seq = splice_coeff(photo['bfmcoeff'].reshape(1, 257), bfm_coeff_seq)[0]
      mat = os.path.join('BFM', 'BFM_model_front.mat')
      rkey = (os.path.abspath(mat), os.path.getmtime(mat))
      if rkey not in _RENDERERS:
        _RENDERERS.clear()
        _RENDERERS[rkey] = ClipRenderer(_BFM(loadmat(mat)))           # the face model's bases on the device: once per process
      face3d_seq = render_faces(_RENDERERS[rkey], int(photo['center_x']),
                                int(photo['center_y']), float(photo['ratio']), coeff_seq, (img_size, img_size, 3),
                                photo['transform_params'], on_device=True)
    else:
      logger.warning('BFM assets unavailable: conditioning every frame on the reference 3-D face panel')

    T = bfm_coeff_seq.shape[1]
    # the three feeds live on the device: only what changes per batch is written (the rendered faces are already there)
    import torch
    dev = torch.device('cuda', torch.cuda.current_device())
    inputs = torch.zeros([nb, img_size, img_size, 6], dtype=torch.float32, device=dev)
    fg_inputs = torch.zeros([nb, img_size, img_size, 3], dtype=torch.float32, device=dev)
    targets = torch.full([nb, img_size, img_size, 3], 0.5, dtype=torch.float32, device=dev)
    refer_t = torch.as_tensor(np.ascontiguousarray(face3d_refer, dtype=np.float32)).to(dev)
    inputs[:, ..., 0:3] = refer_t
    fg_inputs[:, ..., 0:3] = torch.as_tensor(np.ascontiguousarray(fg_refer, dtype=np.float32)).to(dev)
    if face3d_seq is None:
      inputs[:, ..., 3:6] = refer_t
    from PIL import Image
    # jpg encoding (2-3 ms per 512 x 512 frame on one core, longer than the generator takes for it) on a small thread pool: PIL's
    # encoder releases the GIL, the files are byte-identical to a serial loop, and the next batch's launches overlap the writes
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2) - 1)))
    pending = []

    def write_jpg(arr_u8, path):
      Image.fromarray(arr_u8).save(path)
    try:
      for i0 in range(0, T, nb):
        idx = [min(i0 + k, T - 1) for k in range(nb)]
        if face3d_seq is not None:
          # render_face returns a BGR canvas that the caller swaps again (infer_bfmvid.py:233): net effect = rasteriser order
          inputs[:, ..., 3:6] = face3d_seq[idx].flip(-1).to(torch.float32) / 255.0
        for k, i in enumerate(idx):
          bg = 'background/{}.jpg'.format(i % 100 + 1)
          if os.path.exists(bg):
            targets[k] = torch.as_tensor(np.ascontiguousarray(ImageLoader(resize=(img_size, img_size)).get_data(bg)[:, :, ::-1], dtype=np.float32)).to(dev)
          else:
            targets[k] = 0.5
        # (the reference fetches 'Outputs' and the unused 'Outputs_FG' as float32 and scales on the host, infer_bfmvid.py:240-243; the
        # uint8 frame is formed on the device here: 6 MB instead of 50 MB across PCIe per batch of 8, identical bytes)
        frames = sess.run([vid2vid_nodes['Outputs_u8']],
                          feed_dict={inputs_holder: inputs, fg_inputs_holder: fg_inputs, targets_holder: targets})[0]
        for k in range(nb):
          if i0 + k < T:
            pending.append(pool.submit(write_jpg, frames[k], os.path.join(out_dir, '{}.jpg'.format(i0 + k))))
      for f in pending:
        f.result()           # every frame is on disk (and any write error surfaces) before ffmpeg reads the directory
    finally:
      pool.shutdown()        # (also on an error in the loop: the writer threads must not outlive the call)

    if shutil.which('ffmpeg'):
      # same command line as infer_bfmvid.py:245, as an argument vector (no shell: the audio path is user input)
      subprocess.call(['ffmpeg', '-i', os.path.join(out_dir, '%d.jpg'), '-i', audio_file, '-c:v', 'libx264', '-c:a', 'aac',
                       '-strict', 'experimental', '-y', out_dir.rstrip('/') + '.mp4'])
    else:
      logger.warning('ffmpeg not found: frames are in output/, no mp4 written')


if (__name__ == '__main__'):
  main()
