#!/usr/bin/env python
# -*- encoding: utf-8 -*-
"""PixReferNet training entry point, same CLI as the reference (voicepuppet/pixrefer/train_pixrefer.py):

    python voicepuppet/pixrefer/train_pixrefer.py --config_path config/params.yml

One process per GPU; under torch.distributed.run (WORLD_SIZE > 1) the G/D gradient arenas are
all-reduced over RCCL every step (plain data parallel, per-replica batch statistics).
Extra, optional flags (defaults reproduce the reference run): --steps, --batch_size, --img_size, --resume (continue from the
latest checkpoint in ckpt_pixrefer - the block the reference keeps commented out at train_pixrefer.py:93-99; what a restart after a
lost rank does: a FRESH process from the last checkpoint, never a re-exec of one that touched the GPU).
"""
import logging
import os
import sys
import time
from optparse import OptionParser

import numpy as np

sys.path.append(os.getcwd())

from voicepuppet_amd.generator.generator import PixReferDataGenerator
from voicepuppet_amd.pixrefer.pixrefer import PixReferNet
from voicepuppet_amd.runtime import Session

logging.basicConfig(level=logging.INFO, format='%(asctime)s - %(name)s - %(levelname)s - %(message)s')
logger = logging.getLogger(__name__)


def mkdir(path):
  if not os.path.exists(path):
    os.makedirs(path)


def save_image(path, arr):
  from PIL import Image
  Image.fromarray((np.clip(arr, 0, 1) * 255).astype(np.uint8)).save(path)


def main(argv=None):
  cmd_parser = OptionParser(usage="usage: %prog [options] --config_path <>")
  cmd_parser.add_option('--config_path', type="string", dest="config_path", help='the config yaml file')
  cmd_parser.add_option('--steps', type="int", dest="steps", default=None, help='iterations to run (default: training.epochs)')
  cmd_parser.add_option('--batch_size', type="int", dest="batch_size", default=2, help='per-GPU batch (reference: 2)')
  cmd_parser.add_option('--img_size', type="int", dest="img_size", default=None, help='override amd.img_size')
  cmd_parser.add_option('--resume', action="store_true", dest="resume", default=False, help='restore the latest checkpoint of save_dir')
  opts, _ = cmd_parser.parse_args(argv)

  if (opts.config_path is None):
    logger.error('Please check your parameters.')
    exit(0)
  config_path = opts.config_path
  if (not os.path.exists(config_path)):
    logger.error('config_path not exists')
    exit(0)

  import torch
  import torch.distributed as dist
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
  group = None
  if world > 1:
    from voicepuppet_amd.parallel import init_distributed
    group = init_distributed('nccl')      # collective timeout + async error handling (parallel.py)

  batch_size = opts.batch_size
  ### Generator for training setting
  train_generator = PixReferDataGenerator(config_path)
  params = train_generator.params
  params.dataset_path = params.train_dataset_path
  params.batch_size = batch_size
  if opts.img_size:
    params.img_size = opts.img_size
  train_generator.set_params(params)
  # the per-sample crop / resize / packing on the device, decoded uint8 frames copied under the previous step;
  # amd: {device_input_pipeline: false} in params.yml restores the reference's host pipeline
  use_device_pipeline = (params.get('amd') or {}).get('device_input_pipeline', True) not in (False, 'false', 'no', 0)
  train_dataset = train_generator.get_device_dataset() if use_device_pipeline else train_generator.get_dataset()

  sess = Session()
  train_iter = train_dataset.make_one_shot_iterator()

  ### Vid2VidNet setting
  vid2vidnet = PixReferNet(config_path)
  params = vid2vidnet.params
  epochs = opts.steps if opts.steps is not None else params.training['epochs']
  params.add_hparam('max_to_keep', 2)
  params.add_hparam('save_dir', 'ckpt_pixrefer')
  params.add_hparam('save_name', 'pixrefernet')
  params.add_hparam('save_step', 5000)
  params.add_hparam('summary_step', 100)
  params.add_hparam('summary_dir', 'log/summary_pixrefer')
  params.batch_size = batch_size
  params.add_hparam('is_training', True)
  params.sess = sess
  params.vgg_model_path = os.path.join(params.model_dir, 'vgg_16.ckpt')
  vid2vidnet.set_params(params)
  vid2vidnet.group = group

  if rank == 0:
    mkdir(params.save_dir)
    mkdir(params.summary_dir)

  train_nodes = vid2vidnet.build_train_op(*train_iter.get_next())
  if use_device_pipeline:
    vid2vidnet.engine.use_streams(3)      # the input prefetcher's stream is the fourth busy one (include/vp_hip.h vp_pixrefer_use_streams)
  # --resume / amd: {resume: true}: continue from the latest checkpoint of save_dir.  `epochs` stays the TOTAL number of iterations of
  # the run (the learning-rate schedule is a function of global_step, which the checkpoint restores), so a run restarted at
  # global_step 120000 of 200000 trains the remaining 40000 iterations, not another 100000.  The data pipeline is NOT part of a
  # checkpoint (nor is it in the reference: tf.data's shuffle state is not saved by its Saver): a resumed run draws fresh shuffles,
  # i.e. it continues the optimisation, not the bit-exact sample sequence of the interrupted run.
  resumed_step = 0
  resume_flag = opts.resume or (params.get('amd') or {}).get('resume', False)
  if resume_flag and rank == 0:     # rank 0 reads the checkpoint; the other replicas receive the WHOLE restored state below
    from voicepuppet_amd.utils import tf_checkpoint
    latest = tf_checkpoint.latest_checkpoint(params.save_dir) if os.path.isdir(params.save_dir) else None
    if latest is not None and not os.path.exists(latest + '.index'):
      logger.error('resume: %s/checkpoint names %s, which does not exist', params.save_dir, latest)
      exit(1)
    if latest is None:
      if opts.resume:          # asked for on the command line: starting from scratch silently would discard a run
        logger.error('resume: no checkpoint state in %s', params.save_dir)
        exit(1)
      logger.warning('resume: no checkpoint in %s, starting from the initial weights', params.save_dir)
    else:
      vid2vidnet.restore(latest)
      resumed_step = vid2vidnet.global_step
      logger.info('resumed from %s (global_step %d)', latest, resumed_step)
  if world > 1:   # identical initial weights on every replica
    for a in (vid2vidnet.engine.params_g, vid2vidnet.engine.params_d, vid2vidnet.engine.params_vgg):
      dist.broadcast(a, 0)
    vid2vidnet.engine.params_changed()
    if resume_flag:
      # ... and on a resumed run everything else a checkpoint restores: the Adam slots, the two update counters and global_step (the
      # number of iterations left and the learning rate are functions of it).  A replica that restored from its own view of save_dir -
      # or saw none and started at step 0 - would run a different number of iterations and hang its peers in a collective
      eng = vid2vidnet.engine
      for key in ('g', 'd'):
        for a in eng.adam[key]:
          dist.broadcast(a, 0)
      st = torch.tensor([vid2vidnet.global_step, eng.t_g, eng.t_d], dtype=torch.int64, device=eng.device)
      dist.broadcast(st, 0)
      vid2vidnet.global_step, eng.t_g, eng.t_d = (int(x) for x in st.tolist())
      resumed_step = vid2vidnet.global_step

  # rank liveness: losses are read only on summary steps, so a peer lost inside a collective shows up as a device
  # stream that stops finishing steps; the watchdog then ends this rank non-zero and torch.distributed.run stops the others
  from voicepuppet_amd.parallel import StepWatchdog
  dog = StepWatchdog(rank=rank, device=torch.cuda.current_device()) if world > 1 else None
  # Saver(max_to_keep) also counts the checkpoints an earlier process of this run left in save_dir
  saved = []
  if rank == 0 and resume_flag:
    from voicepuppet_amd.utils import tf_checkpoint
    saved = tf_checkpoint.list_checkpoints(params.save_dir, params.save_name)
  t0 = time.time()
  remaining = max(0, epochs - resumed_step // 2)      # both optimisers bump global_step: 2 per iteration
  for i in range(remaining):
    ### Run training
    # the reference fetches the three losses every step and prints them every summary_step (train_pixrefer.py:134-143).  Reading a
    # loss waits for the step, and a host that waits every step cannot enqueue the next one under it: the losses are fetched on the
    # steps that print them (both apply_gradients bump global_step: + 2 per iteration)
    summary = (vid2vidnet.global_step + 2) % params.summary_step == 0
    fetch = [train_nodes['Train_op'], train_nodes['Lr'], train_nodes['Global_step']]
    if summary:
      fetch += [train_nodes['Gen_loss_GAN'], train_nodes['Gen_loss_L1'], train_nodes['Discrim_loss']]
    vals = sess.run(fetch)
    lr, global_step = vals[1], vals[2]
    if dog is not None:
      ev = torch.cuda.Event()
      ev.record()
      dog.beat(ev)
    if summary:
      gen_loss_GAN, gen_loss_L1, discrim_loss = vals[3:6]
    if (summary and rank == 0):
      print('Step {}, Lr= {:.2e}: \n\tgen_loss_GAN= {:.3f}, \n\tgen_loss_L1= {:.3f}, \n\tdiscrim_loss= {:.3f}'.format(
          global_step, lr, gen_loss_GAN, gen_loss_L1, discrim_loss))
      fps = (i + 1) * batch_size * world / (time.time() - t0)
      logger.info('%.1f frames/s', fps)
      eng = vid2vidnet.engine
      save_image(os.path.join(params.summary_dir, 'outputs_%d.png' % global_step), ((eng.tensor('Outputs_raw')[0] + 1) / 2).cpu().numpy())

    ### Save checkpoint
    if (global_step % params.save_step == 0 and rank == 0):
      # train_pixrefer.py:150: Saver(max_to_keep).save(sess, 'ckpt_pixrefer/pixrefernet', global_step) - a TensorFlow V2 checkpoint
      if dog is not None:
        dog.pause()                     # a long host-side save is not a hung rank
      path = vid2vidnet.save(os.path.join(params.save_dir, '%s-%d' % (params.save_name, global_step)))
      if dog is not None:
        dog.resume()
      if path not in saved:
        saved.append(path)
      while len(saved) > params.max_to_keep:
        old = saved.pop(0)
        for suffix in ('.index', '.data-00000-of-00001'):
          if os.path.exists(old + suffix):
            os.remove(old + suffix)
  if dog is not None:
    dog.close()
  if world > 1:
    dist.destroy_process_group()


if (__name__ == '__main__'):
  main()
