#!/usr/bin/env python
# -*- encoding: utf-8 -*-
"""BFMNet training entry point, same CLI as the reference (voicepuppet/bfmnet/train_bfmnet.py):

    python voicepuppet/bfmnet/train_bfmnet.py --config_path config/params.yml

Single GPU, as the reference (it pins CUDA_VISIBLE_DEVICES to one device).  Extra, optional flags (defaults reproduce the reference
run): --steps, --batch_size, --eval_step, --save_step.  The evaluation plot of the reference (plot_bfm_coeff_seq, a matplotlib
figure of the mesh) is outside the path; the evaluation loss is printed.
"""
import logging
import os
import sys
import time
from optparse import OptionParser

sys.path.append(os.getcwd())

from voicepuppet_amd.bfmnet.bfmnet import BFMNet
from voicepuppet_amd.generator.generator import BFMNetDataGenerator
from voicepuppet_amd.runtime import Session

logging.basicConfig(level=logging.INFO, format='%(asctime)s - %(name)s - %(levelname)s - %(message)s')
logger = logging.getLogger(__name__)


def mkdir(path):
  if not os.path.exists(path):
    os.makedirs(path)


def main(argv=None):
  cmd_parser = OptionParser(usage="usage: %prog [options] --config_path <>")
  cmd_parser.add_option('--config_path', type="string", dest="config_path", help='the config yaml file')
  cmd_parser.add_option('--steps', type="int", dest="steps", default=None, help='iterations to run (default: training.epochs)')
  cmd_parser.add_option('--batch_size', type="int", dest="batch_size", default=4, help='clips per step (reference: 4)')
  cmd_parser.add_option('--eval_step', type="int", dest="eval_step", default=1000)
  cmd_parser.add_option('--save_step', type="int", dest="save_step", default=5000)
  opts, _ = cmd_parser.parse_args(argv)

  if (opts.config_path is None):
    logger.error('Please check your parameters.')
    exit(0)
  config_path = opts.config_path
  if (not os.path.exists(config_path)):
    logger.error('config_path not exists')
    exit(0)

  batch_size = opts.batch_size
  ### Generator for training setting
  train_generator = BFMNetDataGenerator(config_path)
  params = train_generator.params
  params.dataset_path = params.train_dataset_path
  params.batch_size = batch_size
  train_generator.set_params(params)
  train_dataset = train_generator.get_dataset()

  ### Generator for evaluation setting
  eval_generator = BFMNetDataGenerator(config_path)
  params = eval_generator.params
  params.dataset_path = params.eval_dataset_path
  params.batch_size = batch_size
  eval_generator.set_params(params)
  eval_dataset = eval_generator.get_dataset()

  sess = Session()
  train_iter = train_dataset.make_one_shot_iterator()
  eval_iter = eval_dataset.make_one_shot_iterator()

  ### BFMNet setting
  bfmnet = BFMNet(config_path)
  params = bfmnet.params
  epochs = opts.steps if opts.steps is not None else params.training['epochs']
  params.add_hparam('max_to_keep', 10)
  params.add_hparam('save_dir', 'ckpt_bfmnet')
  params.add_hparam('save_name', 'bfmnet')
  params.add_hparam('save_step', opts.save_step)
  params.add_hparam('eval_step', opts.eval_step)
  params.batch_size = batch_size
  bfmnet.set_params(params)

  mkdir(params.save_dir)

  train_nodes = bfmnet.build_train_op(*train_iter.get_next())
  eval_nodes = bfmnet.build_eval_op(*eval_iter.get_next())

  # Restore from save_dir
  if ('checkpoint' in os.listdir(params.save_dir)):
    print('Restore from {}\n'.format(params.save_dir))
    bfmnet.restore(params.save_dir)

  saved = []
  t0 = time.time()
  for i in range(epochs):
    ### Run training
    _, loss, lr, global_step = sess.run([train_nodes['Train_op'], train_nodes['Loss'], train_nodes['Lr'], train_nodes['Global_step']])
    print('Step {}: Loss= {:.3f}, Lr= {:.2e}'.format(global_step, loss, lr))

    ### Run evaluation
    if (global_step % params.eval_step == 0):
      loss, seq_len, real_bfm_coeff_seq, bfm_coeff_seq = sess.run([eval_nodes['Loss'], eval_nodes['Seq_len'], eval_nodes['BFM_coeff_seq'],
                                                                   eval_nodes['BFMCoeffDecoder']])
      print('\r\nEvaluation >>> Loss= {:.3f}'.format(loss))
      logger.info('%.1f clips/s', (i + 1) * batch_size / (time.time() - t0))

    ### Save checkpoint
    if (global_step % params.save_step == 0):
      path = bfmnet.save(os.path.join(params.save_dir, '%s-%d' % (params.save_name, global_step)))
      saved.append(path)
      while len(saved) > params.max_to_keep:
        old = saved.pop(0)
        for suffix in ('.index', '.data-00000-of-00001'):
          if os.path.exists(old + suffix):
            os.remove(old + suffix)


if (__name__ == '__main__'):
  main()
