#!/usr/bin/env python
# -*- encoding: utf-8 -*-
"""BASELINE config 5: many clips through infer_bfmvid, sharded over the GPUs of one node - one process per GPU, clips dealt
round-robin, NO collective (every clip is independent: audio -> BFM coefficients -> raster -> PixReferNet).

    python -m voicepuppet_amd.pixrefer.infer_clips --config_path config/params.yml --gpus 8 clips.txt

clips.txt: one clip per line, `<image 1536x512> <audio.wav> [<bfmcoeff.npz>]` (paths relative to the working directory, which
must be the one infer_bfmvid.py runs from: it holds ckpt_bfmnet/, ckpt_pixrefer/, BFM/).  Clip i is written to
<out_root>/clip_<i>/<frame>.jpg (+ <out_root>/clip_<i>.mp4 when ffmpeg exists).  The reference handles one clip per invocation
(voicepuppet/pixrefer/infer_bfmvid.py:231-243); this launcher is that loop, spread over ranks.

The parent never touches the GPU: it starts the ranks as child processes (RANK / LOCAL_RANK / WORLD_SIZE in their environment)
and exits with the worst child status.
"""
import logging
import os
import subprocess
import sys
from optparse import OptionParser

logger = logging.getLogger(__name__)


def read_clip_list(path):
  clips = []
  for line in open(path):
    parts = line.split('#', 1)[0].split()
    if not parts:
      continue
    if len(parts) not in (2, 3):
      raise ValueError('%s: expected "<image> <audio> [<bfmcoeff.npz>]", got %r' % (path, line.strip()))
    clips.append(tuple(parts))
  return clips


def rank_commands(opts, clip_list, world):
  """[(argv, env)] of the child ranks."""
  out = []
  for r in range(world):
    argv = [sys.executable, '-m', 'voicepuppet_amd.pixrefer.infer_clips', '--config_path', opts.config_path,
            '--frame_batch', str(opts.frame_batch), '--out_root', opts.out_root, '--gpus', str(world), clip_list]
    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    out.append((argv, env))
  return out


def run_rank(opts, clips, rank, world):
  import torch
  from voicepuppet_amd.parallel import shard_round_robin
  from voicepuppet_amd.pixrefer import infer_bfmvid
  torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', rank)) % max(1, torch.cuda.device_count()))
  mine = shard_round_robin(len(clips), rank, world)
  logger.info('rank %d/%d: clips %s', rank, world, mine)
  for i in mine:
    clip = clips[i]
    argv = ['--config_path', opts.config_path, '--frame_batch', str(opts.frame_batch),
            '--output_dir', os.path.join(opts.out_root, 'clip_%d' % i)]
    if len(clip) == 3:
      argv += ['--bfmcoeff', clip[2]]
    infer_bfmvid.main(argv + [clip[0], clip[1]])
  return len(mine)


def main(argv=None):
  cmd_parser = OptionParser(usage="usage: %prog [options] --config_path <> clips.txt")
  cmd_parser.add_option('--config_path', type="string", dest="config_path", help='the config yaml file')
  cmd_parser.add_option('--gpus', type="int", dest="gpus", default=1, help='ranks (one per GPU)')
  cmd_parser.add_option('--frame_batch', type="int", dest="frame_batch", default=8, help='frames per device batch')
  cmd_parser.add_option('--out_root', type="string", dest="out_root", default='output_clips', help='clip_<i>/ directories go here')
  opts, args = cmd_parser.parse_args(argv)
  if (opts.config_path is None or len(args) != 1):
    logger.error('Please check your parameters.')
    exit(0)
  if (not os.path.exists(opts.config_path)):
    logger.error('config_path not exists')
    exit(0)
  clips = read_clip_list(args[0])
  world = max(1, opts.gpus)
  if 'WORLD_SIZE' in os.environ or world == 1:
    rank = int(os.environ.get('RANK', '0'))
    if int(os.environ.get('WORLD_SIZE', '1')) != world:
      logger.error('--gpus %d but WORLD_SIZE=%s', world, os.environ.get('WORLD_SIZE'))
      sys.exit(2)
    run_rank(opts, clips, rank, world)
    return 0
  procs = [subprocess.Popen(a, env=e) for a, e in rank_commands(opts, args[0], world)]
  rc = 0
  for p in procs:
    rc = max(rc, abs(p.wait()))
  if rc:
    sys.exit(rc)
  return 0


if (__name__ == '__main__'):
  logging.basicConfig(level=logging.INFO, format='%(asctime)s - %(name)s - %(levelname)s - %(message)s')
  main()
