#!/usr/bin/env python
# -*- encoding: utf-8 -*-
"""Launcher kept at the reference's path so its command line works unchanged:
    python voicepuppet/bfmnet/train_bfmnet.py --config_path config/params.yml
The implementation lives in voicepuppet_amd/bfmnet/train_bfmnet.py."""
import os
import sys

sys.path.append(os.getcwd())

from voicepuppet_amd.bfmnet.train_bfmnet import main

if (__name__ == '__main__'):
  main()
