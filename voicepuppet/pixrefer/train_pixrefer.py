#!/usr/bin/env python
# -*- encoding: utf-8 -*-
"""Launcher kept at the reference's path so its command line works unchanged:
    python voicepuppet/pixrefer/train_pixrefer.py --config_path config/params.yml ...
The implementation lives in voicepuppet_amd/pixrefer/train_pixrefer.py."""
import os
import sys

sys.path.append(os.getcwd())

from voicepuppet_amd.pixrefer.train_pixrefer import main

if (__name__ == '__main__'):
  main()
