import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
  config.addinivalue_line("markers", "slow: CPU test that takes tens of seconds")


def pytest_collection_modifyitems(config, items):
  # No test of this suite runs for minutes: a per-test ceiling (pytest-timeout, when installed) turns a wedged rendezvous or a device
  # that stops answering into a failure with a traceback instead of a run that sits until the driver's own limit.
  if config.pluginmanager.hasplugin("timeout"):
    for item in items:
      if item.get_closest_marker("timeout") is None:
        item.add_marker(pytest.mark.timeout(900))
  # GPU tests are selected explicitly with `-m gpu`; skip them when no device is visible.
  import torch
  if torch.cuda.is_available():
    return
  skip = pytest.mark.skip(reason="no GPU visible")
  for item in items:
    if "gpu" in item.keywords:
      item.add_marker(skip)
