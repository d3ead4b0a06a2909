"""bench.py pieces that run without a GPU: the optional sysfs sampler degrades to nothing when the hwmon files are absent, and
the synthetic problem builders are deterministic (the GPU tests and the oracle legs rely on identical inputs)."""
import os
import sys

import numpy as np
import torch

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_device_sampler_without_hwmon_files_reports_nothing():
    import bench
    with bench.DeviceSampler("0000:ff:1f.7") as s:          # no such card: falls back to every card, here none with hwmon files
        s.read_once()
    assert s.summary() is None or isinstance(s.summary(), dict)
    s2 = bench.DeviceSampler(None, period=None)
    s2.read_once()
    assert s2.summary() is None or "cards_sampled" in s2.summary()


def test_synthetic_inputs_are_deterministic():
    import bench
    ue1, ud1 = bench.pems_like_graph(64, 71, seed=0)
    ue2, ud2 = bench.pems_like_graph(64, 71, seed=0)
    assert torch.equal(ue1, ue2) and torch.equal(ud1, ud2)
    assert ue1.shape == (142, 2)                               # both directions of 71 undirected edges
    y1 = bench.synth_y(64, 5, 12, seed=1, offset=3, device=torch.device("cpu"))
    y2 = bench.synth_y(64, 5, 12, seed=1, offset=3, device=torch.device("cpu"))
    assert y1.shape == (5, 12, 64, 1) and torch.equal(y1, y2)
    assert not torch.equal(y1, bench.synth_y(64, 5, 12, seed=1, offset=4, device=torch.device("cpu")))
