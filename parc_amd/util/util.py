"""Seeding of every random source the tracker draws from (python, numpy, torch host + all visible devices); the entry point
the reference's run.py calls per rank (util/util.py:5-11)."""
import random

import numpy as np
import torch


def set_rand_seed(seed):
    s = int(seed)
    for seeder, value in ((random.seed, s), (np.random.seed, s % (1 << 32)), (torch.manual_seed, s)):
        seeder(value)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(s)
