"""Seeding helper (mirror of the reference's util/util.py:5-11)."""
import random

import numpy as np
import torch


def set_rand_seed(seed):
    seed = int(seed)
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
