"""Seeding helper the training scripts import (reference: training_utils.py:8-11; the TF seed becomes the torch seed)."""
import random

import numpy as np
import torch


def initialize_random_seed(seed):
    np.random.seed(seed)
    torch.manual_seed(seed)
    random.seed(seed)
