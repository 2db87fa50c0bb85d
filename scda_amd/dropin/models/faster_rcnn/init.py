"""weight initialisers named as in models/faster_rcnn/init.py"""
import numpy as np
import torch.nn.init as init


def gaussian_weights_init(m):
    # every module whose class name STARTS with 'Conv' gets N(0, 0.02) weights (init.py:8-12)
    if m.__class__.__name__.find('Conv') == 0:
        m.weight.data.normal_(0.0, 0.02)


def xavier_weights_init(m):
    if m.__class__.__name__.find('Conv') != -1:
        init.xavier_uniform_(m.weight, gain=np.sqrt(2))
        init.constant_(m.bias, 0.1)
