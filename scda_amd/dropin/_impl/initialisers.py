"""The two weight initialisers the GAN-side modules `.apply()`: names as in the reference's models/faster_rcnn/init.py.

gaussian_weights_init: modules whose CLASS NAME begins with "Conv" (Conv2d, ConvTranspose2d, this package's Conv2d /
ConvTranspose1x1 -- but not e.g. LeakyReLUConv2d wrappers themselves) get N(0, 0.02) weights; biases are left alone.
xavier_weights_init: any class name containing "Conv": Xavier-uniform weights with gain sqrt(2), biases 0.1."""
import math

from torch.nn import init


def _class_name(module):
    return type(module).__name__


def gaussian_weights_init(m):
    if _class_name(m).startswith('Conv'):
        m.weight.data.normal_(0.0, 0.02)


def xavier_weights_init(m):
    if 'Conv' in _class_name(m):
        init.xavier_uniform_(m.weight, gain=math.sqrt(2.0))
        init.constant_(m.bias, 0.1)
