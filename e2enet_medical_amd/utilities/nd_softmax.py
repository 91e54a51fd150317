"""reference e2enet/utilities/nd_softmax.py:22-23"""
import torch.nn.functional as F


def softmax_helper(x):
    return F.softmax(x, 1)
