"""reference e2enet/utilities/to_torch.py:18-31"""
import torch


def maybe_to_torch(d):
    if isinstance(d, list):
        return [maybe_to_torch(i) if not isinstance(i, torch.Tensor) else i for i in d]
    if not isinstance(d, torch.Tensor):
        return torch.from_numpy(d).float()
    return d


def to_cuda(data, non_blocking=True, gpu_id=0):
    if isinstance(data, list):
        return [i.cuda(gpu_id, non_blocking=non_blocking) for i in data]
    return data.cuda(gpu_id, non_blocking=non_blocking)
