"""reference e2enet/training/learning_rate/poly_lr.py:16-17"""


def poly_lr(epoch, max_epochs, initial_lr, exponent=0.9):
    return initial_lr * (1 - epoch / max_epochs) ** exponent
