"""Precision helpers.  The reference's fp16 loss-scaling machinery (fp16_util.py) is out of scope;
`zero_grad` is kept because callers use it."""


def zero_grad(model_params):
    for p in model_params:
        if p.grad is not None:
            p.grad.detach_()
            p.grad.zero_()
