"""Forward value of the reference's training losses (``tensortools/losses.py``) on the MI355X.
Only the forward is provided: the scoring path never back-propagates; these exist so that a
validation / monitoring loop can report the same numbers without TensorFlow.
"""
import numpy as np

from .. import _lib


def masked_softmax_cross_entropy(labels, logits, mask, num_classes, weight=0.0, label_smoothing=0.0,
                                 scope="XEntropy"):
    """reference tensortools/losses.py:3-74: softmax cross entropy with label smoothing, masked,
    optionally ENet-weighted (``weight > 1``); returns a float64 scalar tensor on the GPU."""
    torch = _lib.require_gpu()
    x = _lib.as_device_f32(logits)
    if x.dim() != 4 or x.shape[-1] != num_classes:
        raise ValueError("logits must be [N,H,W,%d] (got %s)" % (num_classes, tuple(x.shape)))
    n, h, w, k = x.shape
    lab = labels if isinstance(labels, torch.Tensor) else torch.as_tensor(np.asarray(labels))
    lab = lab.to(device=x.device, dtype=torch.uint8).reshape(n, h, w).contiguous()
    mk = mask if isinstance(mask, torch.Tensor) else torch.as_tensor(np.asarray(mask))
    mk = mk.to(device=x.device, dtype=torch.float32).reshape(n, h, w).contiguous()
    L = _lib.lib()
    with torch.cuda.device(x.device):
        ws = torch.empty(int(L.ssal_xent_workspace_bytes(h, w)), dtype=torch.uint8, device=x.device)
        out = torch.empty((1,), dtype=torch.float64, device=x.device)
        _lib.check(L.ssal_masked_softmax_cross_entropy(
            _lib.dev_ptr(x), _lib.dev_ptr(lab), _lib.dev_ptr(mk), n, h, w, k, float(weight),
            float(label_smoothing), _lib.dev_ptr(out), _lib.dev_ptr(ws), ws.numel(), _lib.stream_ptr()))
    return out[0]


def L2_regularization(kernels, weight, scope=None):
    """weight / len(kernels) * sum_k tf.nn.l2_loss(k) = sum(k**2) / 2   (reference :159-179);
    host-side: the kernels are the model's (host) weight arrays."""
    total = 0.0
    for k in kernels:
        a = np.asarray(k.numpy() if hasattr(k, "numpy") else k, dtype=np.float32)
        total += float(np.sum(a.astype(np.float64) ** 2) / 2.0)
    return total * (weight / float(len(kernels)))


__all__ = ["masked_softmax_cross_entropy", "L2_regularization"]
