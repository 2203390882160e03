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


def resize_nearest_neighbor(x, size):
    """tf.image.resize_nearest_neighbor with the TF-1.13 defaults (align_corners=False):
    src = floor(dst * in / out), clamped; x is [N,H,W] or [N,H,W,C] on the device (index plumbing,
    no arithmetic on the values)"""
    torch = _lib.require_gpu()
    h, w = int(x.shape[1]), int(x.shape[2])
    oh, ow = int(size[0]), int(size[1])
    # float32 scale and product, as the TF kernel computes them
    sy, sx = np.float32(h) / np.float32(oh), np.float32(w) / np.float32(ow)
    iy = np.minimum(np.floor(np.arange(oh, dtype=np.float32) * sy).astype(np.int64), h - 1)
    ix = np.minimum(np.floor(np.arange(ow, dtype=np.float32) * sx).astype(np.int64), w - 1)
    iy = torch.as_tensor(iy, device=x.device)
    ix = torch.as_tensor(ix, device=x.device)
    return x.index_select(1, iy).index_select(2, ix).contiguous()


def _conv1x1(x, kernel):
    """tf.nn.conv2d(x, kernel [1,1,C,K], strides 1, "VALID") through the C ABI"""
    torch = _lib.require_gpu()
    n, h, w, c = x.shape
    kernel = np.ascontiguousarray(kernel, dtype=np.float32)
    if kernel.ndim != 4 or tuple(kernel.shape[:3]) != (1, 1, c):
        raise ValueError("kernel must be [1,1,%d,K] (got %s)" % (c, tuple(kernel.shape)))
    k = kernel.shape[3]
    k2 = 1 << max(0, (k - 1).bit_length())  # the generic conv kernel wants a power-of-two output width:
    padded = np.zeros((1, 1, c, k2), dtype=np.float32)  # zero columns are independent outputs, dropped below
    padded[..., :k] = kernel
    kd = _lib.as_device_f32(padded).to(x.device)
    y = torch.empty((n, h, w, k2), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ssal_conv2d_same(_lib.dev_ptr(x), n, h, w, c, _lib.dev_ptr(kd), 1, 1, k2, 1, 1,
                                               _lib.dev_ptr(y), _lib.stream_ptr()))
    return y[..., :k].contiguous()


def multiscale_masked_softmax_cross_entropy(labels, logits, mask, num_classes, weight=0.0, label_smoothing=0.0,
                                            normalize=False, scope="MultiXEntropy", kernels=None, seed=0):
    """reference tensortools/losses.py:76-157 (forward value): ``logits`` is a list in decrementing scale
    (e.g. one entry of ``ENet.endpoint_outputs``); entry 0 are the class logits, every further entry gets a
    1x1 prediction head ``Kernel_<i>`` [1,1,C,num_classes] and is scored against the nearest-neighbour
    resized labels / mask.  Returns ``(loss, kernels)`` like the reference (which returns its head
    variables so that they can be saved).

    ``kernels``: the head kernels (list of [1,1,C,K] arrays); when None they are drawn glorot-uniform from
    ``seed`` -- the default initializer of ``tf.get_variable`` -- since there is no variable store here.
    ``normalize=True`` raises TypeError exactly as the reference does (it evaluates ``len()`` of a scalar
    tensor, ``losses.py:152-153``)."""
    torch = _lib.require_gpu()
    if normalize:
        raise TypeError("object of type 'Tensor' has no len()  (reference losses.py:153: `loss / len(loss)`)")
    lg0 = _lib.as_device_f32(logits[0])
    n, h, w, _ = lg0.shape
    lab = labels if isinstance(labels, torch.Tensor) else torch.as_tensor(np.asarray(labels))
    lab = lab.to(device=lg0.device, dtype=torch.uint8).reshape(n, h, w)
    mk = mask if isinstance(mask, torch.Tensor) else torch.as_tensor(np.asarray(mask))
    mk = mk.to(device=lg0.device, dtype=torch.float32).reshape(n, h, w)
    losses = [masked_softmax_cross_entropy(lab, lg0, mk, num_classes, weight, label_smoothing)]
    rng = np.random.default_rng(seed)
    heads = []
    for i, feat in enumerate(logits[1:]):
        f = _lib.as_device_f32(feat)
        c = int(f.shape[-1])
        if kernels is not None:
            krnl = np.asarray(kernels[i], dtype=np.float32)
        else:
            lim = np.sqrt(6.0 / (c + num_classes))
            krnl = rng.uniform(-lim, lim, size=(1, 1, c, num_classes)).astype(np.float32)
        heads.append(krnl)
        head_logits = _conv1x1(f, krnl)
        size = head_logits.shape[1:3]
        losses.append(masked_softmax_cross_entropy(resize_nearest_neighbor(lab, size), head_logits,
                                                   resize_nearest_neighbor(mk, size), num_classes, weight,
                                                   label_smoothing))
    total = losses[0]
    for extra in losses[1:]:
        total = total + extra
    return total, heads


def L2_regularization(kernels, weight, scope=None):
    """weight / len(kernels) * sum_k tf.nn.l2_loss(k) = sum(k**2) / 2   (reference :159-179);
    host-side: the kernels are the model's (host) weight arrays."""
    total = 0.0
    for k in kernels:
        a = np.asarray(k.numpy() if hasattr(k, "numpy") else k, dtype=np.float32)
        total += float(np.sum(a.astype(np.float64) ** 2) / 2.0)
    return total * (weight / float(len(kernels)))


__all__ = ["masked_softmax_cross_entropy", "multiscale_masked_softmax_cross_entropy", "resize_nearest_neighbor",
           "L2_regularization"]
