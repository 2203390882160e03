"""Mirror of the reference's ``models/util/extra_ops.py`` (xops): prelu, unpool_2d, batch_norm,
spatial_dropout -- running the stand-alone HIP operators of libssal_hip.so on GPU-resident torch
tensors (NHWC fp32).  No CPU fallback.
"""
from ... import _lib


def prelu(x, alpha, name="PReLU"):
    """relu(x) - alpha * relu(-x), alpha per last-dim channel (reference extra_ops.py:9-26)."""
    torch = _lib.require_gpu()
    x = _lib.as_device_f32(x)
    a = _lib.as_device_f32(alpha)
    c = x.shape[-1]
    if a.numel() != c:
        raise ValueError("alpha must match the channel depth (%d), got %d" % (c, a.numel()))
    y = torch.empty_like(x)
    _lib.check(_lib.lib().ssal_prelu(_lib.dev_ptr(x), x.numel() // c, c, _lib.dev_ptr(a),
                                     _lib.dev_ptr(y), _lib.stream_ptr()))
    return y


def max_pool_with_argmax(x, include_batch_in_index=False):
    """tf.nn.max_pool_with_argmax(ksize 2x2, strides 2, SAME, Targmax=int64) as used by
    BottleneckDownsample (reference enet_modules.py:927-929); first maximum wins ties."""
    torch = _lib.require_gpu()
    x = _lib.as_device_f32(x)
    n, h, w, c = x.shape
    y = torch.empty((n, h // 2, w // 2, c), dtype=torch.float32, device=x.device)
    idx = torch.empty((n, h // 2, w // 2, c), dtype=torch.int64, device=x.device)
    _lib.check(_lib.lib().ssal_max_pool_with_argmax_2x2(
        _lib.dev_ptr(x), n, h, w, c, _lib.dev_ptr(y), _lib.dev_ptr(idx),
        1 if include_batch_in_index else 0, _lib.stream_ptr()))
    return y, idx


def unpool_2d(inputs, idx, strides=[1, 2, 2, 1], name="Unpool2D", idx_has_batch=False):
    """Scatter ``inputs`` into zeros([N,2H,2W,C]) at the flattened argmax positions
    (reference extra_ops.py:28-86).  ``idx_has_batch=False`` is the reference's GPU branch
    (per-image indices, batch offset added here, :73-79); True is its CPU branch (:80-81)."""
    torch = _lib.require_gpu()
    if list(strides) != [1, 2, 2, 1]:
        raise NotImplementedError("unpool_2d: strides [1,2,2,1] only")
    x = _lib.as_device_f32(inputs)
    n, h, w, c = x.shape
    if not isinstance(idx, torch.Tensor):
        idx = torch.as_tensor(idx)
    idx = idx.to(device=x.device, dtype=torch.int64).contiguous()
    if tuple(idx.shape) != tuple(x.shape):
        raise ValueError("idx shape %s != inputs shape %s" % (tuple(idx.shape), tuple(x.shape)))
    y = torch.empty((n, 2 * h, 2 * w, c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ssal_unpool_2d(_lib.dev_ptr(x), _lib.dev_ptr(idx), n, h, w, c,
                                         1 if idx_has_batch else 0, _lib.dev_ptr(y), _lib.stream_ptr()))
    return y


def spatial_dropout(inputs, drop_rate, name="SpatialDropout", seed=0):
    """Channel-wise (whole feature-plane) dropout: ``tf.nn.dropout(inputs, rate, noise_shape=[N,1,1,C])``
    (reference extra_ops.py:137-151; used by the bottlenecks only when ``training`` -- enet_modules.py:591-594 --
    so it is the identity on the scoring path).  ``y = (x / (1 - rate)) * keep[n, c]``.  The keep draw is a seeded
    counter-based hash (splitmix64 of ``seed ^ (n*C + c) * K``, top 24 bits; restated for the tests in
    ``oracle/dropout_oracle.py``); TensorFlow's own random stream cannot be reproduced."""
    torch = _lib.require_gpu()
    if not 0.0 <= float(drop_rate) < 1.0:
        raise ValueError("drop_rate must be in [0, 1)")
    x = _lib.as_device_f32(inputs)
    if x.dim() != 4:
        raise ValueError("inputs must be NHWC rank-4")
    n, h, w, c = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.lib().ssal_spatial_dropout(_lib.dev_ptr(x), n, h * w, c, float(drop_rate), int(seed),
                                               _lib.dev_ptr(y), _lib.stream_ptr()))
    return y


def batch_norm(inputs, mean, var, gamma, beta, training=True, decay=0.9):
    """tf.nn.fused_batch_norm wrapper (reference extra_ops.py:154-185), inference branch only:
    (x - mean) * rsqrt(var + 1e-3) * gamma + beta, folded to one fma per element.
    Returns (out, None, None) like the reference's non-training branch."""
    torch = _lib.require_gpu()
    if training:
        raise NotImplementedError("batch_norm(training=True) is outside the MI355X scoring path")
    x = _lib.as_device_f32(inputs)
    c = x.shape[-1]
    m, v, g, b = (_lib.as_device_f32(t) for t in (mean, var, gamma, beta))
    for t in (m, v, g, b):
        if t.numel() != c:
            raise ValueError("batch-norm statistics must have %d elements" % c)
    y = torch.empty_like(x)
    _lib.check(_lib.lib().ssal_batch_norm_inference(
        _lib.dev_ptr(x), x.numel() // c, c, _lib.dev_ptr(m), _lib.dev_ptr(v), _lib.dev_ptr(g),
        _lib.dev_ptr(b), _lib.dev_ptr(y), _lib.stream_ptr()))
    return y, None, None
