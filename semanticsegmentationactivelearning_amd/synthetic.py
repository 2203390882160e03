"""Seeded synthetic weights and Cityscapes-shaped frames.

There is no network, dataset or checkpoint on the GPU box, so benchmarks and parity tests run on
synthetic data:

* weights: ``default_rng(seed)``; conv kernels glorot-uniform (TF fan rules), batch-norm statistics
  randomised so BN is not the identity, PReLU slopes U(0.1, 0.4) (per channel, so that a channel
  indexing bug cannot hide behind a constant), ``Final`` kernel scaled by ``final_gain`` so the
  softmax is not near-uniform and per-image scores separate.
* frames: frame ``f`` of a pool is a pure function of ``(seed, f)`` built from a counter-based
  hash (splitmix64), so the SAME uint8 stream is produced on the host (here, numpy) and on the
  device (``ssal_synth_frames_nhwc``): coarse 8x8 colour blocks + fine noise in [-16,16], scaled by
  a per-frame brightness in [0.2,1.0], clipped to uint8, then ``x = u8 * float32(1/255)``
  (reference tensortools/input.py:289-290 ``convert_image_dtype``).
"""
import numpy as np

from . import _lib

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    """vectorised splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def synth_frame_u8(frame, h, w, c, seed=0):
    """uint8 [h,w,c] frame ``frame`` of the synthetic pool (host twin of k_synth_frames)."""
    if h % 8 or w % 8:
        raise ValueError("H and W must be divisible by 8")
    with np.errstate(over="ignore"):
        fk = _splitmix64(np.array([np.uint64(seed) ^ (np.uint64(frame) * np.uint64(0xD1342543DE82EF95))],
                                  dtype=np.uint64))[0]
        u = np.float32(_splitmix64(np.array([fk ^ np.uint64(0xB5)], dtype=np.uint64))[0] >> np.uint64(40)) \
            * np.float32(1.0 / 16777216.0)
        bright = np.float32(0.2) + np.float32(0.8) * u
        yy, xx, cc = np.meshgrid(np.arange(h, dtype=np.uint64), np.arange(w, dtype=np.uint64),
                                 np.arange(c, dtype=np.uint64), indexing="ij")
        r = (yy * np.uint64(w) + xx) * np.uint64(c) + cc
        ic = ((yy // np.uint64(8)) * np.uint64(w // 8) + (xx // np.uint64(8))) * np.uint64(c) + cc
        coarse = (_splitmix64(fk + np.uint64(2) * ic) >> np.uint64(56)).astype(np.int32)
        fine = ((_splitmix64(fk + np.uint64(2) * r + np.uint64(1)) >> np.uint64(32)) % np.uint64(33)).astype(np.int32) - 16
    v = (coarse + fine).astype(np.float32) * bright
    v = np.minimum(np.maximum(v, np.float32(0.0)), np.float32(255.0))
    return v.astype(np.uint8)  # truncation, like the device cast


def u8_to_f32(u8):
    """tf.image.convert_image_dtype(uint8 -> float32): x * (1/255) (tensortools/input.py:289-290)."""
    return u8.astype(np.float32) * np.float32(1.0 / 255.0)


def synth_frames_f32(frames, h, w, c, seed=0):
    """float32 [len(frames),h,w,c] host batch."""
    return np.stack([u8_to_f32(synth_frame_u8(int(f), h, w, c, seed)) for f in frames])


def synth_frames_device(first_frame, count, h, w, c, seed=0, out=None, device=None, dtype=None):
    """Generate ``count`` consecutive frames directly in HBM: float32 NHWC (= uint8 * 1/255, the default) or,
    with ``dtype=torch.uint8`` / a uint8 ``out``, the decoded uint8 frames themselves."""
    torch = _lib.require_gpu()
    if out is None:
        out = torch.empty((count, h, w, c), dtype=dtype if dtype is not None else torch.float32,
                          device=device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    if tuple(out.shape) != (count, h, w, c):
        raise ValueError("out has shape %s, expected %s" % (tuple(out.shape), (count, h, w, c)))
    if out.dtype not in (torch.float32, torch.uint8):
        raise ValueError("frames are float32 or uint8 (got %s)" % out.dtype)
    with torch.cuda.device(out.device):
        gen = _lib.lib().ssal_synth_frames_nhwc_u8 if out.dtype == torch.uint8 else _lib.lib().ssal_synth_frames_nhwc
        _lib.check(gen(int(seed), int(first_frame), count, h, w, c, _lib.dev_ptr(out, out.dtype, "out"),
                       _lib.stream_ptr()))
    return out


# ------------------------------------------------------------------------------------------------
# weights
# ------------------------------------------------------------------------------------------------
def _glorot(rng, shape):
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def randomize_enet(model, seed=0, final_gain=8.0):
    """Fill every variable of a built ``models.ENet`` with the seeded synthetic recipe, layer by layer
    in ``add_weight`` creation order (deterministic for a given architecture)."""
    rng = np.random.default_rng(seed)
    for layer in model.layers:
        for var in layer.creation_order_variables:
            leaf = var.name.rsplit("/", 1)[-1]
            shape = var.shape
            if leaf.startswith("Kernel"):
                val = _glorot(rng, shape)
                if layer.name == "Final":
                    val = val * np.float32(final_gain)
            elif leaf == "Alpha":
                val = rng.uniform(0.1, 0.4, size=shape).astype(np.float32)
            elif leaf == "Mean":
                val = rng.normal(0.0, 0.1, size=shape).astype(np.float32)
            elif leaf == "Variance":
                val = rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
            elif leaf == "Gamma":
                val = rng.uniform(0.8, 1.2, size=shape).astype(np.float32)
            elif leaf == "Beta":
                val = rng.normal(0.0, 0.1, size=shape).astype(np.float32)
            else:
                raise RuntimeError("unexpected variable %s" % var.name)
            var.assign(val)
    return model


def enet_params_dict(model):
    """{"<Layer>.<attr>": float32 ndarray} in the C-ABI naming (what the parity oracle consumes)."""
    out = {}
    for layer in model.layers:
        for attr, var in layer.abi_tensors().items():
            out["%s.%s" % (layer.name, attr)] = var.numpy()
    return out


def randomize_icnet(model, seed=0, cls_gain=1.0):
    """Seeded synthetic weights for a built ``models.ICNet`` (layer by layer, ``add_weight`` order): kernels
    He-uniform (ReLU network; the ``*_1x1_increase`` convs at a third of that so the 16 residual bottlenecks do not
    blow the activations up), batch-norm statistics randomised, classifier scaled by ``cls_gain`` so the softmax is
    not near-uniform and per-image scores separate."""
    rng = np.random.default_rng(seed)
    for layer in model.layers:
        for var in layer.creation_order_variables:
            leaf = var.name.rsplit("/", 1)[-1]
            shape = var.shape
            if leaf == "Kernel":
                lim = np.sqrt(6.0 / (shape[0] * shape[1] * shape[2]))
                val = rng.uniform(-lim, lim, size=shape).astype(np.float32)
                if layer.name.endswith("_1x1_increase"):
                    val = val * np.float32(1.0 / 3.0)
                if layer.name == "conv6_cls":
                    val = val * np.float32(cls_gain)
            elif leaf == "Bias":
                val = rng.normal(0.0, 0.1, size=shape).astype(np.float32)
            elif leaf == "Mean":
                val = rng.normal(0.0, 0.1, size=shape).astype(np.float32)
            elif leaf == "Variance":
                val = rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
            elif leaf == "Gamma":
                val = rng.uniform(0.8, 1.2, size=shape).astype(np.float32)
            elif leaf == "Beta":
                val = rng.normal(0.0, 0.1, size=shape).astype(np.float32)
            else:
                raise RuntimeError("unexpected variable %s" % var.name)
            var.assign(val)
    return model


def icnet_params_dict(model):
    """{"<layer>.<attr>": float32 ndarray} in the C-ABI naming (what the parity oracle consumes)"""
    return enet_params_dict(model)
