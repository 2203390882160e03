"""Test-split inference: host-side mirror of the reference's ``inference.py:61-153`` on the MI355X path.

logits = net(image, training=False) -> optional ``tf.image.resize_bilinear(logits, size)`` (the
reference resizes the LOGITS, :96-99; TF-1.13 legacy mapping) -> argmax (first maximum) -> reverse
embedding trainId -> dataset id (:101-106) or colour map (:107-109) -> PNG (:110-119).
Dataset tables (``embedding_reversed``, ``colormap``) are passed in by the caller (the reference's
``datasets`` package is out of scope).
"""
import os

import numpy as np

from . import _lib
from . import active_learning as al


def resize_bilinear(x, size):
    """tf.image.resize_bilinear(x, size) with TF-1.13 defaults (align_corners=False, src = dst*in/out)."""
    torch = _lib.require_gpu()
    x = _lib.as_device_f32(x)
    n, h, w, c = x.shape
    oh, ow = int(size[0]), int(size[1])
    y = torch.empty((n, oh, ow, c), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ssal_resize_bilinear(_lib.dev_ptr(x), n, h, w, c, oh, ow, _lib.dev_ptr(y),
                                                   _lib.stream_ptr()))
    return y


def predict_labels(net, images, size=None):
    """uint8 train-id map [N,H',W'] on the GPU (reference :95-99: argmax of the (resized) logits)"""
    logits = net(images, training=False)
    if size is not None:
        logits = resize_bilinear(logits, size)
    _, extra = al.score_logits(logits, "confidence", return_label=True)
    return extra["label"]


def reverse_embedding(pred, embedding_reversed):
    """trainId -> dataset id through a 256-entry table (reference :101-106, tf.gather_nd)"""
    torch = _lib.require_gpu()
    lut = torch.as_tensor(np.asarray(embedding_reversed, dtype=np.uint8), device=pred.device)
    return lut[pred.long()]


def colorize(pred, colormap):
    """trainId -> RGB through a [256,3] table (reference :107-109)"""
    torch = _lib.require_gpu()
    lut = torch.as_tensor(np.asarray(colormap, dtype=np.uint8), device=pred.device)
    return lut[pred.long()]


def write_png(path, array):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(array)).save(path, format="PNG")


def run_inference(net, batches, output_dir, embedding_reversed=None, colormap=None, size=None):
    """``batches`` yields (images NHWC float32, file ids); writes ``<output_dir>/<id>.png`` per example
    (reference :110-147) and returns the list of written paths."""
    os.makedirs(output_dir, exist_ok=True)
    written = []
    for images, ids in batches:
        pred = predict_labels(net, images, size)
        if colormap is not None:
            out = colorize(pred, colormap)
        elif embedding_reversed is not None:
            out = reverse_embedding(pred, embedding_reversed)
        else:
            out = pred
        out = out.cpu().numpy()
        for k, fid in enumerate(ids):
            fid = fid.decode() if isinstance(fid, bytes) else str(fid)
            path = os.path.join(output_dir.rstrip("/"), fid + ".png")
            write_png(path, out[k])
            written.append(path)
    return written
