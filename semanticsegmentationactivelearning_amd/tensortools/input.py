"""Input front-end without TensorFlow: mirror of the reference's ``tensortools/input.py``
(``InputStage`` :34-329, ``NumpyCapsule`` :331-435, ``generate_mask`` :17-31).

One example per ``<id>.tfrecord`` file -> parse (6 default keys, :165-172) -> decode image (PNG/JPEG via
Pillow; ``tf.image.decode_image``, :246-248) and label (PNG, or a 255-filled plane when the record has
no label, :249-260) -> evaluation path: centre crop to ``input_shape`` (:278-284); ranking / training
path (``augment=True``): random crop + random left-right flip, returning the UNDISTORTED image as
``image`` and a channel-scaled copy as ``image_dist`` (:296-329) -> float32 in [0,1]
(``convert_image_dtype`` = x * 1/255, :289-290) -> NHWC batches with the auxiliary side channels
(``labelled``, ``index``) zipped in (:145-149).  The last batch may be partial (no drop_remainder,
:193-194); iterating past the end raises ``StopIteration`` (the reference: ``tf.errors.OutOfRangeError``).

Decoding runs in a thread pool of ``cpu_share - 1`` workers (the reference uses tf.data's C++ pool with
``cpu_count - 1`` parallel calls, :15,177-190); Pillow releases the GIL while decoding.  Batches are
host numpy arrays; ``models.ENet`` / ``rank_confidence`` move them to the GPU.

New relative to the reference: ``modalities=("nir", ...)`` concatenates ``<modality>/data`` channels
after the RGB channels (the reference hard-codes 3 channels and never decodes modalities, :261-269);
BASELINE config C5 (RGB+NIR) needs it.
"""
import glob
import io
import logging
import os
from concurrent.futures import ThreadPoolExecutor

import weakref

import numpy as np

from . import tfrecord
from .._lib import usable_cores

DEFAULT_FORMAT = {  # reference input.py:165-172 (key -> FixedLenFeature default)
    "image/channels": -1, "image/data": b"", "label": b"", "height": -1, "width": -1, "id": b"",
}


def generate_mask(labels, mask_index=255):
    """binary mask that is zero where ``labels == mask_index``; masked labels are mapped to zero
    (reference :17-31).  labels: [...,1] or [...]; returns (labels, mask) without the channel axis."""
    lab = np.asarray(labels)
    if lab.ndim >= 3 and lab.shape[-1] == 1:
        lab = lab[..., 0]
    mask_bool = lab != mask_index
    mask = mask_bool.astype(lab.dtype)
    return np.where(mask_bool, lab, mask), mask


def _decode_image(data):
    from PIL import Image
    img = Image.open(io.BytesIO(data))
    arr = np.asarray(img)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    if arr.dtype != np.uint8:
        raise ValueError("only 8-bit images are supported (got %s)" % arr.dtype)
    return arr


class _Dataset:
    def __init__(self, filenames, aux, batch_size, augment, count):
        self.filenames, self.aux, self.batch_size, self.augment, self.count = \
            filenames, aux, batch_size, augment, count


# page-locked ring slots of every live InputStage: base address -> (bytes, weakref(stage), slot)
_PINNED_SLOTS = {}


def _drop_slots(bases):
    """finalizer of an InputStage: its ring slots leave the table with it"""
    for base in list(bases):
        _PINNED_SLOTS.pop(base, None)


def copy_issued(tensor, event):
    """Tell the InputStage that owns the page-locked memory behind ``tensor`` (a batch it yielded, or any slice / view /
    re-wrapped copy-free alias of one) that an asynchronous host-to-device copy reading it has been issued and is
    complete once ``event`` (a ``torch.cuda.Event``) has fired.  The stage waits for the event before it writes the next
    batch into that slot.  Returns True if the memory belongs to a ring slot, False otherwise (pageable batches, arrays
    the caller made itself): nothing to protect then."""
    try:
        ptr = int(tensor.data_ptr()) if hasattr(tensor, "data_ptr") else int(tensor.__array_interface__["data"][0])
    except Exception:
        return False
    for base, (nbytes, stage_ref, slot) in list(_PINNED_SLOTS.items()):
        if base <= ptr < base + nbytes:
            stage = stage_ref()
            if stage is None:  # a dead stage's stale range: drop it and keep looking (a live stage may own this memory now)
                _PINNED_SLOTS.pop(base, None)
                continue
            stage._pin_events.setdefault(slot, []).append(event)
            return True
    return False


class InputStage:
    """Holds named datasets and ONE re-initialisable iterator shared by them (reference :34-233)."""

    def __init__(self, input_shape=[512, 512], scope="Dataset", modalities=(), seed=None, workers=None,
                 image_dtype=np.float32, pin_memory=False, pin_buffers=6):
        """``image_dtype=np.uint8`` (not in the reference): hand out the undistorted image as the decoded uint8
        frame instead of float32 in [0,1]; ``ENet.score`` converts it on the GPU (same bits, a quarter of the
        host-to-device bytes).

        ``pin_memory=True`` (not in the reference; needs the GPU runtime): the image batch is assembled directly
        in page-locked memory and handed out as a CPU torch tensor, so that ``rank_confidence(..., prefetch=2)``
        copies it asynchronously while the previous batch is being scored.  The batches are views of a ring of
        ``pin_buffers`` buffers: a batch stays valid until ``pin_buffers - 1`` further batches have been drawn."""
        self.logger = logging.getLogger(__name__)
        if np.dtype(image_dtype) not in (np.dtype(np.float32), np.dtype(np.uint8)):
            raise ValueError("image_dtype must be float32 or uint8")
        self.image_dtype = np.dtype(image_dtype)
        self.pin_memory = bool(pin_memory)
        self.pin_buffers = int(pin_buffers)
        if self.pin_memory and self.pin_buffers < 2:
            raise ValueError("pin_buffers must be >= 2")
        self._pinned = []  # ring of page-locked image batch buffers (allocated lazily, reused)
        self._pin_pos = 0
        self._pin_events = {}  # slot -> event of the async host-to-device copy still reading that slot
        self._slot_bases = set()  # this stage's keys in _PINNED_SLOTS; removed when the stage dies
        weakref.finalize(self, _drop_slots, self._slot_bases)
        if len(input_shape) == 3:
            self.shape = list(input_shape)
        elif len(input_shape) == 2:
            self.shape = list(input_shape) + [None]
        else:
            self.logger.warning("Proceeding with unknown inputshape.")
            self.shape = [None, None, None]
        self.modalities = tuple(modalities)
        self.datasets = {}
        self._rng = np.random.default_rng(seed)
        self._workers = workers if workers is not None else max(1, usable_cores(cap=32) - 1)
        self._iter = None

    # ---- dataset registration -------------------------------------------------------------------
    def add_dataset(self, name, file_patterns, batch_size, epochs=1, parse_fn=None, decode_fn=None,
                    augment=None):
        """file-pattern form (reference :66-118); directories get the default ``*.tfrecord`` glob"""
        if not isinstance(file_patterns, list):
            file_patterns = [file_patterns]
        files = []
        for pat in file_patterns:
            if os.path.isdir(pat):
                pat = os.path.join(pat, "*.tfrecord")
            files.extend(sorted(glob.glob(pat)))
        files = files * max(1, int(epochs))
        self.datasets[name] = _Dataset(np.asarray(files), (), batch_size, bool(augment), len(files))
        return len(files)

    def add_dataset_from_placeholders(self, name, filenames, *aux_placeholders, batch_size=8,
                                      parse_fn=None, decode_fn=None, augment=None):
        """``filenames`` / aux arrays are supplied (or re-supplied through ``feed_dict``) at
        ``init_iterator`` time, exactly one epoch, caller shuffles (reference :120-155)."""
        n = len(filenames) if filenames is not None and hasattr(filenames, "__len__") else None
        self.datasets[name] = _Dataset(filenames, tuple(aux_placeholders), batch_size, bool(augment), n)
        return n

    def get_datset(self, name):  # sic: the reference spells it this way (:210-211)
        return self.datasets[name]

    # ---- iteration --------------------------------------------------------------------------------
    def init_iterator(self, name, sess=None, feed_dict=None):
        """(re)start the shared iterator on dataset ``name`` (reference :213-227).  ``feed_dict``
        maps the placeholder objects given to ``add_dataset_from_placeholders`` (e.g. the
        attributes of a ``NumpyCapsule``) to arrays, like ``NumpyCapsule.feed_dict``."""
        ds = self.datasets[name]
        files, aux = ds.filenames, ds.aux
        if feed_dict is not None:
            files = _resolve(files, feed_dict)
            aux = tuple(_resolve(a, feed_dict) for a in aux)
        files = [f.decode() if isinstance(f, bytes) else str(f) for f in np.asarray(files).tolist()]
        aux = tuple(np.asarray(a) for a in aux)
        for a in aux:
            if len(a) != len(files):
                raise ValueError("auxiliary array length %d != number of files %d" % (len(a), len(files)))
        self._iter = self._batches(files, aux, ds.batch_size, ds.augment)

    def get_output(self):
        """next batch of the shared iterator; raises StopIteration at the end of the dataset
        (the reference returns the iterator's get_next() tensors, :229-233)"""
        if self._iter is None:
            raise RuntimeError("init_iterator() has not been called")
        return next(self._iter)

    def __iter__(self):
        if self._iter is None:
            raise RuntimeError("init_iterator() has not been called")
        return self._iter

    def _batches(self, files, aux, batch_size, augment):
        seeds = self._rng.integers(0, 2 ** 63 - 1, size=len(files))
        with ThreadPoolExecutor(self._workers) as pool:
            window = max(2 * batch_size, self._workers)  # prefetch depth (reference: prefetch(batch_size))
            futures = []
            pos = 0

            def submit_until(limit):
                nonlocal pos
                while pos < len(files) and len(futures) < limit:
                    futures.append(pool.submit(self._load_one, files[pos], augment, int(seeds[pos])))
                    pos += 1

            start = 0
            while start < len(files):
                submit_until(window)
                n = min(batch_size, len(files) - start)
                items = [futures.pop(0).result() for _ in range(n)]
                submit_until(window)
                cols = list(zip(*items))
                first = self._stack_pinned(cols[0]) if self.pin_memory else np.stack(cols[0])
                batch = (first,) + tuple(np.stack(c) for c in cols[1:]) + tuple(a[start:start + n] for a in aux)
                start += n
                yield batch

    def _stack_pinned(self, images):
        """np.stack(images) written into the next buffer of the page-locked ring -> CPU torch tensor.

        A consumer that copies the batch to the GPU asynchronously (``active_learning.prefetch_to_device``) reports
        the copy-done event through ``copy_issued(tensor, event)`` (module level, below), which finds the ring slot by
        the ADDRESS of the tensor's memory -- slices, views, ``torch.as_tensor`` / numpy round trips of the batch all
        resolve to their slot.  A slot is never rewritten on the host while a DMA read of it is still in flight."""
        import torch
        shape = (len(images),) + tuple(images[0].shape)
        dtype = torch.uint8 if images[0].dtype == np.uint8 else torch.float32
        need = int(np.prod(shape))
        slot = self._pin_pos % self.pin_buffers
        self._pin_pos += 1
        if slot >= len(self._pinned):
            self._pinned.append(None)
        for ev in self._pin_events.pop(slot, ()):
            ev.synchronize()  # the previous batch in this slot has left for the GPU
        if self._pinned[slot] is None or self._pinned[slot].numel() < need or self._pinned[slot].dtype != dtype:
            if self._pinned[slot] is not None:
                _PINNED_SLOTS.pop(self._pinned[slot].data_ptr(), None)
                self._slot_bases.discard(self._pinned[slot].data_ptr())
            self._pinned[slot] = torch.empty(need, dtype=dtype, pin_memory=True)
            buf = self._pinned[slot]
            _PINNED_SLOTS[buf.data_ptr()] = (buf.numel() * buf.element_size(), weakref.ref(self), slot)
            self._slot_bases.add(buf.data_ptr())
        out = self._pinned[slot][:need].view(shape)
        np.stack(images, out=out.numpy())
        return out

    def copy_issued(self, tensor, event):
        """see the module-level ``copy_issued``"""
        return copy_issued(tensor, event)

    # ---- per-example work -------------------------------------------------------------------------
    def _load_one(self, filename, augment, seed):
        rec = tfrecord.read_tfrecord(filename)
        fmt = dict(DEFAULT_FORMAT)
        for m in self.modalities:
            fmt["%s/data" % m] = b""
        example = tfrecord.parse_single_example(rec, fmt)
        return self.default_decoder(example, augment=augment, rng=np.random.default_rng(seed))

    def default_decoder(self, example, *other_outputs, augment=False, rng=None):
        """reference :235-294 (+ ``_default_augmentation`` :296-329)"""
        image = _decode_image(example["image/data"])[:, :, :3]
        for m in self.modalities:
            data = example.get("%s/data" % m, b"")
            if not data:
                raise ValueError("record has no '%s/data' feature" % m)
            image = np.concatenate([image, _decode_image(data)], axis=2)
        h, w = image.shape[:2]
        if example["label"]:
            label = _decode_image(example["label"])[:, :, :1]
        else:
            hh = example["height"] if example["height"] > 0 else h
            ww = example["width"] if example["width"] > 0 else w
            label = np.full((hh, ww, 1), 255, dtype=np.uint8)
        channels = image.shape[2]
        ch, cw = self.shape[0], self.shape[1]
        if ch is None or cw is None:
            ch, cw = h, w
        if h < ch or w < cw:
            raise ValueError("example %dx%d is smaller than the network input %dx%d" % (h, w, ch, cw))
        stack = np.concatenate([image, label], axis=2)
        if augment:
            rng = rng if rng is not None else self._rng
            top = int(rng.integers(0, h - ch + 1))
            left = int(rng.integers(0, w - cw + 1))
            crop = stack[top:top + ch, left:left + cw]
            if rng.random() < 0.5:  # tf.image.random_flip_left_right
                crop = crop[:, ::-1]
            img = crop[:, :, :channels].astype(np.float32) * np.float32(1.0 / 255.0)
            px_scaling = rng.uniform(0.8, 1.4, size=channels).astype(np.float32)
            img_dist = np.clip(img * px_scaling, 0.0, 1.0).astype(np.float32)
            lab, mask = generate_mask(crop[:, :, channels:])
            raw = crop[:, :, :channels] if self.image_dtype == np.uint8 else img
            return (np.ascontiguousarray(raw), img_dist, lab, mask) + tuple(other_outputs)
        cy, cx = h // 2, w // 2  # reference :278-284 (height//2, width//2 of the record)
        top, left = cy - ch // 2, cx - cw // 2
        crop = stack[top:top + ch, left:left + cw]
        if self.image_dtype == np.uint8:
            img = crop[:, :, :channels]
        else:
            img = crop[:, :, :channels].astype(np.float32) * np.float32(1.0 / 255.0)
        lab, mask = generate_mask(crop[:, :, channels:])
        return (np.ascontiguousarray(img), lab, mask) + tuple(other_outputs)


def _resolve(obj, feed_dict):
    """look a placeholder object up in a feed_dict (by identity, then by name); arrays pass through"""
    for k, v in feed_dict.items():
        if k is obj:
            return v
    if isinstance(obj, Placeholder):
        for k, v in feed_dict.items():
            if isinstance(k, Placeholder) and k.name == obj.name:
                return v
        raise KeyError("placeholder %s is not in the feed_dict" % obj.name)
    return obj


class Placeholder:
    """stand-in for a tf.placeholder: a named handle that feed_dicts are keyed on"""

    def __init__(self, name, dtype, ndim):
        self.name, self.dtype, self.ndim = name, dtype, ndim

    def __repr__(self):
        return "<Placeholder %s>" % self.name


class NumpyCapsule:
    """Index / shuffle / sample bookkeeping over equally long numpy arrays (reference :331-435).
    Assigning an ndarray attribute registers it (``capsule.filenames = np.array([...])``); the
    attribute then reads back as the placeholder handle used to key ``feed_dict``."""

    def __init__(self, shuffle=True, seed=None):
        object.__setattr__(self, "_values", {})        # name -> ndarray
        object.__setattr__(self, "_placeholders", {})  # name -> Placeholder
        self.shuffle = shuffle
        self._length = 0
        self._cur_length = 0
        self._indices = np.zeros((0,), dtype=np.int64)
        self._full_range = np.zeros((0,), dtype=np.int64)
        self._sample_set = np.zeros((0,), dtype=np.int64)
        self._sample_size = 0
        self._sample_prob = None
        self._rng = np.random.default_rng(seed)

    @property
    def feed_dict(self):
        """{placeholder: array[selected indices]}; shuffled (plus ``sample_size`` examples drawn
        without replacement from the sample set) when ``shuffle`` (reference :347-367).  The
        reference's non-shuffle branch indexes a dict with an array (a defect, :364-366); here it
        returns the un-shuffled selection."""
        indices = np.asarray(self._indices).copy()
        if self.shuffle:
            if self._sample_size > 0:
                rand = self._rng.choice(self._sample_set, self._sample_size, replace=False, p=self._sample_prob)
                indices = np.concatenate((indices, rand))
            self._rng.shuffle(indices)
        return {self._placeholders[n]: self._values[n][indices] for n in self._placeholders}

    def set_indices(self, indices=None, sample_indices=None, sample_prob=None):
        """restrict to a subset of the values (reference :369-395); None = everything"""
        if indices is None:
            self._indices = self._full_range
            self._cur_length = self._length
            self._sample_set = np.zeros((0,), dtype=np.int64)
            self._sample_size = 0
            self._sample_prob = None
        else:
            self._indices = np.asarray(indices)
            self._cur_length = len(self._indices)
            if sample_indices is None:
                self._sample_set = self._full_range[np.isin(self._full_range, self._indices, invert=True)]
            else:
                self._sample_set = np.asarray(sample_indices)
                if sample_prob is not None and len(sample_prob) == len(self._sample_set):
                    self._sample_prob = sample_prob

    def set_sample_size(self, size):
        self._sample_size = size
        return self._sample_size

    def get_value(self, attribute):
        """array behind a placeholder handle (or attribute name)"""
        name = attribute.name if isinstance(attribute, Placeholder) else attribute
        return self._values[name]

    @property
    def size(self):
        return self._cur_length + self._sample_size

    def __setattr__(self, name, value):
        if isinstance(value, np.ndarray) and not name.startswith("_"):
            if name not in self._placeholders:
                self._placeholders[name] = Placeholder(name, value.dtype, value.ndim)
            self._values[name] = value
            if self._length != len(value):  # user keeps the arrays equally long (reference :422-427)
                object.__setattr__(self, "_length", len(value))
                object.__setattr__(self, "_full_range", np.arange(len(value)))
                self.set_indices()
            object.__setattr__(self, name, self._placeholders[name])
        else:
            object.__setattr__(self, name, value)


__all__ = ["InputStage", "NumpyCapsule", "generate_mask"]
