"""ICNet model (BASELINE config C4: multi-scale 1/4, 1/2, 1 at 1024x2048, margin acquisition).

The reference's ``models/icnet/icnet.py:1-7`` is an empty class whose docstring cites the ICNet paper, so there is
no reference interface or behaviour to mirror: ``ICNET_SPEC.md`` pins the architecture (the paper's three-branch
cascade with the half-width PSPNet50 backbone) and expresses every operator with the semantics the reference
repository defines for it.  The surface follows ``models.ENet``: ``ICNet(classes, ...)(inputs_NHWC, training=False)
-> logits [N,H,W,classes]``, ``.layers[i].variables[j]`` in Keras order, ``.score()`` = forward + softmax +
acquisition measure + float64 per-image mean (``active_learning.py:229-263``) fused on the GPU.  All arithmetic runs
in the hand-written HIP kernels of libssal_hip.so through the C ABI (``include/ssal_icnet.h``); no CPU fallback.
"""
import ctypes

from ... import _lib
from ..enet import enet_modules as mod

# (name, cin, mid, cout, stride, dilation, projection shortcut) -- ICNET_SPEC.md sections 1 and 2
BOTTLENECKS = ([("conv2_1", 64, 32, 128, 1, 1, True), ("conv2_2", 128, 32, 128, 1, 1, False),
                ("conv2_3", 128, 32, 128, 1, 1, False), ("conv3_1", 128, 64, 256, 2, 1, True)]
               + [("conv3_%d" % i, 256, 64, 256, 1, 1, False) for i in (2, 3, 4)]
               + [("conv4_1", 256, 128, 512, 1, 2, True)]
               + [("conv4_%d" % i, 512, 128, 512, 1, 2, False) for i in (2, 3, 4, 5, 6)]
               + [("conv5_1", 512, 256, 1024, 1, 4, True)]
               + [("conv5_%d" % i, 1024, 256, 1024, 1, 4, False) for i in (2, 3)])


class ConvBN(mod.Layer):
    """conv (SAME, no bias) -> batch-norm [-> ReLU]: the unit every ICNet layer is built from.  Weights:
    ``kernel`` HWIO, ``gamma`` / ``beta`` (trainable), moving ``mean`` / ``variance`` (non-trainable, so they come
    last in ``.variables`` as in Keras)."""

    def __init__(self, name, kernel_size, cin, cout, stride=1, dilation=1, relu=True, kernel_initializer=None):
        super().__init__(name)
        self.kernel_size, self.cin, self.cout = int(kernel_size), int(cin), int(cout)
        self.stride, self.dilation, self.relu = int(stride), int(dilation), bool(relu)
        self.kernel_initializer = kernel_initializer or mod.glorot_uniform()

    def build(self, input_shape=None):
        if self.built:
            return
        k = self.kernel_size
        self.kernel = self.add_weight("Kernel", (k, k, self.cin, self.cout), self.kernel_initializer)
        self.mean = self.add_weight("BatchNorm/Mean", [self.cout], mod.zeros, trainable=False)
        self.variance = self.add_weight("BatchNorm/Variance", [self.cout], mod.ones, trainable=False)
        self.gamma = self.add_weight("BatchNorm/Gamma", [self.cout], mod.ones)
        self.beta = self.add_weight("BatchNorm/Beta", [self.cout], mod.zeros)
        self.built = True

    def abi_tensors(self):
        return {"kernel": self.kernel, "mean": self.mean, "variance": self.variance, "gamma": self.gamma,
                "beta": self.beta}


class Classifier(mod.Layer):
    """conv6_cls: 1x1 convolution with bias, no batch-norm / activation (ICNET_SPEC section 4)."""

    def __init__(self, name, cin, classes, kernel_initializer=None):
        super().__init__(name)
        self.cin, self.classes = int(cin), int(classes)
        self.kernel_initializer = kernel_initializer or mod.glorot_uniform()

    def build(self, input_shape=None):
        if self.built:
            return
        self.kernel = self.add_weight("Kernel", (1, 1, self.cin, self.classes), self.kernel_initializer)
        self.bias = self.add_weight("Bias", [self.classes], mod.zeros)
        self.built = True

    def abi_tensors(self):
        return {"kernel": self.kernel, "bias": self.bias}


def conv_layers(c_in, classes, kernel_initializer=None):
    """the flat list of parameterised layers in ICNET_SPEC order (names = C-ABI layer names)"""
    ki = kernel_initializer
    L = [ConvBN("conv1_1_3x3_s2", 3, c_in, 32, 2, kernel_initializer=ki),
         ConvBN("conv1_2_3x3", 3, 32, 32, kernel_initializer=ki),
         ConvBN("conv1_3_3x3", 3, 32, 64, kernel_initializer=ki)]
    for name, cin, mid, cout, s, d, proj in BOTTLENECKS:
        L.append(ConvBN(name + "_1x1_reduce", 1, cin, mid, s, kernel_initializer=ki))
        L.append(ConvBN(name + "_3x3", 3, mid, mid, 1, d, kernel_initializer=ki))
        L.append(ConvBN(name + "_1x1_increase", 1, mid, cout, relu=False, kernel_initializer=ki))
        if proj:
            L.append(ConvBN(name + "_1x1_proj", 1, cin, cout, s, relu=False, kernel_initializer=ki))
    L += [ConvBN("conv5_4_k1", 1, 1024, 256, kernel_initializer=ki),
          ConvBN("conv_sub4", 3, 256, 128, 1, 2, relu=False, kernel_initializer=ki),
          ConvBN("conv3_1_sub2_proj", 1, 256, 128, relu=False, kernel_initializer=ki),
          ConvBN("conv_sub2", 3, 128, 128, 1, 2, relu=False, kernel_initializer=ki),
          ConvBN("conv1_sub1", 3, c_in, 32, 2, kernel_initializer=ki),
          ConvBN("conv2_sub1", 3, 32, 32, 2, kernel_initializer=ki),
          ConvBN("conv3_sub1", 3, 32, 64, 2, kernel_initializer=ki),
          ConvBN("conv3_sub1_proj", 1, 64, 128, relu=False, kernel_initializer=ki),
          Classifier("conv6_cls", 128, classes, kernel_initializer=ki)]
    return L


class ICNet(_lib.DeviceState):
    """
    http://openaccess.thecvf.com/content_ECCV_2018/papers/Hengshuang_Zhao_ICNet_for_Real-Time_ECCV_2018_paper.pdf
    (the paper the reference's empty ``models/icnet/icnet.py:3`` cites; architecture pinned in ICNET_SPEC.md)
    """

    def __init__(self, classes, kernel_initializer=None, weight_regularization=None, name="ICNet"):
        self.classes = int(classes)
        self.name = name
        self.kernel_initializer = kernel_initializer
        self.built = False
        self._layers = []
        self._c_in = None
        self._init_device_state()  # per-device handles, per-(device, stream) workspaces (_lib.DeviceState)
        self.outputs = []

    # ---- keras-like surface ----------------------------------------------------------------
    @property
    def layers(self):
        return list(self._layers)

    @property
    def variables(self):
        return [v for l in self._layers for v in l.variables]

    weights = variables

    def build(self, input_shape):
        if self.built:
            return
        self._c_in = int(input_shape[-1])
        self._layers = conv_layers(self._c_in, self.classes, self.kernel_initializer)
        for l in self._layers:
            l.build()
            l._owner = self
            setattr(self, l.name, l)
        self.built = True

    def assign_named(self, named, strict=False):
        """name-keyed weight copy (same contract as ``ENet.assign_named``)"""
        mine = {v.name: v for v in self.variables}
        seen = set()
        for name, value in named.items():
            key = name[:-2] if name.endswith(":0") else name
            while key not in mine and "/" in key:
                key = key.split("/", 1)[1]
            if key not in mine:
                raise KeyError("no variable of %s matches '%s'" % (self.name, name))
            mine[key].assign(value)
            seen.add(key)
        if strict and len(seen) != len(mine):
            raise KeyError("missing variables: %s" % sorted(set(mine) - seen)[:5])
        return len(seen)

    # ---- device handle ---------------------------------------------------------------------
    def _create_handle(self, L):
        h = ctypes.c_void_p()
        _lib.check(L.ssal_icnet_create(self._c_in, self.classes, ctypes.byref(h)))
        return h

    def _push_tensors(self, L, handle):
        for layer in self._layers:
            for attr, var in layer.abi_tensors().items():
                arr = var.numpy()
                _lib.check(L.ssal_icnet_set_tensor(handle, ("%s.%s" % (layer.name, attr)).encode(),
                                                   arr.ctypes.data_as(ctypes.c_void_p), arr.size))
        _lib.check(L.ssal_icnet_commit(handle, _lib.stream_ptr()))

    def _destroy_handle(self, L, handle):
        L.ssal_icnet_destroy(handle)

    def __del__(self):
        self._release_device_state()

    def _prepare(self, inputs, training):
        if training:
            raise NotImplementedError("training=True (auxiliary heads + batch statistics) is outside the MI355X "
                                      "scoring path; call with training=False")
        x = _lib.as_device_image(inputs)
        if x.dim() != 4:
            raise ValueError("inputs must be NHWC rank-4 (got shape %s)" % (tuple(x.shape),))
        if not self.built:
            self.build(tuple(x.shape))
        if x.shape[-1] != self._c_in:
            raise ValueError("model was built for %d input channels, got %d" % (self._c_in, x.shape[-1]))
        if x.shape[1] % 32 or x.shape[2] % 32:
            raise ValueError("ICNet needs H and W divisible by 32 (got %dx%d)" % (x.shape[1], x.shape[2]))
        return x

    # ---- forward ---------------------------------------------------------------------------
    def __call__(self, inputs, training=False):
        torch = _lib.require_gpu() if not training else None
        x = self._prepare(inputs, training)
        n, h, w, _ = x.shape
        L = _lib.lib()
        with torch.cuda.device(x.device):
            handle = self._sync_handle()
            ws = self._workspace(L.ssal_icnet_workspace_bytes(handle, n, h, w), x.device)
            logits = torch.empty((n, h, w, self.classes), dtype=torch.float32, device=x.device)
            fwd = L.ssal_icnet_forward_nhwc_u8 if x.dtype == torch.uint8 else L.ssal_icnet_forward_nhwc
            _lib.check(fwd(handle, _lib.dev_ptr(x), n, h, w, _lib.dev_ptr(logits), _lib.dev_ptr(ws), ws.numel(),
                           _lib.stream_ptr()))
            self._note_call(ws, (n, h, w), "forward")
        self.outputs = [logits]  # eager: keep only the most recent call
        return logits

    call = __call__

    def score(self, inputs, measure="margin", threshold=0.0, return_label=False, return_mask=False,
              return_confidence=False, out=None):
        """forward(training=False) + softmax + acquisition measure + float64 per-image mean; the 4x bilinear
        up-sampling of the 1/4-resolution class scores happens inside the score kernel (the full-resolution
        logits never reach HBM).  Returns scores [N] float64 (device), optionally the per-pixel maps."""
        if measure not in _lib.MEASURES:
            raise NotImplementedError("Uncertainty function not implemented.")
        torch = _lib.require_gpu()
        x = self._prepare(inputs, False)
        n, h, w, _ = x.shape
        L = _lib.lib()
        with torch.cuda.device(x.device):
            handle = self._sync_handle()
            ws = self._workspace(L.ssal_icnet_workspace_bytes(handle, n, h, w), x.device)
            scores = out if out is not None else torch.empty((n,), dtype=torch.float64, device=x.device)
            label = torch.empty((n, h, w), dtype=torch.uint8, device=x.device) if return_label else None
            mask = torch.empty((n, h, w), dtype=torch.uint8, device=x.device) if return_mask else None
            conf = torch.empty((n, h, w), dtype=torch.float32, device=x.device) if return_confidence else None
            fn = L.ssal_icnet_score_nhwc_u8 if x.dtype == torch.uint8 else L.ssal_icnet_score_nhwc
            _lib.check(fn(handle, _lib.dev_ptr(x), n, h, w, _lib.MEASURES[measure], float(threshold),
                          _lib.dev_ptr(scores, torch.float64, "scores"), _lib.dev_ptr(label), _lib.dev_ptr(mask),
                          _lib.dev_ptr(conf), _lib.dev_ptr(ws), ws.numel(), _lib.stream_ptr()))
            self._note_call(ws, (n, h, w), "score")
        if return_label or return_mask or return_confidence:
            return scores, {"label": label, "mask": mask, "confidence": conf}
        return scores

    # ---- intermediate tensors of the most recent call --------------------------------------
    def endpoint_names(self):
        L = _lib.lib()
        handle = self._sync_handle()
        names = []
        for i in range(L.ssal_icnet_num_endpoints(handle)):
            p = ctypes.c_char_p()
            _lib.check(L.ssal_icnet_endpoint_name(handle, i, ctypes.byref(p)))
            names.append(p.value.decode())
        return names

    def endpoint(self, name):
        """view of the named ICNET_SPEC layer output of the calling thread's most recent call (overwritten by the next
        one).  After ``__call__`` every layer output is materialised.  ``score()`` runs fused launches (branch fronts,
        projection shortcuts inside the increase launch, whole identity blocks) that never write some layer outputs:
        asking for one of those after a ``score()`` raises ``RuntimeError`` instead of handing out stale memory
        (``ssal_icnet_endpoint_valid_after_score``, include/ssal_icnet.h)."""
        torch = _lib.require_gpu()
        if self._last_dims is None:
            raise RuntimeError("no forward pass has run yet")
        n, h, w = self._last_dims
        L = _lib.lib()
        if self._last_call == "score":
            ok = L.ssal_icnet_endpoint_valid_after_score(self._handle, name.encode(), h, w)
            if ok < 0:
                _lib.check(_lib.SSAL_EINVAL)
            if ok == 0:
                raise RuntimeError("'%s' is not written by score() (it lives inside a fused launch); call the model "
                                   "(forward) to materialise every layer output" % name)
        off = ctypes.c_int64()
        dims = (ctypes.c_int64 * 4)()
        _lib.check(L.ssal_icnet_endpoint_info(self._handle, name.encode(), n, h, w, ctypes.byref(off), dims))
        shp = tuple(int(d) for d in dims)
        cnt = shp[0] * shp[1] * shp[2] * shp[3]
        return self._ws[off.value:off.value + 4 * cnt].view(torch.float32).view(shp)
