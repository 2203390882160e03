"""ENet building blocks: host-side mirror of the reference's ``models/enet/enet_modules.py``.

Same class names, constructor arguments, weight attribute names and weight layouts as the
reference (kernels HWIO, transposed-conv kernels HW-O-I, fp32), so checkpoints / positional
weight copies keyed on ``layer.variables`` keep working.  The arithmetic is NOT here: a layer
called with ``training=False`` runs the hand-written HIP kernels of ``libssal_hip.so`` through
the C ABI (``ssal_enet_run_layer``); there is no CPU or framework fallback.

reference: models/enet/enet_modules.py  Initial :45-224, Bottleneck :226-599,
BottleneckDownsample :601-938, BottleneckUpsample :940-1292, Final :1294-1381.
"""
import numpy as np

from ... import _lib


# ------------------------------------------------------------------------------------------------
# initializers (stand-ins for tf.initializers.* used as constructor defaults, enet.py:11-12)
# ------------------------------------------------------------------------------------------------
class glorot_uniform:
    """tf.initializers.glorot_uniform: U(-l, l), l = sqrt(6 / (fan_in + fan_out)), fans computed
    the TF way (receptive field x shape[-2] / shape[-1]; 1-D shapes use the length for both)."""

    def __init__(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def __call__(self, shape):
        shape = tuple(int(s) for s in shape)
        if len(shape) < 1:
            fan_in = fan_out = 1
        elif len(shape) == 1:
            fan_in = fan_out = shape[0]
        elif len(shape) == 2:
            fan_in, fan_out = shape
        else:
            rf = int(np.prod(shape[:-2]))
            fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
        limit = np.sqrt(6.0 / (fan_in + fan_out))
        return self._rng.uniform(-limit, limit, size=shape).astype(np.float32)


class constant:
    """tf.initializers.constant"""

    def __init__(self, value=0.0):
        self.value = value

    def __call__(self, shape):
        return np.full(tuple(int(s) for s in shape), self.value, dtype=np.float32)


def zeros(shape):
    return np.zeros(tuple(int(s) for s in shape), dtype=np.float32)


def ones(shape):
    return np.ones(tuple(int(s) for s in shape), dtype=np.float32)


class Variable:
    """Minimal stand-in for a tf.Variable: a named fp32 host array with assign()/numpy()."""

    def __init__(self, name, value, trainable=True):
        self.name = name
        self.trainable = trainable
        self._value = np.ascontiguousarray(value, dtype=np.float32)
        self.version = 0

    @property
    def shape(self):
        return self._value.shape

    def numpy(self):
        return self._value

    def assign(self, value):
        value = np.asarray(value.numpy() if hasattr(value, "numpy") else value, dtype=np.float32)
        if value.shape != self._value.shape:
            raise ValueError("cannot assign shape %s to variable %s of shape %s"
                             % (value.shape, self.name, self._value.shape))
        self._value = np.ascontiguousarray(value)
        self.version += 1
        return self

    def __array__(self, dtype=None, copy=None):
        return self._value if dtype is None else self._value.astype(dtype)

    def __repr__(self):
        return "<Variable %s shape=%s>" % (self.name, self._value.shape)


class Layer:
    """Common plumbing: weight registry (Keras ordering of ``.variables``) + dispatch to the owning ENet handle."""

    def __init__(self, name):
        self.name = name
        self.built = False
        self._variables = []
        self._owner = None  # set by ENet

    def add_weight(self, name, shape, initializer, trainable=True, **_ignored):
        v = Variable("%s/%s" % (self.name, name), initializer(shape), trainable=trainable)
        self._variables.append(v)
        return v

    @property
    def trainable_weights(self):
        return [v for v in self._variables if v.trainable]

    @property
    def non_trainable_weights(self):
        return [v for v in self._variables if not v.trainable]

    @property
    def weights(self):
        """TF-1.13 Keras order: ``trainable_weights + non_trainable_weights``, each group in creation
        order -- i.e. the moving batch-norm ``mean`` / ``variance`` (created ``trainable=False``,
        reference enet_modules.py:150-163, 384-397, ...) come LAST.  This is the order the reference's
        positional weight copy ``val_net.layers[i].variables[j] <- train_net...`` walks
        (active_learning.py:475-482), so a copy from a real ``tf_net`` lands on the right tensors."""
        return self.trainable_weights + self.non_trainable_weights

    @property
    def variables(self):
        return self.weights

    @property
    def creation_order_variables(self):
        """the same variables in ``add_weight`` order (the seeded synthetic recipes draw in this order)"""
        return list(self._variables)

    # name -> Variable mapping in C-ABI naming ("<Layer>.<attr>")
    def abi_tensors(self):
        raise NotImplementedError

    def _run(self, inputs, training, argmax_in=None, want_argmax=False, arithmetic="f32"):
        if training:
            raise NotImplementedError(
                "training=True (spatial dropout + batch statistics) is outside the MI355X "
                "scoring path; call with training=False")
        if self._owner is None:
            raise RuntimeError("layer %s is not attached to an ENet; layers execute through the "
                               "owning model's device handle" % self.name)
        x = _lib.as_device_f32(inputs)
        if x.dim() != 4:
            raise ValueError("inputs must be NHWC rank-4 (got shape %s)" % (tuple(x.shape),))
        if not self.built:
            self.build(tuple(x.shape))
        return self._owner._run_layer(self, x, argmax_in, want_argmax, arithmetic)

    def output_shape(self, n, h, w):
        raise NotImplementedError


def _bn(layer, prefix, tag, c):
    """the four batch-norm statistics in the reference's creation order"""
    setattr(layer, prefix + "mean", layer.add_weight(tag + "BatchNorm/Mean", [c], zeros, trainable=False))
    setattr(layer, prefix + "variance", layer.add_weight(tag + "BatchNorm/Variance", [c], ones, trainable=False))
    setattr(layer, prefix + "gamma", layer.add_weight(tag + "BatchNorm/Gamma", [c], ones))
    setattr(layer, prefix + "beta", layer.add_weight(tag + "BatchNorm/Beta", [c], zeros))


class Initial(Layer):
    """concat[conv3x3/s2 (Cin -> 16-Cin), maxpool2x2/s2] -> BatchNorm -> PReLU
    (reference enet_modules.py:45-224)."""

    def __init__(self, output_channels, kernel_size=(3, 3), strides=(2, 2), pool_size=(2, 2),
                 padding="SAME", dilation_rate=(1, 1), kernel_initializer=None,
                 alpha_initializer=None, trainable=True, kernel_regularizer=None,
                 alpha_regularizer=None, regularization_scaling=False, batch_norm_momentum=0.90,
                 name="Initial", **kwargs):
        super().__init__(name)
        if (tuple(kernel_size), tuple(strides), tuple(pool_size), padding, tuple(dilation_rate)) != \
                ((3, 3), (2, 2), (2, 2), "SAME", (1, 1)) or output_channels != 16:
            raise NotImplementedError("the HIP Initial block implements the ENet configuration only "
                                      "(3x3/s2 conv + 2x2/s2 max-pool, SAME, 16 channels)")
        self.output_channels = output_channels
        self.kernel_initializer = kernel_initializer or glorot_uniform()
        self.alpha_initializer = alpha_initializer or constant(0.25)

    def build(self, input_shape):
        if self.built:
            return
        cin = int(input_shape[-1])
        self.kernel = self.add_weight("Convolution/Kernel", (3, 3, cin, self.output_channels - cin),
                                      self.kernel_initializer)
        _bn(self, "", "Convolution/", self.output_channels)
        self.alpha = self.add_weight("Residual/Alpha", [self.output_channels], self.alpha_initializer)
        self.built = True

    def abi_tensors(self):
        return {"kernel": self.kernel, "mean": self.mean, "variance": self.variance,
                "gamma": self.gamma, "beta": self.beta, "alpha": self.alpha}

    def output_shape(self, n, h, w):
        return (n, h // 2, w // 2, self.output_channels)

    def __call__(self, inputs, training, **kwargs):
        return self._run(inputs, training)


class Bottleneck(Layer):
    """1x1 proj -> {3x3 (dilated) | (5,1)+(1,5)} -> 1x1 exp, + identity residual, PReLU
    (reference enet_modules.py:226-599)."""

    def __init__(self, output_channels, kernel_size=(3, 3), asymmetric=False, padding="SAME",
                 projection_rate=4, dilation_rate=(1, 1), kernel_initializer=None,
                 kernel_regularizer=None, regularization_scaling=False, alpha_initializer=None,
                 trainable=True, drop_rate=0.1, batch_norm_momentum=0.90, name="Bottleneck",
                 **kwargs):
        super().__init__(name)
        self.output_channels = output_channels
        self.kernel_size = tuple(kernel_size)
        self.asymmetric = asymmetric
        self.projection_rate = projection_rate
        self.padding = padding
        self.dilation_rate = tuple(dilation_rate)
        self.drop_rate = drop_rate
        self.kernel_initializer = kernel_initializer or glorot_uniform()
        self.alpha_initializer = alpha_initializer or constant(0.25)
        if padding != "SAME" or projection_rate != 4 or self.dilation_rate[0] != self.dilation_rate[1]:
            raise NotImplementedError("HIP Bottleneck: SAME padding, projection_rate 4, square dilation only")
        if (asymmetric and self.kernel_size != (5, 5)) or (not asymmetric and self.kernel_size != (3, 3)):
            raise NotImplementedError("HIP Bottleneck: 3x3 or asymmetric (5,5) kernels only")

    def build(self, input_shape):
        if self.built:
            return
        c = int(input_shape[-1])
        f = c // self.projection_rate
        self.proj_kernel = self.add_weight("Projection/Kernel", (1, 1, c, f), self.kernel_initializer)
        self.proj_alpha = self.add_weight("Projection/Alpha", [f], self.alpha_initializer)
        _bn(self, "proj_", "Projection/", f)
        if self.asymmetric:
            self.conv_kernel = [
                self.add_weight("Convolution/KernelCol", (self.kernel_size[0], 1, f, f), self.kernel_initializer),
                self.add_weight("Convolution/KernelRow", (1, self.kernel_size[1], f, f), self.kernel_initializer),
            ]
        else:
            self.conv_kernel = self.add_weight("Convolution/Kernel", self.kernel_size + (f, f),
                                               self.kernel_initializer)
        # NOTE the reference initialises conv_alpha with the *kernel* initializer (enet_modules.py:442-449)
        self.conv_alpha = self.add_weight("Convolution/Alpha", [f], self.kernel_initializer)
        _bn(self, "conv_", "Convolution/", f)
        self.exp_kernel = self.add_weight("Expansion/Kernel", (1, 1, f, self.output_channels),
                                          self.kernel_initializer)
        _bn(self, "exp_", "Expansion/", self.output_channels)
        self.residual_alpha = self.add_weight("Residual/Alpha", [self.output_channels], self.alpha_initializer)
        self.built = True

    def abi_tensors(self):
        t = {"proj_kernel": self.proj_kernel, "proj_alpha": self.proj_alpha}
        for p in ("proj_", "conv_", "exp_"):
            for s in ("mean", "variance", "gamma", "beta"):
                t[p + s] = getattr(self, p + s)
        if self.asymmetric:
            t["conv_kernel.0"], t["conv_kernel.1"] = self.conv_kernel
        else:
            t["conv_kernel"] = self.conv_kernel
        t["conv_alpha"] = self.conv_alpha
        t["exp_kernel"] = self.exp_kernel
        t["residual_alpha"] = self.residual_alpha
        return t

    def output_shape(self, n, h, w):
        return (n, h, w, self.output_channels)

    def __call__(self, inputs, training, arithmetic="f32", **kwargs):
        return self._run(inputs, training, arithmetic=arithmetic)


class BottleneckDownsample(Layer):
    """2x2/s2 proj -> 3x3 -> 1x1 exp, + zero-padded max_pool_with_argmax residual, PReLU; returns
    (output, argmax)  (reference enet_modules.py:601-938)."""

    def __init__(self, output_channels, kernel_size=(3, 3), padding="SAME", projection_rate=4,
                 dilation_rate=(1, 1), kernel_initializer=None, kernel_regularizer=None,
                 regularization_scaling=False, alpha_initializer=None, trainable=True,
                 drop_rate=0.1, batch_norm_momentum=0.90, name="BottleneckDownsample", **kwargs):
        super().__init__(name)
        self.output_channels = output_channels
        self.kernel_size = tuple(kernel_size)
        self.projection_rate = projection_rate
        self.padding = padding
        self.dilation_rate = tuple(dilation_rate)
        self.drop_rate = drop_rate
        self.kernel_initializer = kernel_initializer or glorot_uniform()
        self.alpha_initializer = alpha_initializer or constant(0.25)
        if padding != "SAME" or projection_rate != 4 or self.kernel_size != (3, 3) or self.dilation_rate != (1, 1):
            raise NotImplementedError("HIP BottleneckDownsample: ENet configuration only")

    def build(self, input_shape):
        if self.built:
            return
        c = int(input_shape[-1])
        f = 2 * (c // self.projection_rate)  # reference :705
        self.zero_padding = [[0, 0], [0, 0], [0, 0], [0, self.output_channels - c]]
        self.proj_kernel = self.add_weight("Projection/Kernel", (2, 2, c, f), self.kernel_initializer)
        self.proj_alpha = self.add_weight("Projection/Alpha", [f], self.alpha_initializer)
        _bn(self, "proj_", "Projection/", f)
        self.conv_kernel = self.add_weight("Convolution/Kernel", self.kernel_size + (f, f), self.kernel_initializer)
        self.conv_alpha = self.add_weight("Convolution/Alpha", [f], self.kernel_initializer)
        _bn(self, "conv_", "Convolution/", f)
        self.exp_kernel = self.add_weight("Expansion/Kernel", (1, 1, f, self.output_channels), self.kernel_initializer)
        _bn(self, "exp_", "Expansion/", self.output_channels)
        self.residual_alpha = self.add_weight("Residual/Alpha", [self.output_channels], self.alpha_initializer)
        self.built = True

    abi_tensors = Bottleneck.abi_tensors
    asymmetric = False

    def output_shape(self, n, h, w):
        return (n, h // 2, w // 2, self.output_channels)

    def __call__(self, inputs, training, arithmetic="f32", **kwargs):
        return self._run(inputs, training, want_argmax=True, arithmetic=arithmetic)


class BottleneckUpsample(Layer):
    """1x1 proj -> conv2d_transpose 3x3/s2 -> 1x1 exp, + unpool_2d(1x1 conv(inputs), argmax), PReLU
    (reference enet_modules.py:940-1292)."""

    def __init__(self, output_channels, kernel_size=(3, 3), padding="SAME", projection_rate=4,
                 dilation_rate=(1, 1), kernel_initializer=None, kernel_regularizer=None,
                 regularization_scaling=False, alpha_initializer=None, trainable=True,
                 drop_rate=0.1, batch_norm_momentum=0.90, name="BottleneckUpsample", **kwargs):
        super().__init__(name)
        self.output_channels = output_channels
        self.kernel_size = tuple(kernel_size)
        self.projection_rate = projection_rate
        self.padding = padding
        self.dilation_rate = tuple(dilation_rate)
        self.drop_rate = drop_rate
        self.kernel_initializer = kernel_initializer or glorot_uniform()
        self.alpha_initializer = alpha_initializer or constant(0.25)
        if padding != "SAME" or projection_rate != 4 or self.kernel_size != (3, 3):
            raise NotImplementedError("HIP BottleneckUpsample: ENet configuration only")

    def build(self, input_shape):
        if self.built:
            return
        c = int(input_shape[-1])
        pf = c // self.projection_rate
        cf = pf // 2
        self.proj_kernel = self.add_weight("Projection/Kernel", (1, 1, c, pf), self.kernel_initializer)
        self.proj_alpha = self.add_weight("Projection/Alpha", [pf], self.alpha_initializer)
        _bn(self, "proj_", "Projection/", pf)
        self.conv_kernel = self.add_weight("Convolution/Kernel", self.kernel_size + (cf, pf), self.kernel_initializer)
        self.conv_alpha = self.add_weight("Convolution/Alpha", [cf], self.kernel_initializer)
        _bn(self, "conv_", "Convolution/", cf)
        self.exp_kernel = self.add_weight("Expansion/Kernel", (1, 1, cf, self.output_channels), self.kernel_initializer)
        _bn(self, "exp_", "Expansion/", self.output_channels)
        self.res_kernel = self.add_weight("Residual/Kernel", (1, 1, c, self.output_channels), self.kernel_initializer)
        self.residual_alpha = self.add_weight("Residual/Alpha", [self.output_channels], self.alpha_initializer)
        self.built = True

    asymmetric = False

    def abi_tensors(self):
        t = Bottleneck.abi_tensors(self)
        t["res_kernel"] = self.res_kernel
        return t

    def output_shape(self, n, h, w):
        return (n, 2 * h, 2 * w, self.output_channels)

    def __call__(self, inputs, unpool_argmax, training, arithmetic="f32", **kwargs):
        return self._run(inputs, training, argmax_in=unpool_argmax, arithmetic=arithmetic)


class Final(Layer):
    """conv2d_transpose 3x3/s2 SAME, 16 -> classes, no bias / BN / activation
    (reference enet_modules.py:1294-1381)."""

    def __init__(self, classes, kernel_size=(3, 3), padding="SAME", dilation_rate=(1, 1),
                 kernel_initializer=None, kernel_regularizer=None, regularization_scaling=False,
                 name="Final", **kwargs):
        super().__init__(name)
        self.classes = classes
        self.kernel_size = tuple(kernel_size)
        self.padding = padding
        self.kernel_initializer = kernel_initializer or glorot_uniform()
        if padding != "SAME" or self.kernel_size != (3, 3):
            raise NotImplementedError("HIP Final: 3x3 SAME transposed conv only")

    def build(self, input_shape):
        if self.built:
            return
        c = int(input_shape[-1])
        self.kernel = self.add_weight("Kernel", self.kernel_size + (self.classes, c), self.kernel_initializer)
        self.built = True

    def abi_tensors(self):
        return {"kernel": self.kernel}

    def output_shape(self, n, h, w):
        return (n, 2 * h, 2 * w, self.classes)

    def __call__(self, inputs, training=False, **kwargs):
        return self._run(inputs, training)
